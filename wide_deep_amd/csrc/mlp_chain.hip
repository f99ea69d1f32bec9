// wide_deep_amd/csrc/mlp_chain.hip -- the one-launch `simple` tower: the C entry points (wd_tower_chain, wd_chain_tail) and the
// dense tail kernel.  The tower kernel itself is csrc/mlp_chain8.hip (two wavefronts per SIMD; it replaced the one-wavefront-
// per-SIMD kernel of rounds 1-3 that lived here: 78 -> 66 us per launch at the C2 shape, profiles/r4_tower8_ablation.txt).
//
// What the launch computes, for one 32-example row tile per workgroup (python/lib/dnn.py:92-141: dense -> activation -> BN per
// layer, then the units=1 logits layer python/lib/dnn.py:226-232, joined with the head python/lib/joint.py:216-222 add_n of the
// wide and deep logits, joint.py:264-269 sigmoid CE): the chain of launches  NN_0 .. NN_{L-1}, logits head, NT_{L-1} .. NT_0  of
// mlp.hip in ONE kernel -- a row tile's forward AND its input-gradient chain depend on nothing but that tile's rows (the loss is a
// batch SUM, BN is the inference affine, applied as written: bn = gamma inv a + beta; nothing is folded into the weights), so the
// tile stays in LDS from the input layer's output x down to the gradient dx that the embedding update consumes.  What remains
// outside are the weight-gradient products G_l = bn_{l-1}^T dz_l, which reduce over the whole batch (split-K GEMMs of mlp.hip);
// the kernel leaves bn_l and dz_l in HBM for them.  Exact fp32 (v_mfma_f32_32x32x2_f32 == an fmaf chain).
#include "mlp_chain.h"

namespace {
using namespace wd_chain;

// ---- the dense tail of a step: split-K partials -> gradients -> Adagrad -> packed operands of the next tower launch --------
// With nothing folded every dense parameter's update depends on that parameter alone (python/lib/joint.py:233-241,
// tf.train.AdagradOptimizer on the dnn scope): one thread per parameter sums its partials in split order, stores the gradient,
// takes the Adagrad step and, for a hidden-layer kernel element, rewrites the two MFMA-packed copies.  No row or column
// reductions, no ordering between workgroups -- the BN and bias gradients arrive as finished column sums (column-sum jobs of
// wd_gemm_tn_splitk_group over the tower kernel's per-tile partials).
struct TailArgs {
  wd_tail_layer_t layer[MAXL + 1];
  int32_t first[MAXL + 2];   // first workgroup of layer l (256 parameters per workgroup)
  int32_t nlayers;
  int32_t mode;
  float *P, *Pacc, *Gflat;
  float inv, lr;
};

__global__ void __launch_bounds__(256) k_chain_tail(TailArgs g) {
  int l = 0;
  while (l + 1 < g.nlayers && (int)blockIdx.x >= g.first[l + 1]) ++l;
  l = uni(l);
  const wd::TailCtx c{g.P, g.Pacc, g.Gflat, g.inv, g.lr};
  wd::tail_element(g.layer[l], c, (int64_t)(blockIdx.x - g.first[l]) * 256 + threadIdx.x, g.mode);
}

}  // namespace

extern "C" int wd_chain_tail(const wd_tail_layer_t *layers, int32_t nlayers, float *P, float *Pacc, float *Gflat, float inv,
                             float lr, int32_t mode, wd_stream_t stream) {
  WD_REQUIRE(layers && P && Gflat, "null pointer");
  WD_REQUIRE(nlayers >= 1 && nlayers <= MAXL + 1, "1 <= nlayers <= WD_CHAIN_MAX_LAYERS + 1");
  WD_REQUIRE(!(mode & WD_TAIL_UPDATE) || Pacc, "the update needs the Adagrad accumulators");
  TailArgs g{};
  int32_t nb = 0;
  for (int l = 0; l < nlayers; ++l) {
    g.layer[l] = layers[l];
    WD_REQUIRE(layers[l].K > 0 && layers[l].N > 0, "layer shape");
    WD_REQUIRE(!(mode & WD_TAIL_GRAD) || layers[l].Gpart, "gradient mode needs the partials");
    g.first[l] = nb;
    nb += (int32_t)wd::ceil_div(layers[l].K * layers[l].N + 3 * layers[l].N, (int64_t)256);
  }
  g.first[nlayers] = nb;
  g.nlayers = nlayers; g.mode = mode; g.P = P; g.Pacc = Pacc; g.Gflat = Gflat; g.inv = inv; g.lr = lr;
  hipLaunchKernelGGL(k_chain_tail, dim3((unsigned)nb), dim3(256), 0, wd::as_stream(stream), g);
  return wd::check_launch("wd_chain_tail");
}

namespace {
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
}  // namespace

extern "C" int64_t wd_tower_chain_lds_bytes(int32_t K0, const int32_t *N, int32_t L, int32_t row_tile) {
  if (row_tile != 0 && row_tile != 32) return -1;
  return wd::chain8_lds_bytes(K0, N, L, 0);
}

static int check_chain_input(const wd_chain_input_t *in) {
  WD_REQUIRE(in->emb && in->slots && in->ids && in->x_out, "wd_chain_input_t: null pointer");
  WD_REQUIRE(in->dim >= 4 && in->dim % 4 == 0 && in->dim <= 256 && 256 % (in->dim / 4) == 0, "dim: multiple of 4 dividing 1024");
  WD_REQUIRE(in->S > 0 && in->S <= WD_CHAIN_MAX_SLOTS && in->ngroup > 0 && in->slot0 >= 0 && in->slot0 + in->ngroup <= in->S,
             "bad slot range (S <= WD_CHAIN_MAX_SLOTS)");
  WD_REQUIRE(in->ncols == 0 || (in->dense && in->cols), "numeric columns need dense + descriptors");
  WD_REQUIRE(!in->wide || in->wide_bias, "wide needs its bias");
  WD_REQUIRE(in->row_stride == 0 || (in->row_stride >= in->dim && in->row_stride % 4 == 0), "row_stride: 0 or >= dim, multiple of 4");
  WD_REQUIRE(!in->wide_in_row || in->row_stride > in->dim, "wide_in_row needs row_stride > dim");
  return WD_OK;
}

extern "C" int64_t wd_tower_chain_windows_lds_bytes(const wd_chain_windows_t *windows, int32_t K0, const int32_t *N, int32_t L) {
  if (!windows || !N) return -1;
  return wd::chain8_windows_lds_bytes(windows, K0, N, L);
}

extern "C" int64_t wd_tower_chain_blocks(int64_t batch, int32_t row_tile) {
  return wd::ceil_div(batch, (int64_t)(row_tile ? row_tile : 32));
}

extern "C" int wd_tower_chain(const float *x, int64_t ld_act, int32_t K0, const wd_chain_layer_t *layers, int32_t L,
                              int32_t act, float inv, const float *w_logits, const float *b_logits,
                              const float *wide_logit, const float *labels, const float *weights, int64_t batch,
                              float *dnn_logit, float *logit, float *prob, float *dlogit, float *loss_sum,
                              float *Gpart_logits, float *dx, int64_t ld_dx, int32_t dx_cols, const wd_chain_opts_t *opts,
                              wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(x && layers && w_logits && b_logits, "null pointer");
  WD_REQUIRE(L >= 1 && L <= MAXL, "1 <= L <= WD_CHAIN_MAX_LAYERS");
  WD_REQUIRE(ld_act % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "x must be 16-byte aligned with ld % 4 == 0");
  const int rt = 32;
  WD_REQUIRE(!opts || opts->row_tile == 0 || opts->row_tile == 32, "wd_chain_opts_t.row_tile must be 0 or 32");
  ChainArgs g{};
  int32_t N[MAXL];
  for (int l = 0; l < L; ++l) {
    g.layer[l] = layers[l];
    N[l] = layers[l].N;
    WD_REQUIRE(layers[l].Wpk && layers[l].a_out, "layer pointers");
    WD_REQUIRE(!layers[l].gamma == !layers[l].beta, "BN needs both gamma and beta");
    WD_REQUIRE((opts && opts->windows) || layers[l].K == (l == 0 ? K0 : layers[l - 1].N), "layer K must equal the previous width");
    if (labels) WD_REQUIRE(layers[l].dz_out && (l == 0 ? (!dx || layers[l].WTpk) : layers[l].WTpk != nullptr), "training needs dz_out / WTpk");
  }
  const wd_chain_windows_t *win = opts ? opts->windows : nullptr;
  if (win) {
    WD_REQUIRE(!opts->input && !opts->wide_vals, "windows: x and the wide logit are given (no fused input layer, no wide_vals)");
    WD_REQUIRE(wd::chain8_windows_lds_bytes(win, K0, N, L) > 0,
               "unsupported concatenating tower (widths / columns multiples of 32, sum of the hidden widths <= K0, k_logits <= 1024, "
               "the row tile must fit the LDS; see wd_tower_chain_windows_lds_bytes)");
    WD_REQUIRE(layers[0].K == K0, "windows: the first layer reads x");
    for (int l = 0; l < L; ++l)
      WD_REQUIRE(layers[l].K > 0 && layers[l].K % 8 == 0 && win->in_col[l] + layers[l].K <= win->cols, "windows: a layer's window lies inside the row");
  } else {
    WD_REQUIRE(wd::chain8_lds_bytes(K0, N, L, 0) > 0,
               "unsupported tower shape (widths must be multiples of 32 and the row tile must fit the LDS; see wd_tower_chain_lds_bytes)");
  }
  const int dxc = dx ? round_up(dx_cols, rt) : 0;
  WD_REQUIRE(dxc <= K0, "dx_cols must be <= K0");
  g.L = L; g.act = act; g.inv = inv; g.K0 = K0; g.dx_cols = dxc;
  g.train = labels != nullptr;
  g.x = x; g.ld_act = ld_act; g.w_logits = w_logits; g.b_logits = b_logits;
  g.wide_logit = wide_logit; g.labels = labels; g.weights = weights; g.batch = batch;
  g.dnn_logit = dnn_logit; g.logit = logit; g.prob = prob; g.dlogit = dlogit; g.loss_sum = loss_sum;
  g.Gpart_logits = Gpart_logits; g.dx = dx; g.ld_dx = ld_dx;
  g.flags_wt = wd::wt_mask() & WD_WT_TOWER ? 1 : 0;
  if (win) {
    g.win = 1; g.KL = win->k_logits; g.cols = win->cols;
    for (int l = 0; l <= L; ++l) { g.seg_col[l] = win->seg_col[l]; g.in_col[l] = win->in_col[l]; }
  }
  if (opts) {
    g.stamps = static_cast<unsigned long long *>(opts->stamps);
    g.tile_stamps = static_cast<unsigned long long *>(opts->tile_stamps);
    if (opts->flags & 4) g.flags_wt = 0;
    g.flags_prio = opts->flags & 8 ? 1 : 0;
    g.loss_part = opts->loss_part;
    if (opts->wide_vals) {
      WD_REQUIRE(!opts->input && opts->wide_bias && opts->wide_S > 0, "wide_vals: needs wide_bias, wide_S > 0 and no fused input");
      WD_REQUIRE(rt * opts->wide_S <= 1024, "wide_vals: row_tile x wide_S must be <= 1024");
      g.wv = opts->wide_vals; g.wv_bias = opts->wide_bias; g.wv_out = opts->wide_out; g.wv_S = opts->wide_S;
    }
    if (opts->input) {
      const int rc = check_chain_input(opts->input);
      if (rc != WD_OK) return rc;
      g.in = *opts->input;
      WD_REQUIRE((int64_t)rt * g.in.S * 4 <= (int64_t)N[0] * (rt + 1) * 4 && rt * g.in.S <= 1024,
                 "input fusion: row_tile x S ids must fit the scratch region ((row_tile + 1) x N_0 floats) and be <= 1024");
    }
  }
  const int rc = wd::chain8_launch(g, stream);
  WD_REQUIRE(rc <= 0, "unsupported call (16-byte aligned a_out / dz_out / dx with row strides % 4 == 0; at most 48 phases)");
  return rc;
}
