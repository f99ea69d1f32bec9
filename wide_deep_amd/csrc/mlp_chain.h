// wide_deep_amd/csrc/mlp_chain.h -- shared by the one-launch tower's entry point (mlp_chain.hip) and its kernel (mlp_chain8.hip):
// the launch arguments, the activation functions (python/lib/utils/model_util.py:28-55 activation_fn) and small helpers.
// Internal to csrc/.
#pragma once
#include "common.h"

namespace wd_chain {

constexpr int MAXL = WD_CHAIN_MAX_LAYERS;

struct ChainArgs {
  wd_chain_layer_t layer[MAXL];
  int32_t a_off[MAXL];   // LDS float offsets of a_l and dz_l
  int32_t dz_off[MAXL];
  int32_t tab_off[MAXL]; // LDS float offset of layer l's BN affine table: s_l[N_l] (= gamma inv) then t_l[N_l] (= beta)
  float inv;             // 1 / sqrt(1 + eps): the inference-mode BN of SURVEY App. C.1
  int32_t L;
  int32_t act;
  int32_t K0;            // width of x
  int32_t dx_cols;       // gradient columns of x wanted (multiple of 32, <= round32(K0)); 0: none
  int32_t train;
  const float *x;        // [batch][ld_act]
  int64_t ld_act;
  const float *w_logits; // [K_L]
  const float *b_logits; // [1]
  const float *wide_logit, *labels, *weights;
  int64_t batch;
  float *dnn_logit, *logit, *prob, *dlogit, *loss_sum, *Gpart_logits;
  float *dx;
  int64_t ld_dx;
  unsigned long long *stamps;   // diagnostics (wd_chain_opts_t.stamps): shader-clock stamps of workgroups 0 and 100
  int32_t flags_prio;                // wd_chain_opts_t.flags bit 3: s_setprio 3 for every wavefront of the launch
  int32_t flags_wt;                  // write-through stores of the HBM outputs (WD_WT bit 0; wd_chain_opts_t.flags bit 2 turns it off)
  unsigned long long *tile_stamps;   // wd_chain_opts_t.tile_stamps: [tile][2] realtime-clock stamps {start, x tile in LDS}
  wd_chain_input_t in;          // in.emb != NULL: the x tile is built here (input layer fused), wd_chain_opts_t.input
  float *loss_part;             // != NULL: this tile's loss is stored to loss_part[tile] (no atomic on loss_sum)
  const float *wv;              // wd_chain_opts_t.wide_vals: per-occurrence wide weights [batch][wv_S] (x from HBM)
  const float *wv_bias;
  float *wv_out;
  int32_t wv_S;
  // concatenating towers (wd_chain_windows_t): the row tile mirrors the activation row
  int32_t win;                  // 0: every layer reads its predecessor only
  int32_t x_off;                // LDS float offset of the x region (simple: 0)
  int32_t in_off[MAXL + 1];     // LDS float offset of layer l's input window ([L]: the logits layer)
  int32_t in_col[MAXL + 1];     // ... its first column (index into the column-wise affine tables)
  int32_t seg_col[MAXL + 1];    // first column of segment s (0: x, l + 1: hidden layer l's output)
  int32_t KL;                   // inputs of the logits layer
  int32_t sall_off, tall_off;   // LDS float offsets of the column-wise BN affine of the whole row (x: 1 / 0)
  int32_t cols;
};

__device__ __forceinline__ float act_fwd(float v, int act) {
  switch (act) {
    case WD_ACT_RELU: return fmaxf(v, 0.f);
    case WD_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case WD_ACT_TANH: return tanhf(v);
    case WD_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
    case WD_ACT_LEAKY_RELU: return v > 0.f ? v : 0.2f * v;
    case WD_ACT_ELU: return v > 0.f ? v : expm1f(v);
    case WD_ACT_SELU: return v > 0.f ? 1.0507009873554805f * v : 1.0507009873554805f * 1.6732632423543772f * expm1f(v);
    case WD_ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));
    case WD_ACT_SOFTSIGN: return v / (1.0f + fabsf(v));
    default: return v;
  }
}

__device__ __forceinline__ float act_bwd(float a, int act) {  // derivative through the activation OUTPUT a
  switch (act) {
    case WD_ACT_RELU: return a > 0.f ? 1.f : 0.f;
    case WD_ACT_SIGMOID: return a * (1.f - a);
    case WD_ACT_TANH: return 1.f - a * a;
    case WD_ACT_RELU6: return (a > 0.f && a < 6.f) ? 1.f : 0.f;
    case WD_ACT_LEAKY_RELU: return a > 0.f ? 1.f : 0.2f;
    case WD_ACT_ELU: return a > 0.f ? 1.f : a + 1.f;
    case WD_ACT_SELU: return a > 0.f ? 1.0507009873554805f : a + 1.0507009873554805f * 1.6732632423543772f;
    case WD_ACT_SOFTPLUS: return 1.f - expf(-a);
    case WD_ACT_SOFTSIGN: { float t = 1.f - fabsf(a); return t * t; }
    default: return 1.f;
  }
}

// Values that are the same in every lane but that the compiler cannot prove uniform (descriptor fields selected by a
// loop index): pin them to SGPRs, otherwise the reduction loops get exec-masked branches and vmcnt(0) joins.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }


}  // namespace wd_chain

namespace wd {
// the kernel (mlp_chain8.hip): LDS bytes of its layout (-1: the shape does not fit), and the launch (> 0: call not supported)
int64_t chain8_lds_bytes(int32_t K0, const int32_t *N, int32_t L, int32_t dx_cols);
int64_t chain8_windows_lds_bytes(const wd_chain_windows_t *w, int32_t K0, const int32_t *N, int32_t L);
int chain8_launch(const wd_chain::ChainArgs &g, wd_stream_t stream);
}  // namespace wd
