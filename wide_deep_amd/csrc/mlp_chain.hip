// wide_deep_amd/csrc/mlp_chain.hip -- the whole `simple` tower of one 32-example row tile in ONE launch (gfx950).
//
// Replaces, for connected_mode `simple` (python/lib/dnn.py:92-141: dense -> activation -> BN per layer, then the
// units=1 logits layer python/lib/dnn.py:226-232) joined with the head (python/lib/joint.py:216-222 add_n of the wide
// and deep logits, joint.py:264-269 sigmoid CE), the chain of launches
//     NN_0 .. NN_{L-1}, logits head, NT_{L-1} .. NT_0
// of mlp.hip by one kernel: a row tile's forward AND its input-gradient chain depend on nothing but that tile's rows
// (the loss is a batch SUM, BN is the inference affine -- applied as written, bn = gamma inv a + beta, to the A fragment as
// it is read: nothing is folded into the weights since round 2), so the tile stays in LDS from the input layer's output x
// down to the gradient dx that the embedding update consumes.  What remains outside are the
// weight-gradient products G_l = [a_{l-1} | 1]^T dz_l, which reduce over the whole batch (split-K GEMMs of mlp.hip);
// the kernel leaves a_l and dz_l in HBM for them.
//
// Why: at batch 8192 the tower is 9 GEMM launches of 8-19 us each, and ~11 us of every launch is fixed cost (ramp,
// cold L2 after the kernel boundary, epilogue; profiles/r1h_gemm_microbench.txt) -- the small layers are pure launch
// latency.  Per CU the tile's work is ~35 us of v_mfma_f32_32x32x2_f32 at full rate.
//
// Row tile RT = 32 (v_mfma_f32_32x32x2_f32, one workgroup per CU; the default) or RT = 16 (v_mfma_f32_16x16x4_f32, WD_CHAIN_RT=16:
// faster alone, slower inside the step where its second wavefront per SIMD is what the side branches need): at batch 8192 a 32-row tile gives exactly 256 workgroups = ONE wavefront per SIMD, and everything that is not
// an MFMA -- the gather of the x tile, the epilogues, the barriers between stages, the head -- leaves the matrix pipe idle
// (profiles/r1w_tower_chain_stage_cycles.txt: 177 k cycles per tile of which 76 k are MFMA issue).  A 16-row tile needs
// half the LDS (61 KB at C2), so TWO workgroups share a CU, drift out of phase, and one computes while the other gathers /
// stores / waits at a barrier.  Same FLOPs per instruction-cycle (16x16x4 issues every 32 cycles), twice the weight
// traffic from L2 (every tile streams all weights), half the rows per epilogue.
//
// The x tile comes from HBM (the input layer is its own launch, wd_prefetch_onehot, issued one step ahead -- round 3 -- or any
// other producer of x) or, wd_chain_opts_t.input, is gathered by the kernel itself (the rows that arrived through the exchange
// of the sharded engine; WD_INPUT_AHEAD=0 on one GPU).
//
// Data flow per workgroup (256 lanes = 4 wavefronts, one per SIMD):
//   * activations live in LDS reduction-major  [k][RT+1]  (RT examples + 1 pad): the A fragment of MFMA step k is the
//     row read lds[(k + lane/RT) * (RT+1) + lane%RT]; an accumulator (col = lane%RT per register row) is written back
//     transposed (2-way bank conflicts at most, free for ds_write_b32).
//   * weights are NOT staged in LDS: wavefront w owns the output columns [32w, 32w+32) (+128 ..), nobody else reads
//     them, so the B fragments are loaded straight from L2 into registers.  wd_chain_tail writes the kernel (and its
//     transpose, for the gradient chain) in MFMA-fragment order (wd_chain_layer_t.Wpk / WTpk) as it updates the parameter: ONE 16-byte load
//     per lane, 1 KB contiguous per wavefront, feeds four MFMA steps.  (One dword per MFMA -- the natural [K][N] layout
//     -- ran at 118 us: the CU's texture-address unit moves ~16 B/clk of dword loads, exactly what four wavefronts of
//     back-to-back fp32 MFMAs consume, so the waves sat in s_waitcnt 54 % of the time.)  A register ring keeps 7-11
//     groups (> 1 us of MFMAs) in flight.
//   * exact fp32 (v_mfma_f32_32x32x2_f32 == an fmaf chain), same numerics class as the per-layer GEMMs.
#include "mlp_chain.h"
#include <type_traits>

namespace {
using namespace wd_chain;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4v __attribute__((ext_vector_type(4)));

typedef float floatx4a __attribute__((ext_vector_type(4)));

#ifndef WD_CHAIN_RING
#define WD_CHAIN_RING 8
#endif
#ifndef WD_CHAIN_RING16
#define WD_CHAIN_RING16 4
#endif

// Geometry of a row tile.  One MFMA covers RT rows x RT columns x KS reduction rows; a "group" is the GK = 4 KS reduction
// rows that ONE 16-byte weight load per lane feeds (four MFMA steps).
template <int RT_> struct Tile;
template <> struct Tile<32> {
  static constexpr int RT = 32, P = 33, KS = 2, GK = 8, NACC = 16, RING = WD_CHAIN_RING, NTMAX = 2;
  typedef floatx16 acc_t;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
};
template <> struct Tile<16> {
  static constexpr int RT = 16, P = 17, KS = 4, GK = 16, NACC = 4, RING = WD_CHAIN_RING16, NTMAX = 4;
  typedef floatx4a acc_t;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row_of(int r, int h) { return r + 4 * h; }
};

// acc[t] += in[32 x K] . B[K x 32-column tile t]  for NT tiles `tstride` float4 apart in the packed operand.
// inA / WB already carry the lane's (lane%32, lane/32).
//
// Ring of D register groups (8 reduction rows each: one 16-byte weight load per tile + four A-fragment LDS reads feed
// 4 x NT MFMAs): while group c is multiplied, groups c+1 .. c+D-1 are in flight.  With ONE wavefront per SIMD nothing
// else hides latency, and everything that is not an MFMA has to issue in the shadow of one: region d = {prefetch of
// group c+D-1, MFMAs of group c} is left to the scheduler as a unit (the prefetch has no consumer inside it, so it is
// not sunk), a sched_barrier only separates regions.  (A barrier between the prefetch and the MFMAs exposed the ~16
// address / load instructions of every group while the matrix pipe idled: 110 instead of 64 cycles per MFMA.)
// Every prefetch is unconditional (past the end it re-reads the last group and is never used): a load inside a branch
// would make the wait at the join vmcnt(0), i.e. serialise the prefetch with the MFMAs it is meant to overlap.
// AFF: the input of the product is the BN affine of the stored activations, bn[k] = a[k] * sA[k] + tA[k] (python/lib/dnn.py:
// 113-114: tf.layers.batch_normalization without training=True = the inference affine), applied to the A
// fragment as it is read (two more LDS reads and a multiply-add per fragment element, shared by the NT tiles): the kernels
// stay the reference's raw variables -- nothing is folded, so the step needs no fold launch and no column sum for a
// folded bias -- and the LDS tile keeps a itself, which the gradient chain needs for act'.
template <typename TL, int NT, bool FULL, bool AFF>
__device__ __forceinline__ void mma_ring(const float *__restrict__ inA, const float4 *__restrict__ WB, int KG,
                                         int tstride, typename TL::acc_t (&acc)[NT], const float *__restrict__ sA,
                                         const float *__restrict__ tA) {
  constexpr int D = TL::RING, P = TL::P, GK = TL::GK, KS = TL::KS;
  float fa[D][4], fs[AFF ? D : 1][4], ft[AFF ? D : 1][4];
  float4 fb[D][NT];
#ifndef WD_CHAIN_EXP
#define WD_CHAIN_EXP 0      // diagnostics (scripts/build_chain_exp.sh): 1 = no MFMAs, 2 = weights loaded once, 3 = A fragments read once
#endif
  // the ring holds what was READ (a, and with AFF its s and t); the affine itself is computed where the fragment is used --
  // computed at load time it waits for its LDS reads inside the prefetch region (+40 cycles per MFMA in the narrow layers)
  auto load = [&](int buf, int c) {
    const int kg = c < KG ? c : KG - 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) fb[buf][t] = WB[(WD_CHAIN_EXP == 2 ? 0 : kg * 64) + t * tstride];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kr = WD_CHAIN_EXP == 3 ? 0 : (GK * kg + KS * j);
      fa[buf][j] = inA[kr * P];
      if (AFF) {
        fs[buf][j] = sA[kr];
        ft[buf][j] = tA[kr];
      }
    }
  };
  auto mfmas = [&](int buf) {
    float av[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) av[j] = AFF ? __fadd_rn(__fmul_rn(fa[buf][j], fs[AFF ? buf : 0][j]), ft[AFF ? buf : 0][j]) : fa[buf][j];
#if WD_CHAIN_EXP == 1
#pragma unroll
    for (int t = 0; t < NT; ++t)
      acc[t][0] += av[0] * fb[buf][t].x + av[1] * fb[buf][t].y + av[2] * fb[buf][t].z + av[3] * fb[buf][t].w;
    return;
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = TL::mfma(av[0], fb[buf][t].x, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = TL::mfma(av[1], fb[buf][t].y, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = TL::mfma(av[2], fb[buf][t].z, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = TL::mfma(av[3], fb[buf][t].w, acc[t]);
  };
#pragma unroll
  for (int i = 0; i < D - 1; ++i) load(i, i);
  __builtin_amdgcn_sched_barrier(0);
  for (int c0 = 0; c0 < KG; c0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      load((d + D - 1) % D, c0 + d + D - 1);
      if (FULL || c0 + d < KG) mfmas(d);   // FULL: KG % D == 0, the body is branch-free
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <typename TL, int NT>
__device__ __forceinline__ void mma_tiles(const float *__restrict__ inA, const float4 *__restrict__ WB, int KG,
                                          int tstride, typename TL::acc_t (&acc)[NT], const float *__restrict__ sA,
                                          const float *__restrict__ tA) {
  if (sA) {
    if (KG % TL::RING == 0) mma_ring<TL, NT, true, true>(inA, WB, KG, tstride, acc, sA, tA);
    else mma_ring<TL, NT, false, true>(inA, WB, KG, tstride, acc, sA, tA);
  } else {
    if (KG % TL::RING == 0) mma_ring<TL, NT, true, false>(inA, WB, KG, tstride, acc, sA, tA);
    else mma_ring<TL, NT, false, false>(inA, WB, KG, tstride, acc, sA, tA);
  }
}

// One product stage: out[RT x N] = epilogue(in[RT x K] . W[K x N]).  Wavefront w takes the RT-column tiles
// w, w+4, w+8, ... NTMAX at a time (shared A fragments, independent accumulator chains).
//   MODE 0 (forward):  a = act(acc + bias[n]) -> LDS out; HBM g_out[b][n] = bn = a * s_out[n] + t_out[n] (the operand of the
//                      next layer's weight-gradient product; s_out NULL: no BN, bn = a).  Input affine: s_in / t_in (or NULL).
//   MODE 1 (gradient): acc = d(bn) of the producing layer; per-tile column sums of acc (-> d beta) and acc * a (-> d gamma);
//                      v = acc * s_out[n] * act'(a) -> LDS out + HBM g_out[b][n] (dz), column sum of v (-> d bias)
//   MODE 2 (dx):       v = acc -> HBM g_out[b][n] only (n < n_store)
struct StageAff {
  const float *s_in, *t_in;     // LDS tables of the INPUT's BN affine (forward) or NULL
  const float *s_out, *t_out;   // LDS tables of the OUTPUT layer's BN affine or NULL
  float *db_out, *dg_out, *dbeta_out;   // MODE 1: this tile's column-sum partials [N] (HBM) or NULL
  // MODE 2 (optional): column n of example b belongs to occurrence b * sc_S + (n >> sc_shift) and goes to
  // sc_out[sc_pos[occurrence] * sc_RS + (n & (sc_dim - 1))] (sc_pos < 0: dropped) instead of g_out
  const int32_t *sc_pos;
  float *sc_out;
  int32_t sc_S, sc_RS, sc_dim, sc_shift;
  // MODE 2 (optional): LDS scratch of 4 x RT x P floats.  When the stage has 4 q + 1 column tiles -- dx at the Criteo shape:
  // 416 columns = 13 tiles, i.e. 4 + 3 + 3 + 3 over the four wavefronts: three of them idle for a quarter of the stage --
  // the LAST tile is split over the reduction instead: every wavefront multiplies a quarter of the k range, the partial tiles
  // meet in this scratch and are added in wavefront order (fixed summation order)
  float *split_scratch;
  // HBM outputs (activations, dz, dx) stored at device scope = written through this XCD's L2 while the kernel computes,
  // instead of staying dirty in it until the end-of-kernel release writes ~45 MB back (common.h; wd_chain_opts_t.flags bit 2: off)
  int32_t wt;
};
__device__ __forceinline__ void gstore(float *p, float v, bool wt) { wd::store1(p, v, wt); }
template <typename TL, int MODE>
__device__ __forceinline__ void stage(const float *__restrict__ in, int K, const float *__restrict__ Wpk, int N,
                                      const float *__restrict__ bias, int act,
                                      const float *__restrict__ a_prev, float *__restrict__ out,
                                      float *__restrict__ g_out, int64_t ld_g, int n_store, int64_t b0, int64_t batch,
                                      const StageAff &af, unsigned long long *dbg = nullptr) {
  constexpr int RT = TL::RT, P = TL::P;
  typedef typename TL::acc_t acc_t;
  const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
  const int c = lane % RT, h = lane / RT;
  K = uni(K); N = uni(N);
  const int KG = K / TL::GK;
  const int ntiles = N / RT;
  const float *inA = in + h * P + c;
  const float *sA = af.s_in ? af.s_in + h : nullptr, *tA = af.s_in ? af.t_in + h : nullptr;
  const bool full_rows = b0 + RT <= batch;   // uniform: no per-element row predicate in the common case
  auto epilogue_t = [&](const acc_t &acc, int n0, float bv, float sv, float tv, auto act_c, auto full_c) {
    constexpr int ACT = decltype(act_c)::value;   // >= 0: activation known at compile time
    constexpr bool FULLR = decltype(full_c)::value;
    const int n = n0 + c;
    const int a_id = ACT >= 0 ? ACT : act;
    float csum = 0.f, gsum = 0.f, bsum = 0.f;
#pragma unroll
    for (int r = 0; r < TL::NACC; ++r) {
      const int m = TL::row_of(r, h);
      float v = acc[r], o = 0.f;
      if (MODE == 0) {
        v = act_fwd(v + bv, a_id);
        o = __fadd_rn(__fmul_rn(v, sv), tv);          // bn: what the next layer (and its weight-gradient product) reads
      }
      if (MODE == 1) {
        const float a = a_prev[n * P + m];
        bsum += v;
        gsum += v * a;
        v = v * sv * act_bwd(a, a_id);
        o = v;
        csum += v;
      }
      if (MODE == 2) o = v;
      if (MODE != 2) out[n * P + m] = v;
      if (MODE == 2 && af.sc_pos) {
        if ((FULLR || b0 + m < batch) && n < af.sc_S * af.sc_dim) {
          const int32_t p = af.sc_pos[(b0 + m) * af.sc_S + (n >> af.sc_shift)];
          if (p >= 0) af.sc_out[(int64_t)p * af.sc_RS + (n & (af.sc_dim - 1))] = o;
        }
      } else if ((FULLR || b0 + m < batch) && n < n_store) gstore(&g_out[(b0 + m) * ld_g + n], o, af.wt);
    }
    if (MODE == 1) {   // this tile's partials of the bias / BN gradients: column sums over its RT rows (rows >= batch are 0)
#pragma unroll
      for (int off = RT; off < 64; off <<= 1) {
        csum += __shfl_xor(csum, off, 64);
        gsum += __shfl_xor(gsum, off, 64);
        bsum += __shfl_xor(bsum, off, 64);
      }
      if (h == 0) {
        if (af.db_out) af.db_out[n] = csum;
        if (af.dg_out) af.dg_out[n] = gsum;
        if (af.dbeta_out) af.dbeta_out[n] = bsum;
      }
    }
  };
  auto epilogue = [&](const acc_t &acc, int n0, float bv, float sv, float tv) {
    using std::integral_constant;
    if (act == WD_ACT_RELU || MODE == 2) {
      if (full_rows) epilogue_t(acc, n0, bv, sv, tv, integral_constant<int, WD_ACT_RELU>{}, integral_constant<bool, true>{});
      else epilogue_t(acc, n0, bv, sv, tv, integral_constant<int, WD_ACT_RELU>{}, integral_constant<bool, false>{});
    } else {
      epilogue_t(acc, n0, bv, sv, tv, integral_constant<int, -1>{}, integral_constant<bool, false>{});
    }
  };
  auto run = [&](int t0, auto nt_c) {
    constexpr int NT = decltype(nt_c)::value;
    const float4 *WB = reinterpret_cast<const float4 *>(Wpk) + (int64_t)t0 * KG * 64 + lane;
    acc_t acc[NT];
    float bv[NT], sv[NT], tv[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
      for (int r = 0; r < TL::NACC; ++r) acc[i][r] = 0.f;
      const int n = (t0 + 4 * i) * RT + c;
      bv[i] = (MODE == 0 && bias) ? bias[n] : 0.f;       // requested before the reduction loop, used after it
      sv[i] = (MODE != 2 && af.s_out) ? af.s_out[n] : 1.0f;
      tv[i] = (MODE == 0 && af.s_out) ? af.t_out[n] : 0.0f;
    }
    if (dbg && threadIdx.x == 0) *dbg++ = __builtin_readcyclecounter();
    mma_tiles<TL, NT>(inA, WB, KG, 4 * KG * 64, acc, sA, tA);
    if (dbg && threadIdx.x == 0) *dbg++ = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < NT; ++i) epilogue(acc[i], (t0 + 4 * i) * RT, bv[i], sv[i], tv[i]);
    if (dbg && threadIdx.x == 0) *dbg++ = __builtin_readcyclecounter();
  };
  using std::integral_constant;
  const bool split = MODE == 2 && TL::RT == 32 && af.split_scratch && (ntiles & 3) == 1 && ntiles > 4 && (KG & 3) == 0;
  const int ntl = split ? ntiles - 1 : ntiles;
  int t0 = wave;
  while (t0 < ntl) {
    const int left = (ntl - t0 + 3) / 4;   // tiles of this wavefront still to do
    if (TL::NTMAX >= 4 && left >= 4) { run(t0, integral_constant<int, (TL::NTMAX >= 4 ? 4 : 1)>{}); t0 += 16; }
    else if (TL::NTMAX >= 4 && left == 3) { run(t0, integral_constant<int, (TL::NTMAX >= 4 ? 3 : 1)>{}); t0 += 12; }
    else if (left >= 2) { run(t0, integral_constant<int, 2>{}); t0 += 8; }
    else { run(t0, integral_constant<int, 1>{}); t0 += 4; }
  }
  if (split) {
    // the last column tile, a quarter of the reduction per wavefront
    const int tl = ntiles - 1, kq = KG / 4;
    acc_t acc[1];
#pragma unroll
    for (int r = 0; r < TL::NACC; ++r) acc[0][r] = 0.f;
    const float4 *WB = reinterpret_cast<const float4 *>(Wpk) + ((int64_t)tl * KG + (int64_t)wave * kq) * 64 + lane;
    mma_tiles<TL, 1>(inA + (int64_t)wave * kq * TL::GK * P, WB, kq, 0, acc, nullptr, nullptr);
    float *scr = af.split_scratch + wave * (RT * P);
#pragma unroll
    for (int r = 0; r < TL::NACC; ++r) scr[c * P + TL::row_of(r, h)] = acc[0][r];
    __syncthreads();
    for (int i = threadIdx.x; i < RT * RT; i += 256) {
      const int m = i / RT, cc = i % RT;
      const float *p0 = af.split_scratch + cc * P + m;
      float v = p0[0];
      v += p0[RT * P];
      v += p0[2 * RT * P];
      v += p0[3 * RT * P];
      const int n = tl * RT + cc;
      if (!(full_rows || b0 + m < batch)) continue;
      if (af.sc_pos) {
        if (n < af.sc_S * af.sc_dim) {
          const int32_t pp = af.sc_pos[(b0 + m) * af.sc_S + (n >> af.sc_shift)];
          if (pp >= 0) af.sc_out[(int64_t)pp * af.sc_RS + (n & (af.sc_dim - 1))] = v;
        }
      } else if (n < n_store) {
        gstore(&g_out[(b0 + m) * ld_g + n], v, af.wt);
      }
    }
  }
}

template <int RT_>
__global__ void __launch_bounds__(256, RT_ == 16 ? 2 : 1) k_tower_chain(ChainArgs g) {
  typedef Tile<RT_> TL;
  constexpr int RT = TL::RT, P = TL::P;
  constexpr int PARTS = 256 / RT;   // lane groups of the head's dot product
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float sdl[RT];
  __shared__ float red[256], red2[256], red3[256];
  __shared__ float swl[512];   // logits-layer kernel
  __shared__ int64_t s_eoff[WD_CHAIN_MAX_SLOTS], s_rbase[WD_CHAIN_MAX_SLOTS];   // fused input layer: slot descriptors
  __shared__ int32_t s_ocol[WD_CHAIN_MAX_SLOTS];
  const int t = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * RT;
  const int L = uni(g.L);
  float *regx = lds;  // x, later the dz_l
  int nstamp = 0;
  auto stamp = [&]() {
    if (g.stamps && (blockIdx.x == 0 || blockIdx.x == 100) && t == 0)
      g.stamps[(blockIdx.x ? 32 : 0) + nstamp] = __builtin_readcyclecounter();
    ++nstamp;
  };
  stamp();
  if (g.tile_stamps && t == 0) g.tile_stamps[2 * blockIdx.x] = wall_clock64();
  // Two workgroups per CU (RT = 16) = two wavefronts per SIMD that start together and do the same work: left alone they stay in
  // lock-step (both gather, both multiply at half rate, both store).  The wavefront in the odd hardware slot gets priority: it
  // runs its matrix phases at full rate and pulls ahead, the other one fills its gaps.
  if (RT == 16 && g.prio_split && (__builtin_amdgcn_s_getreg((4) | (0 << 6) | (3 << 11)) & 1u)) __builtin_amdgcn_s_setprio(3);
  // (row tile 32, one wavefront per SIMD: priority over the wavefronts of the kernels that run beside the tower -- bucketing,
  // sort and gather of the next batch; wd_chain_opts_t.flags bit 3)
  if (g.flags_prio) __builtin_amdgcn_s_setprio(3);

  // ---- head inputs: requested now, consumed after the last hidden layer (no exposed latency there) -------------
  const int KL = uni(g.layer[L - 1].N);
  for (int n = t; n < KL; n += 256) swl[n] = g.w_logits[n];
  float h_wide = 0.f, h_y = 0.f, h_w = 1.0f, h_bias = 0.f;
  if (t < RT && b0 + t < g.batch) {
    if (g.wide_logit) h_wide = g.wide_logit[b0 + t];

    if (g.labels) h_y = g.labels[b0 + t];
    if (g.weights) h_w = g.weights[b0 + t];
  }
  if (t < RT) h_bias = g.b_logits[0];
  // BN affine tables of every hidden layer -> LDS: s = gamma * inv, t = beta (no BN: the identity, and the stages skip it)
  for (int l = 0; l < L; ++l) {
    const wd_chain_layer_t &ly = g.layer[l];
    float *tab = lds + g.tab_off[l];
    for (int n = t; n < ly.N; n += 256) {
      tab[n] = ly.gamma ? ly.gamma[n] * g.inv : 1.0f;
      tab[ly.N + n] = ly.beta ? ly.beta[n] : 0.0f;
    }
  }

  float *s_wide = red;   // [RT] wide logit of the tile's examples (gather mode); `red` is free until the head
  float wv[4] = {0.f, 0.f, 0.f, 0.f};   // gather mode: this lane's wide weights, in flight from the tile gather to the head
  if (g.wv) {
    // prefetched input layer: the tile's RT x S wide weights are one contiguous run of the per-occurrence list -- requested here,
    // coalesced, consumed in the head (registers -> LDS -> per-example sum, the code path of the fused gather)
    const int nbag = RT * g.wv_S;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = t + 256 * u;
      if (i < nbag && b0 * g.wv_S + i < g.batch * g.wv_S) wv[u] = g.wv[b0 * g.wv_S + i];
    }
  }
  if (g.in.emb) {
    // ---- input layer of the tile, fused (python/lib/dnn.py:88-90 input_layer + python/lib/linear.py:29-36 linear_model
    // for one-id-per-bag batches): x[m, out_col_s ..] = E_s[id(m, s)], numeric columns, wide logit.  RT x 26 random 64-byte
    // rows per workgroup (C2): ids first (coalesced), then every row / wide line load of the tile in flight at once.
    const wd_chain_input_t &I = g.in;
    const int S = uni(I.S), NG = uni(I.ngroup), D = uni(I.dim), LG = D >> 2;
    const int RS_ = uni(I.row_stride > 0 ? I.row_stride : I.dim);   // floats between rows (exchange buffers: dim + 4)
    int32_t *s_id = reinterpret_cast<int32_t *>(lds + g.a_off[0]);   // a_0's region is dead until layer 0 writes it
    for (int i = t; i < S; i += 256) {                               // slot descriptors -> LDS (48-byte structs in HBM)
      const wd_slot_t sl = I.slots[i];
      s_eoff[i] = sl.emb_off;
      s_ocol[i] = sl.out_col;
      s_rbase[i] = sl.wide ? (int64_t)sl.row_base : (int64_t)-1;
    }
    const int nbag = RT * S;
    for (int i = t; i < nbag; i += 256) {
      const int m = i / S;
      s_id[i] = b0 + m < g.batch ? I.ids[b0 * S + i] : -1;
    }
    for (int i = t; i < (int)g.K0 * P; i += 256) regx[i] = 0.f;     // pad columns and dropped ids read as zero
    __syncthreads();
    // Request order = the order things are needed in: the embedding rows and the numeric columns (the x tile: needed by the
    // first product), THEN the wide weights (one 16-byte line per occurrence, needed by the head three products later).  The
    // x tile is complete as soon as the rows have landed; the wide lines -- as many random requests again -- stay in flight
    // into the first product (registers wv, written to LDS just before the head), so the tile does not wait for them.
    constexpr int WV = 4, DV = 2;
    float dv[DV];
    const int lg = t % LG, grp = t / LG, ngrp = 256 / LG;
    const int nwork = RT * NG;
    // QB bags per lane group in flight: the whole tile in ONE round at the Criteo shape (RT x 26 bags / 64 lane groups = 13
    // at RT 32, 7 at RT 16)
    constexpr int QB = RT == 32 ? 16 : 8;
    bool first_round = true;
    for (int w0 = grp; w0 < nwork || first_round; w0 += QB * ngrp) {
      floatx4v r[QB];
      int col[QB], mm[QB];
      bool hit[QB];
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        const int w = w0 + q * ngrp;
        r[q] = floatx4v{0.f, 0.f, 0.f, 0.f};
        col[q] = -1;
        mm[q] = 0;
        hit[q] = false;
        if (w < nwork) {
          const int m = w / NG, sidx = I.slot0 + (w - m * NG);
          const int id = s_id[m * S + sidx];
          mm[q] = m;
          col[q] = s_ocol[sidx] + 4 * lg;
          hit[q] = id >= 0;
          if (id >= 0) r[q] = __builtin_nontemporal_load(reinterpret_cast<const floatx4v *>(I.emb + s_eoff[sidx] + (int64_t)id * RS_) + lg);
        }
      }
      if (first_round) {      // behind the first round of rows: numeric columns, then the wide lines
        first_round = false;
#pragma unroll
        for (int u = 0; u < DV; ++u) {
          const int i = t + 256 * u;
          dv[u] = 0.f;
          if (i < RT * I.ncols && b0 + i % RT < g.batch) dv[u] = I.dense[(b0 + i % RT) * I.ld_dense + i / RT];
        }
        if (I.wide) {
#pragma unroll
          for (int u = 0; u < WV; ++u) {
            const int i = t + 256 * u;
            wv[u] = 0.f;
            if (i < nbag) {
              const int64_t rb = s_rbase[i % S];
              const int id = s_id[i];
              if (rb >= 0 && id >= 0)
                wv[u] = I.wide_in_row ? I.wide[s_eoff[i % S] + (int64_t)id * RS_ + D]     // weight travels behind its row
                                      : __builtin_nontemporal_load(I.wide + (rb + id) * 4);
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        if (col[q] < 0) continue;
        const int c0 = col[q], m = mm[q];
        if (hit[q]) {
          regx[(c0 + 0) * P + m] = r[q].x; regx[(c0 + 1) * P + m] = r[q].y;
          regx[(c0 + 2) * P + m] = r[q].z; regx[(c0 + 3) * P + m] = r[q].w;
        }
        if (b0 + m < g.batch) *reinterpret_cast<floatx4v *>(I.x_out + (b0 + m) * g.ld_act + c0) = r[q];
      }
    }
    auto put_dense = [&](int i, float v) {
      const int m = i % RT, j = i / RT;
      const wd_dense_col_t c = I.cols[j];
      if (c.kind == 1) v = (v - c.p0) / (c.p1 - c.p0);
      else if (c.kind == 2) v = (v - c.p0) / c.p1;
      else if (c.kind == 3) v = logf(v);
      regx[c.out_col * P + m] = v;
      I.x_out[(b0 + m) * g.ld_act + c.out_col] = v;
    };
#pragma unroll
    for (int u = 0; u < DV; ++u) {
      const int i = t + 256 * u;
      if (i < RT * I.ncols && b0 + i % RT < g.batch) put_dense(i, dv[u]);
    }
    for (int i = t + 256 * DV; i < RT * I.ncols; i += 256)
      if (b0 + i % RT < g.batch) put_dense(i, I.dense[(b0 + i % RT) * I.ld_dense + i / RT]);
  } else {
  // ---- x tile -> LDS [k][P] (rows beyond the batch read as zero): all loads of a column block in flight, then the
  // transposing LDS stores ----------------------------------------------------------------------------------
  {
    constexpr int RH = RT / 16;           // 16-row passes per tile
    constexpr int KB = 64 * 16 / RH;      // columns covered by the 16 loads of a lane
    const int kq = t & 15, mr = t >> 4;   // 16 float4 (64 columns, 256 B) per example row, 16 rows per pass
    for (int kb = 0; kb < g.K0; kb += KB) {
      float4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = mr + 16 * (i % RH), k = kb + 64 * (i / RH) + 4 * kq;
        const int64_t row = b0 + m < g.batch ? b0 + m : g.batch - 1;   // clamped: the load is unconditional
        v[i] = *reinterpret_cast<const float4 *>(g.x + row * g.ld_act + (k < g.K0 ? k : 0));
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = mr + 16 * (i % RH), k = kb + 64 * (i / RH) + 4 * kq;
        if (k < g.K0) {
          const bool live = b0 + m < g.batch;
          regx[(k + 0) * P + m] = live ? v[i].x : 0.f;
          regx[(k + 1) * P + m] = live ? v[i].y : 0.f;
          regx[(k + 2) * P + m] = live ? v[i].z : 0.f;
          regx[(k + 3) * P + m] = live ? v[i].w : 0.f;
        }
      }
    }
  }
  }
  __syncthreads();
  stamp();
  if (g.tile_stamps && t == 0) g.tile_stamps[2 * blockIdx.x + 1] = wall_clock64();

  // ---- forward --------------------------------------------------------------------------------------------
  const float *in = regx;
  int K = g.K0;
  for (int l = 0; l < L; ++l) {
    const wd_chain_layer_t &ly = g.layer[l];
    float *out = lds + g.a_off[l];
    StageAff af{};
    af.wt = g.flags_wt;
    if (l > 0 && g.layer[l - 1].gamma) { af.s_in = lds + g.tab_off[l - 1]; af.t_in = af.s_in + g.layer[l - 1].N; }
    if (ly.gamma) { af.s_out = lds + g.tab_off[l]; af.t_out = af.s_out + ly.N; }
    stage<TL, 0>(in, K, ly.Wpk, ly.N, ly.bias, g.act, nullptr, out, ly.a_out, g.ld_act, ly.N, b0, g.batch, af,
                 (g.stamps && blockIdx.x == 0 && l == 0) ? g.stamps + 16 : nullptr);
    __syncthreads();
    stamp();
    in = out;
    K = ly.N;
  }

  // ---- logits layer + head (in = a_{L-1} [K][P]) ------------------------------------------------------------
  {
    const int m = t % RT, part = t / RT;
    const float *sL = lds + g.tab_off[L - 1], *tL = sL + K;     // affine of the last hidden layer (identity without BN)
    const bool wlist = (g.in.emb && g.in.wide) || g.wv;      // wide weights of the tile's occurrences in the wv registers
    if (wlist) {
      // the wide weights requested with the tile have long arrived: registers -> LDS (the x region is dead since the first
      // product), then the wide logit of each example, slots in order (fixed summation order)
      const int S = uni(g.wv ? g.wv_S : g.in.S), nbag = RT * S;
      const float *wbias = g.wv ? g.wv_bias : g.in.wide_bias;
      float *wout = g.wv ? g.wv_out : g.in.wide_out;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (t + 256 * u < nbag) regx[t + 256 * u] = wv[u];
      __syncthreads();
      if (t < RT) {
        float acc = 0.f;
        for (int sidx = 0; sidx < S; ++sidx) acc += regx[t * S + sidx];
        acc += wbias[0];
        s_wide[t] = acc;
        if (wout && b0 + t < g.batch) wout[b0 + t] = acc;
      }
    }
    float d = 0.f;
    for (int n = part; n < K; n += PARTS) d += __fadd_rn(__fmul_rn(in[n * P + m], sL[n]), tL[n]) * swl[n];
    const float h_wide_lds = (wlist && t < RT) ? s_wide[t] : 0.f;   // read before `red` is reused below
    __syncthreads();
    red[part * RT + m] = d;
    __syncthreads();
    if (t < RT) {
      float dn = 0.f;
#pragma unroll
      for (int p = 0; p < PARTS; ++p) dn += red[p * RT + t];
      dn += h_bias;
      const int64_t b = b0 + t;
      float dl = 0.f, ls = 0.f;
      if (b < g.batch) {
        const float x = dn + ((g.in.emb || g.wv) ? h_wide_lds : h_wide);
        const float y = h_y;
        const float w = h_w;
        const float e = expf(-fabsf(x));
        const float p = x >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
        dl = w * (p - y);
        ls = w * (fmaxf(x, 0.f) - x * y + log1pf(e));
        if (g.dnn_logit) g.dnn_logit[b] = dn;
        if (g.logit) g.logit[b] = x;
        if (g.prob) g.prob[b] = p;
        if (g.train && g.dlogit) g.dlogit[b] = dl;
        if (g.train && g.sc_pos)       // the example's dlogit behind the gradient row of each of its occurrences (wide part)
          for (int sidx = 0; sidx < g.sc_S; ++sidx) {
            const int32_t p = g.sc_pos[b * g.sc_S + sidx];
            if (p >= 0) g.sc_out[(int64_t)p * g.sc_RS + g.sc_dim] = dl;
          }
      }
      sdl[t] = dl;
      if (g.train && (g.loss_sum || g.loss_part)) {
        for (int off = RT / 2; off > 0; off >>= 1) ls += __shfl_down(ls, off, 64);
        if (t == 0) {
          if (g.loss_part) g.loss_part[blockIdx.x] = ls;
          else atomicAdd(g.loss_sum, ls);
        }
      }
    }
  }
  if (!g.train) return;
  __syncthreads();

  // d(bn_{L-1}) = dlogit w^T;  dz_{L-1} = d(bn) * s * act'(a_{L-1});  per-tile partials of d beta, d gamma, d bias of layer
  // L-1 and of the logits-layer kernel gradient (layout of wd_chain_tail: [tile][K + 1], the last entry = sum dlogit)
  {
    float *dz = lds + g.dz_off[L - 1];
    const wd_chain_layer_t &ll = g.layer[L - 1];
    float *gdz = ll.dz_out;
    const float *sL = lds + g.tab_off[L - 1], *tL = sL + K;
    float *Gp = g.Gpart_logits ? g.Gpart_logits + (int64_t)blockIdx.x * (K + 1) : nullptr;
    const int KT = K < 256 ? K : 256;   // lanes along n: coalesced HBM rows, conflict-free LDS
    const int MQ = 256 / KT;            // K < 256: several example groups in parallel
    const int nq = t % KT, mq = t / KT;
    float *dbp = ll.db_part ? ll.db_part + (int64_t)blockIdx.x * K : nullptr;
    float *dgp = ll.dgamma_part ? ll.dgamma_part + (int64_t)blockIdx.x * K : nullptr;
    float *dtp = ll.dbeta_part ? ll.dbeta_part + (int64_t)blockIdx.x * K : nullptr;
    if (mq < MQ) {
      for (int n = nq; n < K; n += KT) {
        const float w = swl[n], sv = sL[n];
        float csum = 0.f, gsum = 0.f, bsum = 0.f;
        for (int m = mq; m < RT; m += MQ) {
          const float a = in[n * P + m];
          const float dbn = sdl[m] * w;
          bsum += dbn;
          gsum += dbn * a;
          const float v = dbn * sv * act_bwd(a, g.act);
          dz[n * P + m] = v;
          if (b0 + m < g.batch) gdz[(b0 + m) * K + n] = v;
          csum += v;
        }
        if (MQ == 1) {
          if (dbp) dbp[n] = csum;
          if (dgp) dgp[n] = gsum;
          if (dtp) dtp[n] = bsum;
        } else {     // K < 256: MQ * KT == 256 partial sums each, combined below in group order
          red[mq * KT + n] = csum;
          red2[mq * KT + n] = gsum;
          red3[mq * KT + n] = bsum;
        }
      }
    }
    if (MQ > 1) {
      __syncthreads();
      if (t < K) {
        float v = red[t], u = red2[t], x = red3[t];
        for (int j = 1; j < MQ; ++j) {
          v += red[j * KT + t];
          u += red2[j * KT + t];
          x += red3[j * KT + t];
        }
        if (dbp) dbp[t] = v;
        if (dgp) dgp[t] = u;
        if (dtp) dtp[t] = x;
      }
    }
    if (Gp) {     // logits kernel gradient: its input is bn_{L-1}
      for (int n = t; n < K; n += 256) {
        float gw = 0.f;
#pragma unroll 8
        for (int m = 0; m < RT; ++m) gw += __fadd_rn(__fmul_rn(in[n * P + m], sL[n]), tL[n]) * sdl[m];
        Gp[n] = gw;
      }
      if (t == 0) {
        float v = 0.f;
        for (int m = 0; m < RT; ++m) v += sdl[m];
        Gp[K] = v;
      }
    }
  }
  __syncthreads();
  stamp();

  // ---- gradient chain: d(bn_{l-1}) = dz_l W_l^T, dz_{l-1} = d(bn_{l-1}) * s_{l-1} * act'(a_{l-1});  dx = dz_0 W_0^T -----
  for (int l = L - 1; l >= 1; --l) {
    const wd_chain_layer_t &ly = g.layer[l];
    const wd_chain_layer_t &lp = g.layer[l - 1];
    StageAff af{};
    af.wt = g.flags_wt;
    if (lp.gamma) { af.s_out = lds + g.tab_off[l - 1]; af.t_out = af.s_out + lp.N; }
    af.db_out = lp.db_part ? lp.db_part + (int64_t)blockIdx.x * lp.N : nullptr;
    af.dg_out = lp.dgamma_part ? lp.dgamma_part + (int64_t)blockIdx.x * lp.N : nullptr;
    af.dbeta_out = lp.dbeta_part ? lp.dbeta_part + (int64_t)blockIdx.x * lp.N : nullptr;
    stage<TL, 1>(lds + g.dz_off[l], ly.N, ly.WTpk, lp.N, nullptr, g.act, lds + g.a_off[l - 1], lds + g.dz_off[l - 1],
                 lp.dz_out, lp.N, lp.N, b0, g.batch, af);
    __syncthreads();
    stamp();
  }
  if (g.dx && g.dx_cols > 0) {
    const wd_chain_layer_t &ly = g.layer[0];
    StageAff af{};
    af.wt = g.flags_wt;
    af.sc_pos = g.sc_pos; af.sc_out = g.sc_out; af.sc_S = g.sc_S; af.sc_RS = g.sc_RS; af.sc_dim = g.sc_dim; af.sc_shift = g.sc_shift;
    // (a_0's LDS region is dead since the gradient stage of layer 1 read it for act')
    if (ly.N >= 4 * RT && !(g.flags_nosplit)) af.split_scratch = lds + g.a_off[0];
    stage<TL, 2>(lds + g.dz_off[0], ly.N, ly.WTpk, g.dx_cols, nullptr, 0, nullptr, nullptr, g.dx, g.ld_dx, g.K0, b0,
                 g.batch, af, (g.stamps && blockIdx.x == 0) ? g.stamps + 20 : nullptr);
  }
  stamp();
}


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
inline bool tile_ok(int rt) { return rt == 16 || rt == 32; }

// LDS layout: [x | dz_{L-1} .. dz_0 aliasing x] [a_0] [a_1] ...     (rows of P = row_tile + 1 floats)
int64_t chain_layout(int32_t K0, const int32_t *N, int32_t L, int32_t rt, int32_t *a_off, int32_t *dz_off,
                     int32_t *tab_off = nullptr) {
  if (!tile_ok(rt) || L < 1 || L > MAXL || K0 <= 0 || K0 % rt) return -1;
  const int64_t P = rt + 1;
  int64_t sum_n = 0;
  for (int l = 0; l < L; ++l) {
    if (N[l] <= 0 || N[l] % rt || N[l] > 512) return -1;
    sum_n += N[l];
  }
  const int64_t regx = (K0 > sum_n ? K0 : sum_n) * P;
  int64_t off = 0;
  for (int l = L - 1; l >= 0; --l) {
    if (dz_off) dz_off[l] = (int32_t)off;
    off += (int64_t)N[l] * P;
  }
  off = regx;
  for (int l = 0; l < L; ++l) {
    if (a_off) a_off[l] = (int32_t)off;
    off += (int64_t)N[l] * P;
  }
  for (int l = 0; l < L; ++l) {       // BN affine tables: s_l[N_l], t_l[N_l]
    if (tab_off) tab_off[l] = (int32_t)off;
    off += 2 * (int64_t)N[l];
  }
  const int64_t bytes = off * 4;
  return bytes <= 150 * 1024 ? bytes : -1;
}

}  // namespace

extern "C" int64_t wd_tower_chain_lds_bytes(int32_t K0, const int32_t *N, int32_t L, int32_t row_tile) {
  return chain_layout(K0, N, L, row_tile ? row_tile : 32, nullptr, nullptr);
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

extern "C" int64_t wd_tower_chain_blocks(int64_t batch, int32_t row_tile) {
  return wd::ceil_div(batch, (int64_t)(row_tile ? row_tile : 32));
}

template <int RT_>
static int launch_chain(const ChainArgs &g, int64_t bytes, wd_stream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_tower_chain<RT_>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) {
      wd::set_error("wd_tower_chain: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return WD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(k_tower_chain<RT_>, dim3((unsigned)wd::ceil_div(g.batch, (int64_t)RT_)), dim3(256), (size_t)bytes,
                     wd::as_stream(stream), g);
  return wd::check_launch("wd_tower_chain");
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
  const int rt = (opts && opts->row_tile) ? opts->row_tile : 32;
  WD_REQUIRE(tile_ok(rt), "wd_chain_opts_t.row_tile must be 0 (= 32), 16 or 32");
  ChainArgs g{};
  int32_t N[MAXL];
  for (int l = 0; l < L; ++l) {
    g.layer[l] = layers[l];
    N[l] = layers[l].N;
    WD_REQUIRE(layers[l].Wpk && layers[l].a_out, "layer pointers");
    WD_REQUIRE(!layers[l].gamma == !layers[l].beta, "BN needs both gamma and beta");
    WD_REQUIRE(layers[l].K == (l == 0 ? K0 : layers[l - 1].N), "layer K must equal the previous width");
    if (labels) WD_REQUIRE(layers[l].dz_out && (l == 0 ? (!dx || layers[l].WTpk) : layers[l].WTpk != nullptr), "training needs dz_out / WTpk");
  }
  const int64_t bytes = chain_layout(K0, N, L, rt, g.a_off, g.dz_off, g.tab_off);
  WD_REQUIRE(bytes > 0, "unsupported tower shape (widths must be multiples of the row tile and fit the LDS; see wd_tower_chain_lds_bytes)");
  const int dxc = dx ? round_up(dx_cols, rt) : 0;
  WD_REQUIRE(dxc <= K0, "dx_cols must be <= K0");
  g.L = L; g.act = act; g.inv = inv; g.K0 = K0; g.dx_cols = dxc;
  g.train = labels != nullptr;
  g.x = x; g.ld_act = ld_act; g.w_logits = w_logits; g.b_logits = b_logits;
  g.wide_logit = wide_logit; g.labels = labels; g.weights = weights; g.batch = batch;
  g.dnn_logit = dnn_logit; g.logit = logit; g.prob = prob; g.dlogit = dlogit; g.loss_sum = loss_sum;
  g.Gpart_logits = Gpart_logits; g.dx = dx; g.ld_dx = ld_dx;
  g.flags_wt = wd::wt_mask() & WD_WT_TOWER ? 1 : 0;
  if (opts) {
    g.stamps = static_cast<unsigned long long *>(opts->stamps);
    g.tile_stamps = static_cast<unsigned long long *>(opts->tile_stamps);
    g.prio_split = opts->flags & 1 ? 0 : 1;
    g.flags_nosplit = opts->flags & 2 ? 1 : 0;
    if (opts->flags & 4) g.flags_wt = 0;
    g.flags_prio = opts->flags & 8 ? 1 : 0;
    g.loss_part = opts->loss_part;
    if (opts->dx_pos) {
      WD_REQUIRE(opts->dx_scatter && opts->dx_S > 0 && opts->dx_rs > opts->dx_dim && opts->dx_dim >= 4 &&
                 (opts->dx_dim & (opts->dx_dim - 1)) == 0 && opts->dx_S * opts->dx_dim <= K0,
                 "dx scatter: power-of-two dim, records of more than dim floats, S * dim columns of x");
      g.sc_pos = opts->dx_pos; g.sc_out = opts->dx_scatter; g.sc_S = opts->dx_S; g.sc_RS = opts->dx_rs; g.sc_dim = opts->dx_dim;
      g.sc_shift = 0;
      while ((1 << g.sc_shift) < opts->dx_dim) ++g.sc_shift;
    }
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
  // two wavefronts per SIMD (mlp_chain8.hip) wherever that kernel takes the call; WD_CHAIN_WAVES=4 / flags bit 4: this file's
  static const bool waves8 = !(getenv("WD_CHAIN_WAVES") && atoi(getenv("WD_CHAIN_WAVES")) == 4);
  if (rt == 32 && waves8 && !g.sc_pos && !(opts && (opts->flags & 16))) {
    const int rc = wd::chain8_launch(g, stream);
    if (rc <= 0) return rc;
  }
  return rt == 16 ? launch_chain<16>(g, bytes, stream) : launch_chain<32>(g, bytes, stream);
}
