// wide_deep_amd/csrc/mlp_chain8.hip -- the one-launch `simple` tower (see mlp_chain.hip for what it computes: python/lib/dnn.py:92-141
// dense -> activation -> BN per layer, dnn.py:226-232 logits, python/lib/joint.py:216-222, 264-269 head, and the input-gradient
// chain) with TWO wavefronts per SIMD: 512 lanes per 32-example row tile, same LDS tile, same packed kernels, same outputs.
//
// Why (round 4).  The round-1..3 kernel runs ONE wavefront per SIMD; profiles/r2c_tower_ablation.txt + r3_tower_chain_stage_cycles.txt:
// its launch is the SUM of its MFMA time (78 k cycles per tile) and of everything else (94 k with the MFMAs compiled out) in
// every stage but the widest -- a stage whose wavefronts own one column tile each has 7 KB of kernel weights in flight per
// wavefront against ~2 k cycles of loaded L2 latency, the first weights of a stage are requested after the previous stage's
// epilogue and barrier, and a __syncthreads() drains the write-through stores of that epilogue (vmcnt(0)) before anybody moves on.
// Here:
//   * 8 wavefronts; a stage's column tiles are dealt one per wavefront, and what does not divide is split over the reduction:
//     a phase of 2^j tiles gives every tile to 8 / 2^j wavefronts, each multiplying a slice of the k range; the partial tiles meet
//     in an LDS scratch (register-major: no transposition, no bank conflicts) and the tile's first wavefront adds them in slice
//     order (fixed summation order) and runs the epilogue.  The C2 tower: F0 8 tiles x 1, F1 4 x 2, F2 2 x 4, B2 4 x 2, B1 8 x 1,
//     dx 8 x 1 + 4 x 2 + 1 x 8 -- every wavefront gets the same number of MFMAs in every stage;
//   * the kernel weights are ONE continuous stream per wavefront through a ring of 8 register groups: while a unit's last
//     groups are multiplied the ring already fills with the NEXT unit's first ones (next phase or next stage -- also across the
//     head), so no stage starts with an empty ring and the requests are older than the epilogue's stores;
//   * the A fragments (LDS) are double-buffered separately from the weight ring (an LDS read needs ~100 cycles, not a ring of 8);
//   * barriers are `s_waitcnt lgkmcnt(0); s_barrier`: the LDS tile must be visible, the HBM outputs need not have landed.
#include "mlp_chain.h"
#include <type_traits>

namespace {
using namespace wd_chain;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4v __attribute__((ext_vector_type(4)));

constexpr int NWAVE = 8, NTHR = 512, RT = 32, P = 33, DW = 8, GK = 8, KS = 2, NACC = 16;
constexpr int MAXPH = 48;      // phases of one launch
constexpr int SCR_SLOTS = 7;   // partial tiles of a phase that are not the owner's: (slices - 1) x tiles <= 7
constexpr int SLOT = NACC * 64;
// dynamic LDS a launch may ask for: 160 KB per workgroup minus the kernel's static arrays (12,928 B, scripts/kernel_resources.py)
constexpr int64_t DYN_LDS_MAX = 147 * 1024;

struct Phase8 {
  const float *Wpk;   // packed operand of the stage (wd_chain_layer_t.Wpk / WTpk)
  int32_t KG;         // reduction groups (8 rows each) of the whole stage
  int32_t tile0;      // first column tile of the phase
  int32_t ntp_log2;   // the phase covers 1 << ntp_log2 column tiles ...
  int32_t ns;         // ... each split into ns slices of the reduction
  int32_t KGs;        // groups per slice
};

struct Args8 {
  ChainArgs g;
  Phase8 ph[MAXPH];
  int32_t ph_first[2 * MAXL + 2];   // stage s = phases [ph_first[s], ph_first[s + 1]); stages: F_0 .. F_{L-1}, B_{L-1} .. B_1, dx
  int32_t nph;                      // phases this launch runs (forward only: the forward stages')
  // per stage: LDS float offsets of the partial-tile scratch (SCR_SLOTS x SLOT floats) and of a second one to alternate with
  // (-1: none) -- both carved out of regions of the tile that are dead while the stage runs (the kernel's LDS stays what the
  // activations need: ~30 KB of the CU are left to the kernels that run beside the tower)
  int32_t scr_off[2 * MAXL + 1];
  int32_t alt_off[2 * MAXL + 1];
};

__device__ __forceinline__ int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ floatx16 mfma(float a, float b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// the LDS tile must be visible to the other wavefronts; stores to HBM and weight loads in flight stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// packed kernels: pointers into GLOBAL memory, said so (a pointer rebuilt from two readfirstlanes is generic to the compiler: flat
// loads, whose waits are vmcnt(0) lgkmcnt(0) -- the ring would drain at every group)
typedef const __attribute__((address_space(1))) floatx4v *gw_t;
__device__ __forceinline__ gw_t uni_gw(const float4 *p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (gw_t)(((uint64_t)hi << 32) | lo);
}

struct Unit {
  gw_t W;            // first group of the unit (uniform); NULL / KG 0: this wavefront has nothing to multiply in the phase
  int KG, tile, slice;
};
__device__ __forceinline__ Unit unit_of(const Phase8 &p, int wave) {
  Unit u;
  const int l2 = uni(p.ntp_log2);
  u.tile = uni(p.tile0) + (wave & ((1 << l2) - 1));
  u.slice = wave >> l2;
  const int KGs = uni(p.KGs);
  u.KG = u.slice < uni(p.ns) ? KGs : 0;
  u.W = uni_gw(reinterpret_cast<const float4 *>(p.Wpk) + ((int64_t)u.tile * uni(p.KG) + (int64_t)u.slice * KGs) * 64);
  return u;
}

// groups 0 .. DW-2 of a unit into the ring (past its end: the last group again, never used)
__device__ __forceinline__ void prefill(floatx4v (&fb)[DW], gw_t Wu, int KG, int lane) {
#pragma unroll
  for (int i = 0; i < DW - 1; ++i) fb[i] = (Wu + (int64_t)(i < KG ? i : KG - 1) * 64)[lane];
}

// every request of the ring has landed: the compiler is shown a use of all its registers (it inserts the one wait, and from here
// on knows them complete -- no vmcnt wait in the next unit's first DW - 1 regions, whatever stores are issued in between)
__device__ __forceinline__ void ring_complete(floatx4v (&fb)[DW]) {
  asm volatile("; ring complete" : : "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fb[4]), "v"(fb[5]), "v"(fb[6]));
}

// acc += in[32 x 8 KG] . B[8 KG x 32]: one unit = one column tile over (a slice of) the reduction.
// On entry the ring holds the unit's groups 0 .. DW-2 in slots 0 .. DW-2; on exit it holds the NEXT unit's (Wn, KGn).
//   KIND 0  KG % DW == 0: the stream never stops -- while the last DW-1 groups are multiplied their slots are refilled with the
//           next unit's first groups, which land in exactly the slots that unit expects
//   KIND 1  KG < DW: everything is in the ring already; the next unit's groups are requested behind the last MFMA
//   KIND 2  other lengths: prefetch clamped to the unit, then as KIND 1
// Every load is unconditional (a load inside a branch makes the wait at the join vmcnt(0)).  Region d = {weights of group
// c + DW - 1, A fragments of group c + 1, the four MFMAs of group c} is left to the scheduler as a unit.
// AFF: the input of the product is the BN affine of the stored activations (python/lib/dnn.py:113-114: batch_normalization
// without training=True = the inference affine), bn[k] = a[k] * sA[k] + tA[k], applied to the fragment.
// SWAP: the kernel fragment is the MFMA's A operand and the activations its B operand -- the tile comes out transposed, lane =
// EXAMPLE, registers = 16 output columns in four runs of four: the epilogue stores 16 bytes per lane (see epilogue8t).
template <bool AFF, int KIND, bool SWAP>
__device__ __forceinline__ void mma_unit(const float *__restrict__ inA, gw_t Wl, int KG, gw_t Wn, int KGn, floatx16 &acc, floatx4v (&fb)[DW],
                                         const float *__restrict__ sA, const float *__restrict__ tA, int lane) {
  // Wl / Wn: UNIFORM pointers (scalar registers); the lane offset is added at the load
#ifndef WD_CHAIN8_EXP
#define WD_CHAIN8_EXP 0   // diagnostics (profiles/run_scripts/build_chain8_exp.sh), bits: 1 no MFMAs, 2 no weight loads in the loop, 4 A fragments
#endif                    // read once, 8 no HBM stores in the epilogues, 16 two accumulators alternate (no dependent MFMA chain), 32 no ring_complete
  float fa[2][4], fs[2][4], ft[2][4];
  floatx16 acc2;
  if (WD_CHAIN8_EXP & 16) {
#pragma unroll
    for (int r = 0; r < NACC; ++r) acc2[r] = 0.f;
  }
  auto loadA = [&](int buf, int c) {
    if ((WD_CHAIN8_EXP & 4) && c > 0) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kr = GK * c + KS * j;
      fa[buf][j] = inA[kr * P];
      if (AFF) {
        fs[buf][j] = sA[kr];
        ft[buf][j] = tA[kr];
      }
    }
  };
  auto mfmas = [&](int buf, const floatx4v &b) {
    float av[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) av[j] = AFF ? __fadd_rn(__fmul_rn(fa[buf][j], fs[buf][j]), ft[buf][j]) : fa[buf][j];
    if (WD_CHAIN8_EXP & 1) {
      acc[0] += av[0] * b.x + av[1] * b.y + av[2] * b.z + av[3] * b.w;
    } else if (WD_CHAIN8_EXP & 16) {
      acc = mfma(av[0], b.x, acc);
      acc2 = mfma(av[1], b.y, acc2);
      acc = mfma(av[2], b.z, acc);
      acc2 = mfma(av[3], b.w, acc2);
    } else if (SWAP) {
      acc = mfma(b.x, av[0], acc);
      acc = mfma(b.y, av[1], acc);
      acc = mfma(b.z, av[2], acc);
      acc = mfma(b.w, av[3], acc);
    } else {
      acc = mfma(av[0], b.x, acc);
      acc = mfma(av[1], b.y, acc);
      acc = mfma(av[2], b.z, acc);
      acc = mfma(av[3], b.w, acc);
    }
  };
  auto region = [&](int c0, auto d_c) {
    constexpr int d = decltype(d_c)::value;
    const int c = c0 + d;
    if (KIND == 0 && !(WD_CHAIN8_EXP & 2)) {
      const int p = c + DW - 1;
      const int q = p - KG < KGn ? p - KG : KGn - 1;
      gw_t src = p < KG ? Wl + (int64_t)p * 64 : Wn + (int64_t)q * 64;
      fb[(d + DW - 1) % DW] = src[lane];
    } else if (KIND == 2 && !(WD_CHAIN8_EXP & 2)) {
      const int p = c + DW - 1 < KG ? c + DW - 1 : KG - 1;
      fb[(d + DW - 1) % DW] = (Wl + (int64_t)p * 64)[lane];
    }
    loadA((d + 1) & 1, c + 1 < KG ? c + 1 : KG - 1);
    mfmas(d & 1, fb[d]);
    __builtin_amdgcn_sched_barrier(0);
  };
  using std::integral_constant;
#define WD_REGIONS(c0)                                                                   \
  do {                                                                                   \
    if (KIND != 0 && (c0) + 0 >= KG) break; region((c0), integral_constant<int, 0>{});   \
    if (KIND != 0 && (c0) + 1 >= KG) break; region((c0), integral_constant<int, 1>{});   \
    if (KIND != 0 && (c0) + 2 >= KG) break; region((c0), integral_constant<int, 2>{});   \
    if (KIND != 0 && (c0) + 3 >= KG) break; region((c0), integral_constant<int, 3>{});   \
    if (KIND != 0 && (c0) + 4 >= KG) break; region((c0), integral_constant<int, 4>{});   \
    if (KIND != 0 && (c0) + 5 >= KG) break; region((c0), integral_constant<int, 5>{});   \
    if (KIND != 0 && (c0) + 6 >= KG) break; region((c0), integral_constant<int, 6>{});   \
    if (KIND != 0 && (c0) + 7 >= KG) break; region((c0), integral_constant<int, 7>{});   \
  } while (0)
  static_assert(DW == 8, "WD_REGIONS lists the ring's slots");
  loadA(0, 0);
  // The first DW groups are straight-line code, NOT part of the loop: on entry the ring is complete (ring_complete() behind the
  // previous unit), so these regions carry no vmcnt wait at all -- inside the loop the compiler has to emit the steady-state
  // vmcnt(DW - 2), and a counter that retires in order makes that wait for the previous epilogue's stores (~3 k cycles each
  // stage, profiles/r4_tower8_ablation.txt: the stores' acknowledgements, not the weights, were what the ring waited for).
  WD_REGIONS(0);
  for (int c0 = DW; c0 < KG; c0 += DW) WD_REGIONS(c0);
#undef WD_REGIONS
  if (KIND != 0 || (WD_CHAIN8_EXP & 2)) prefill(fb, Wn, KGn, lane);
  if (WD_CHAIN8_EXP & 16) {
#pragma unroll
    for (int r = 0; r < NACC; ++r) acc[r] += acc2[r];
  }
}

// Epilogue of one finished 32 x 32 tile (lane = (column c, row half h), register r = row row_of(r, h)):
//   MODE 0 (forward):  a = act(acc + bias[n]) -> LDS out; HBM g_out[b][n] = bn = a * s_out[n] + t_out[n] (the operand of the
//                      next layer's weight-gradient product; no BN: bn = a)
//   MODE 1 (gradient): acc = d(bn) of the producing layer; per-tile column sums of acc (-> d beta) and acc * a (-> d gamma);
//                      v = acc * s_out[n] * act'(a) -> LDS out + HBM g_out[b][n] (dz), column sum of v (-> d bias)
//   MODE 2 (dx):       v = acc -> HBM g_out[b][n] only (n < n_store)
struct Aff8 {
  const float *s_in, *t_in;     // LDS tables of the INPUT's BN affine (forward) or NULL
  const float *s_out, *t_out;   // LDS tables of the OUTPUT layer's BN affine or NULL
  float *db_out, *dg_out, *dbeta_out;   // MODE 1: this row tile's column-sum partials [N] (HBM) or NULL
  int32_t wt;                   // HBM outputs stored at device scope (write-through), common.h
  // concatenating towers, MODE 1 / 2: the logits layer reads this segment too -- acc[m][n] += dl[m] * wl[n] (dl: the tile's
  // dlogit, wl: the segment's logits weights); NULL: the segment has one consumer (simple towers)
  const float *dl, *wl;
};

// MODE 1: the wavefront's own tile, as its epilogue left it in LDS, row by row to HBM -- 8 lanes x 16 bytes per example row
// (the whole 128-byte line of the row), 8 rows per pass, 4 store instructions instead of 16
__device__ __forceinline__ void tile_rows_to_hbm(const float *__restrict__ out, int n0, float *__restrict__ g_out, int64_t ld_g,
                                                 int64_t b0, int64_t batch, int wt, int lane) {
  const int nq = lane & 7, mr = lane >> 3;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int m = 8 * pass + mr;
    const float *src = out + (n0 + 4 * nq) * P + m;
    const float4 v = make_float4(src[0], src[P], src[2 * P], src[3 * P]);
    if (!(WD_CHAIN8_EXP & 8) && b0 + m < batch) wd::store4(reinterpret_cast<float4 *>(g_out + (b0 + m) * ld_g + n0 + 4 * nq), v, wt);
  }
}

template <int MODE, int ACT, bool FULLR>
__device__ __forceinline__ void epilogue8(const floatx16 &acc, int n0, float bv, float sv, float tv, int act,
                                          const float *__restrict__ a_prev, float *__restrict__ out, float *__restrict__ g_out,
                                          int64_t ld_g, int n_store, int64_t b0, int64_t batch, const Aff8 &af, int c, int h) {
  const int n = n0 + c;
  const int a_id = ACT >= 0 ? ACT : act;
  float csum = 0.f, gsum = 0.f, bsum = 0.f;
#pragma unroll
  for (int r = 0; r < NACC; ++r) {
    const int m = row_of(r, h);
    float v = acc[r], o = 0.f;
    if (MODE == 0) {
      v = act_fwd(v + bv, a_id);
      o = __fadd_rn(__fmul_rn(v, sv), tv);
    }
    if (MODE == 1) {
      const float a = a_prev[n * P + m];
      if (af.wl) v += af.dl[m] * af.wl[n];
      bsum += v;
      gsum += v * a;
      v = v * sv * act_bwd(a, a_id);
      o = v;
      csum += v;
    }
    if (MODE == 2) o = v;
    if (MODE != 2) out[n * P + m] = v;
    if (MODE != 1 && !(WD_CHAIN8_EXP & 8) && (FULLR || b0 + m < batch) && n < n_store) wd::store1(&g_out[(b0 + m) * ld_g + n], o, af.wt);
  }
  if (MODE == 1) {   // column sums over the tile's 32 rows (rows >= batch are 0): the two row halves meet by one shuffle
    csum += __shfl_xor(csum, 32, 64);
    gsum += __shfl_xor(gsum, 32, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) {
      if (af.db_out) af.db_out[n] = csum;
      if (af.dg_out) af.dg_out[n] = gsum;
      if (af.dbeta_out) af.dbeta_out[n] = bsum;
    }
    tile_rows_to_hbm(out, n0, g_out, ld_g, b0, batch, af.wt, c + 32 * h);
  }
}

// Epilogue of a TRANSPOSED tile (mma_unit SWAP; MODE 0 and 2): lane = (example c, half h), register 4 q + i = column
// n0 + 8 q + 4 h + i -- four consecutive columns of one example per q, i.e. ONE 16-byte store where the other layout needs four
// 4-byte ones (what an epilogue costs is store INSTRUCTIONS: ~24 cycles each at the CU whatever their width,
// profiles/r4_tower8_stage_stamps.txt).  bias / s_out / t_out: LDS tables (float4 reads, the same for every lane of a half).
template <int MODE, int ACT, bool FULLR>
__device__ __forceinline__ void epilogue8t(const floatx16 &acc, int n0, int act, const float *__restrict__ bias_t,
                                           float *__restrict__ out, float *__restrict__ g_out, int64_t ld_g, int n_store,
                                           int64_t b0, int64_t batch, const Aff8 &af, int c, int h) {
  const int a_id = ACT >= 0 ? ACT : act;
  const bool row_ok = FULLR || b0 + c < batch;
  float *grow = g_out + (b0 + c) * ld_g;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int nb = n0 + 8 * q + 4 * h;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(1.f, 1.f, 1.f, 1.f), t4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 0) {
      if (bias_t) b4 = *reinterpret_cast<const float4 *>(bias_t + nb);
      if (af.s_out) {
        s4 = *reinterpret_cast<const float4 *>(af.s_out + nb);
        t4 = *reinterpret_cast<const float4 *>(af.t_out + nb);
      }
    }
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w}, tt[4] = {t4.x, t4.y, t4.z, t4.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = acc[4 * q + i];
      if (MODE == 0) {
        v = act_fwd(v + bb[i], a_id);
        out[(nb + i) * P + c] = v;
        o[i] = __fadd_rn(__fmul_rn(v, ss[i]), tt[i]);
      } else {
        o[i] = (MODE == 2 && af.wl) ? v + af.dl[c] * af.wl[nb + i] : v;
      }
    }
    if ((WD_CHAIN8_EXP & 8) || !row_ok) continue;
    if (nb + 3 < n_store) {
      wd::store4(reinterpret_cast<float4 *>(grow + nb), make_float4(o[0], o[1], o[2], o[3]), af.wt);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (nb + i < n_store) wd::store1(grow + nb + i, o[i], af.wt);
    }
  }
}

// One product stage: out[32 x N] = epilogue(in[32 x K] . W[K x N]), phase by phase (Args8.ph).
template <int MODE>
__device__ __forceinline__ void stage8(const Args8 &A, int s, const float *__restrict__ in, float *__restrict__ out,
                                       const float *__restrict__ a_prev, float *__restrict__ g_out, int64_t ld_g, int n_store,
                                       const float *__restrict__ bias /* LDS table */, int act, const Aff8 &af, floatx4v (&fb)[DW],
                                       float *lds, int64_t b0, int64_t batch, bool last_stage) {
  // diagnostics (wd_chain_opts_t.stamps): workgroup 0's wavefronts 0 and 4 stamp stage s at [64 + 16 s + 8 (wave / 4) + i]:
  // i = 0 entry, 1 first unit multiplied, 2 its ring complete, 3 before the epilogue / reduction barrier, 4 epilogue done, 5 stage end
  unsigned long long *dbg = (A.g.stamps && blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256) && s < 8)
                                ? A.g.stamps + 64 + 16 * s + (threadIdx.x ? 8 : 0) : nullptr;
  auto dstamp = [&](int i) { if (dbg) dbg[i] = __builtin_readcyclecounter(); };
  dstamp(0);
  using std::integral_constant;
  const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
  const int c = lane & 31, h = lane >> 5;
  const float *inA = in + h * P + c;
  const float *sA = af.s_in ? af.s_in + h : nullptr, *tA = af.s_in ? af.t_in + h : nullptr;
  float *scrA = lds + uni(A.scr_off[s]);
  const int alt = uni(A.alt_off[s]);
  float *scrB = alt >= 0 ? lds + alt : nullptr;    // (a stage without scratch has no split phase: scrA is not used then)
  const bool full_rows = b0 + RT <= batch;
  const int ph0 = uni(A.ph_first[s]), ph1 = uni(A.ph_first[s + 1]), nph = uni(A.nph);
  int nsplit_seen = 0;
  constexpr bool SWAP = MODE != 1;   // forward and dx tiles come out transposed (16-byte stores), gradient tiles column-per-lane
  auto epilogue = [&](const floatx16 &acc, int n0, float bv, float sv, float tv) {
    if (MODE == 1) {
      if (act == WD_ACT_RELU) epilogue8<1, WD_ACT_RELU, false>(acc, n0, bv, sv, tv, act, a_prev, out, g_out, ld_g, n_store, b0, batch, af, c, h);
      else epilogue8<1, -1, false>(acc, n0, bv, sv, tv, act, a_prev, out, g_out, ld_g, n_store, b0, batch, af, c, h);
    } else if (act == WD_ACT_RELU || MODE == 2) {
      if (full_rows) epilogue8t<MODE, WD_ACT_RELU, true>(acc, n0, act, bias, out, g_out, ld_g, n_store, b0, batch, af, c, h);
      else epilogue8t<MODE, WD_ACT_RELU, false>(acc, n0, act, bias, out, g_out, ld_g, n_store, b0, batch, af, c, h);
    } else {
      epilogue8t<MODE, -1, false>(acc, n0, act, bias, out, g_out, ld_g, n_store, b0, batch, af, c, h);
    }
  };
  for (int ph = ph0; ph < ph1; ++ph) {
    const Unit u = unit_of(A.ph[ph], wave);
    Unit nx{nullptr, 0, 0, 0};
    if (ph + 1 < nph) nx = unit_of(A.ph[ph + 1], wave);
    const bool on = u.KG > 0;
    gw_t Wl = u.W;
    // the stream behind this unit: the next unit's first groups, or (nothing follows) this unit's last group again
    gw_t Wn = nx.KG > 0 ? nx.W : Wl + (int64_t)(on ? u.KG - 1 : 0) * 64;
    const int KGn = nx.KG > 0 ? nx.KG : 1;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < NACC; ++r) acc[r] = 0.f;
    const int n0 = u.tile * RT, n = n0 + c;
    float bv = 0.f, sv = 1.0f, tv = 0.f;
    if (MODE == 1 && on && u.slice == 0 && af.s_out) sv = af.s_out[n];     // requested before the reduction loop, used after it
    if (on) {
      const float *ia = inA + (int64_t)u.slice * u.KG * GK * P;
      const float *sa = sA ? sA + u.slice * u.KG * GK : nullptr, *ta = tA ? tA + u.slice * u.KG * GK : nullptr;
      if (sA) {
        if ((u.KG & (DW - 1)) == 0) mma_unit<true, 0, SWAP>(ia, Wl, u.KG, Wn, KGn, acc, fb, sa, ta, lane);
        else if (u.KG < DW) mma_unit<true, 1, SWAP>(ia, Wl, u.KG, Wn, KGn, acc, fb, sa, ta, lane);
        else mma_unit<true, 2, SWAP>(ia, Wl, u.KG, Wn, KGn, acc, fb, sa, ta, lane);
      } else {
        if ((u.KG & (DW - 1)) == 0) mma_unit<false, 0, SWAP>(ia, Wl, u.KG, Wn, KGn, acc, fb, nullptr, nullptr, lane);
        else if (u.KG < DW) mma_unit<false, 1, SWAP>(ia, Wl, u.KG, Wn, KGn, acc, fb, nullptr, nullptr, lane);
        else mma_unit<false, 2, SWAP>(ia, Wl, u.KG, Wn, KGn, acc, fb, nullptr, nullptr, lane);
      }
    } else if (nx.KG > 0) {
      prefill(fb, Wn, KGn, lane);
    }
    // on every path, before any store of the epilogue: the next unit's first groups have arrived (see mma_unit)
    if (ph == ph0) dstamp(1);
    if (nx.KG > 0 && !(WD_CHAIN8_EXP & 32)) ring_complete(fb);
    if (ph == ph0) dstamp(2);
    const int ns = uni(A.ph[ph].ns), l2 = uni(A.ph[ph].ntp_log2);
    if (ns > 1) {
      // two scratch areas alternate when the stage has a dead region to spare (then a phase never writes what the previous
      // phase's owners may still read); with one area, wait for those readers first
      float *scr = ((nsplit_seen & 1) && scrB) ? scrB : scrA;
      if (nsplit_seen > 0 && !scrB) lds_barrier();
      const int tl = u.tile - uni(A.ph[ph].tile0);
      if (on && u.slice > 0) {
        float *dst = scr + (((u.slice - 1) << l2) + tl) * SLOT + lane;
#pragma unroll
        for (int r = 0; r < NACC; ++r) dst[r * 64] = acc[r];
      }
      lds_barrier();
      if (ph == ph0) dstamp(3);
      if (on && u.slice == 0) {
        for (int sl = 1; sl < ns; ++sl) {
          const float *src = scr + (((sl - 1) << l2) + tl) * SLOT + lane;
#pragma unroll
          for (int r = 0; r < NACC; ++r) acc[r] += src[r * 64];
        }
        epilogue(acc, n0, bv, sv, tv);
      }
      ++nsplit_seen;
    } else if (on) {
      if (ph == ph0) dstamp(3);
      epilogue(acc, n0, bv, sv, tv);
    }
    if (ph == ph0) dstamp(4);
  }
  if (!last_stage) lds_barrier();
  dstamp(5);
}

static_assert(sizeof(Args8) <= 4096, "kernel arguments");

__global__ void __launch_bounds__(NTHR, 2) k_tower_chain8(Args8 A) {
  const ChainArgs &g = A.g;
  constexpr int PARTS = NTHR / RT;   // lane groups of the head's dot product
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float sdl[RT];
  __shared__ float red[NTHR], red2[NTHR], red3[NTHR];
  __shared__ float swl[1024];  // logits-layer kernel (concatenating towers: the whole row)
  __shared__ int64_t s_eoff[WD_CHAIN_MAX_SLOTS], s_rbase[WD_CHAIN_MAX_SLOTS];   // fused input layer: slot descriptors
  __shared__ int32_t s_ocol[WD_CHAIN_MAX_SLOTS];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = uni(t >> 6);
  const int64_t b0 = (int64_t)blockIdx.x * RT;
  const int L = uni(g.L);
  float *regx = lds + uni(g.x_off);  // x; the dz_l take this region once x is dead
  int nstamp = 0;
  auto stamp = [&]() {
    if (g.stamps && (blockIdx.x == 0 || blockIdx.x == 100) && t == 0)
      g.stamps[(blockIdx.x ? 32 : 0) + nstamp] = __builtin_readcyclecounter();
    ++nstamp;
  };
  stamp();
  // (the same two workgroups also stamp the 100 MHz realtime clock at their start and end, [28] / [29] and [60] / [61]: cycle
  // stamps over realtime = the shader clock this launch really ran at, scripts/sclk_probe.py)
  if (g.stamps && (blockIdx.x == 0 || blockIdx.x == 100) && t == 0) g.stamps[(blockIdx.x ? 32 : 0) + 28] = wall_clock64();
  if (g.tile_stamps && t == 0) g.tile_stamps[2 * blockIdx.x] = wall_clock64();
  if (g.flags_prio) __builtin_amdgcn_s_setprio(3);

  // ---- the weight stream starts here: the first groups of this wavefront's unit of stage F_0 ---------------------------
  floatx4v fb[DW];
  {
    const Unit u0 = unit_of(A.ph[0], wave);
    if (u0.KG > 0) prefill(fb, u0.W, u0.KG, lane);
  }

  // ---- head inputs: requested now, consumed after the last hidden layer (no exposed latency there) -------------
  const int KL = g.win ? uni(g.KL) : uni(g.layer[L - 1].N);
  for (int n = t; n < KL; n += NTHR) swl[n] = g.w_logits[n];
  float h_wide = 0.f, h_y = 0.f, h_w = 1.0f, h_bias = 0.f;
  if (t < RT && b0 + t < g.batch) {
    if (g.wide_logit) h_wide = g.wide_logit[b0 + t];
    if (g.labels) h_y = g.labels[b0 + t];
    if (g.weights) h_w = g.weights[b0 + t];
  }
  if (t < RT) h_bias = g.b_logits[0];
  // BN affine tables of every hidden layer -> LDS: s = gamma * inv, t = beta (no BN: the identity, and the stages skip it); bias
  for (int l = 0; l < L; ++l) {
    const wd_chain_layer_t &ly = g.layer[l];
    float *tab = lds + g.tab_off[l];
    for (int n = t; n < ly.N; n += NTHR) {
      tab[n] = ly.gamma ? ly.gamma[n] * g.inv : 1.0f;
      tab[ly.N + n] = ly.beta ? ly.beta[n] : 0.0f;
      tab[2 * ly.N + n] = ly.bias ? ly.bias[n] : 0.0f;
    }
  }

  if (g.win) {
    // concatenating towers: the BN affine of the WHOLE row, column by column (x: the identity) -- a layer's input window spans
    // several segments, its fragment reads index these tables by column
    float *sall = lds + g.sall_off, *tall = lds + g.tall_off;
    for (int c = t; c < g.cols; c += NTHR) { sall[c] = 1.0f; tall[c] = 0.0f; }
    lds_barrier();
    for (int l = 0; l < L; ++l) {
      const wd_chain_layer_t &ly = g.layer[l];
      if (!ly.gamma) continue;
      const int c0 = g.seg_col[l + 1];
      for (int n = t; n < ly.N; n += NTHR) {
        sall[c0 + n] = ly.gamma[n] * g.inv;
        tall[c0 + n] = ly.beta[n];
      }
    }
  }

  float *s_wide = red2;  // [RT] wide logit of the tile's examples; `red2` is free until the gradient of the last hidden layer
  float wv[2] = {0.f, 0.f};   // this lane's wide weights, in flight from here (or from the tile gather) to the head
  if (g.wv) {
    // prefetched input layer: the tile's RT x S wide weights are one contiguous run of the per-occurrence list -- requested here,
    // coalesced, consumed in the head (registers -> LDS -> per-example sum, slots in order)
    const int nbag = RT * g.wv_S;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = t + NTHR * u;
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
    for (int i = t; i < S; i += NTHR) {                              // slot descriptors -> LDS (48-byte structs in HBM)
      const wd_slot_t sl = I.slots[i];
      s_eoff[i] = sl.emb_off;
      s_ocol[i] = sl.out_col;
      s_rbase[i] = sl.wide ? (int64_t)sl.row_base : (int64_t)-1;
    }
    const int nbag = RT * S;
    for (int i = t; i < nbag; i += NTHR) {
      const int m = i / S;
      s_id[i] = b0 + m < g.batch ? I.ids[b0 * S + i] : -1;
    }
    for (int i = t; i < (int)g.K0 * P; i += NTHR) regx[i] = 0.f;     // pad columns and dropped ids read as zero
    __syncthreads();
    // Request order = the order things are needed in: the embedding rows and the numeric columns (the x tile: needed by the
    // first product), THEN the wide weights (one line per occurrence, needed by the head three products later).
    constexpr int WV = 2, DV = 1;
    float dv[DV];
    const int lg = t % LG, grp = t / LG, ngrp = NTHR / LG;
    const int nwork = RT * NG;
    constexpr int QB = 8;      // bags per lane group in flight: the whole tile in ONE round at the Criteo shape (32 x 26 / 128 = 7)
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
          const int i = t + NTHR * u;
          dv[u] = 0.f;
          if (i < RT * I.ncols && b0 + i % RT < g.batch) dv[u] = I.dense[(b0 + i % RT) * I.ld_dense + i / RT];
        }
        if (I.wide) {
#pragma unroll
          for (int u = 0; u < WV; ++u) {
            const int i = t + NTHR * u;
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
      const int i = t + NTHR * u;
      if (i < RT * I.ncols && b0 + i % RT < g.batch) put_dense(i, dv[u]);
    }
    for (int i = t + NTHR * DV; i < RT * I.ncols; i += NTHR)
      if (b0 + i % RT < g.batch) put_dense(i, I.dense[(b0 + i % RT) * I.ld_dense + i / RT]);
  } else {
    // ---- x tile -> LDS [k][P] (rows beyond the batch read as zero): one example row per 16 lanes (16 float4 = 64 columns,
    // 256 B), up to 8 column blocks in flight per lane, then the transposing LDS stores ------------------------------------
    constexpr int NLD = 8;
    const int kq = t & 15, m = t >> 4;
    const bool live = b0 + m < g.batch;
    const int64_t row = live ? b0 + m : g.batch - 1;   // clamped: the load is unconditional
    for (int kb = 0; kb < g.K0; kb += 64 * NLD) {
      float4 v[NLD];
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int k = kb + 64 * i + 4 * kq;
        v[i] = *reinterpret_cast<const float4 *>(g.x + row * g.ld_act + (k < g.K0 ? k : 0));
      }
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int k = kb + 64 * i + 4 * kq;
        if (k < g.K0) {
          regx[(k + 0) * P + m] = live ? v[i].x : 0.f;
          regx[(k + 1) * P + m] = live ? v[i].y : 0.f;
          regx[(k + 2) * P + m] = live ? v[i].z : 0.f;
          regx[(k + 3) * P + m] = live ? v[i].w : 0.f;
        }
      }
    }
  }
  lds_barrier();
  stamp();
  if (g.tile_stamps && t == 0) g.tile_stamps[2 * blockIdx.x + 1] = wall_clock64();

  // ---- forward --------------------------------------------------------------------------------------------
  const float *in = regx;
  int K = g.K0;
  for (int l = 0; l < L; ++l) {
    const wd_chain_layer_t &ly = g.layer[l];
    float *out = lds + g.a_off[l];
    Aff8 af{};
    af.wt = g.flags_wt;
    if (g.win) {
      in = lds + g.in_off[l];       // the layer's window of the row tile ([h_{l-1} | .. | x] or [x | .. | h_{l-1}])
      if (l > 0) { af.s_in = lds + g.sall_off + g.in_col[l]; af.t_in = lds + g.tall_off + g.in_col[l]; }
    } else if (l > 0 && g.layer[l - 1].gamma) {
      af.s_in = lds + g.tab_off[l - 1]; af.t_in = af.s_in + g.layer[l - 1].N;
    }
    if (ly.gamma) { af.s_out = lds + g.tab_off[l]; af.t_out = af.s_out + ly.N; }
    stage8<0>(A, l, in, out, nullptr, ly.a_out, g.ld_act, ly.N, ly.bias ? lds + g.tab_off[l] + 2 * ly.N : nullptr, g.act, af, fb, lds,
              b0, g.batch, false);
    stamp();
    in = out;
    K = ly.N;
  }

  // ---- logits layer + head (in = a_{L-1} [K][P]) ------------------------------------------------------------
  if (g.win) {     // the logits layer reads its window of the row tile
    in = lds + g.in_off[L];
    K = KL;
  }
  const float *sL_ = g.win ? lds + g.sall_off + g.in_col[L] : lds + g.tab_off[L - 1];
  const float *tL_ = g.win ? lds + g.tall_off + g.in_col[L] : sL_ + K;
  // the last hidden layer inside that window: its activations, logits weights and affine (simple: the window IS that layer)
  const int NL = uni(g.layer[L - 1].N), woffL = g.win ? uni(g.seg_col[L] - g.in_col[L]) : 0;
  const float *aL = g.win ? lds + g.a_off[L - 1] : in;
  {
    const int m = t % RT, part = t / RT;
    const float *sL = sL_, *tL = tL_;     // affine of the logits layer's inputs (identity without BN / for x)
    const bool wlist = (g.in.emb && g.in.wide) || g.wv;      // wide weights of the tile's occurrences in the wv registers
    if (wlist) {
      // the wide weights requested with the tile have long arrived: registers -> LDS (the x region is dead since the first
      // product), then the wide logit of each example, slots in order (fixed summation order)
      const int S = uni(g.wv ? g.wv_S : g.in.S), nbag = RT * S;
      const float *wbias = g.wv ? g.wv_bias : g.in.wide_bias;
      float *wout = g.wv ? g.wv_out : g.in.wide_out;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (t + NTHR * u < nbag) regx[t + NTHR * u] = wv[u];
      lds_barrier();
      // 8 lanes per example: lane j adds the slots j, j + 8, ... in ascending order, the 8 partial sums meet in a fixed tree
      // (one wavefront used to walk all S slots of an example serially while the other 480 lanes waited)
      if (t < 8 * RT) {
        const int e = t >> 3, j = t & 7;
        float acc = 0.f;
        for (int sidx = j; sidx < S; sidx += 8) acc += regx[e * S + sidx];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        acc += wbias[0];
        if (j == 0) {
          s_wide[e] = acc;
          if (wout && b0 + e < g.batch) wout[b0 + e] = acc;
        }
      }
    }
    float d = 0.f;
    for (int n = part; n < K; n += PARTS) d += __fadd_rn(__fmul_rn(in[n * P + m], sL[n]), tL[n]) * swl[n];
    red[part * RT + m] = d;
    lds_barrier();
    const float h_wide_lds = (wlist && t < RT) ? s_wide[t] : 0.f;
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
  lds_barrier();

  // d(bn_{L-1}) = dlogit w^T;  dz_{L-1} = d(bn) * s * act'(a_{L-1});  per-tile partials of d beta, d gamma, d bias of layer
  // L-1 and of the logits-layer kernel gradient (layout of wd_chain_tail: [tile][K + 1], the last entry = sum dlogit)
  {
    float *dz = lds + g.dz_off[L - 1];
    const wd_chain_layer_t &ll = g.layer[L - 1];
    float *gdz = ll.dz_out;
    const float *sL = sL_, *tL = tL_;
    float *Gp = g.Gpart_logits ? g.Gpart_logits + (int64_t)blockIdx.x * (K + 1) : nullptr;
    auto logits_kernel_gradient = [&]() {     // its input is the logits layer's window of bn values (simple: bn_{L-1})
      for (int n = t; n < K; n += NTHR) {
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
    };
    if (g.win && Gp) {     // the window holds x, and the dz tiles below take x's region: read it first
      logits_kernel_gradient();
      lds_barrier();
    }
    const float *wl = swl + woffL, *sLs = sL + woffL;      // the last hidden layer's share of the logits layer
    const int KT = NL < NTHR ? NL : NTHR;   // lanes along n: coalesced HBM rows, conflict-free LDS
    const int MQ = NTHR / KT;               // N_{L-1} < 512: several example groups in parallel
    const int nq = t % KT, mq = t / KT;
    float *dbp = ll.db_part ? ll.db_part + (int64_t)blockIdx.x * NL : nullptr;
    float *dgp = ll.dgamma_part ? ll.dgamma_part + (int64_t)blockIdx.x * NL : nullptr;
    float *dtp = ll.dbeta_part ? ll.dbeta_part + (int64_t)blockIdx.x * NL : nullptr;
    if (mq < MQ) {
      for (int n = nq; n < NL; n += KT) {
        const float w = wl[n], sv = sLs[n];
        float csum = 0.f, gsum = 0.f, bsum = 0.f;
        for (int m = mq; m < RT; m += MQ) {
          const float a = aL[n * P + m];
          const float dbn = sdl[m] * w;
          bsum += dbn;
          gsum += dbn * a;
          const float v = dbn * sv * act_bwd(a, g.act);
          dz[n * P + m] = v;
          if (b0 + m < g.batch) wd::store1(&gdz[(b0 + m) * NL + n], v, g.flags_wt);
          csum += v;
        }
        if (MQ == 1) {
          if (dbp) dbp[n] = csum;
          if (dgp) dgp[n] = gsum;
          if (dtp) dtp[n] = bsum;
        } else {     // N_{L-1} < 512: MQ * KT <= 512 partial sums each, combined below in group order
          red[mq * KT + n] = csum;
          red2[mq * KT + n] = gsum;
          red3[mq * KT + n] = bsum;
        }
      }
    }
    if (MQ > 1) {
      lds_barrier();
      if (t < NL) {
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
    if (Gp && !g.win) logits_kernel_gradient();
  }
  lds_barrier();
  stamp();

  // ---- gradient chain: d(bn_{l-1}) = dz_l W_l^T, dz_{l-1} = d(bn_{l-1}) * s_{l-1} * act'(a_{l-1});  dx = dz_0 W_0^T -----
  int s = L;
  const bool want_dx = g.dx && g.dx_cols > 0;
  for (int l = L - 1; l >= 1; --l, ++s) {
    const wd_chain_layer_t &ly = g.layer[l];
    const wd_chain_layer_t &lp = g.layer[l - 1];
    Aff8 af{};
    af.wt = g.flags_wt;
    if (lp.gamma) { af.s_out = lds + g.tab_off[l - 1]; af.t_out = af.s_out + lp.N; }
    af.db_out = lp.db_part ? lp.db_part + (int64_t)blockIdx.x * lp.N : nullptr;
    af.dg_out = lp.dgamma_part ? lp.dgamma_part + (int64_t)blockIdx.x * lp.N : nullptr;
    af.dbeta_out = lp.dbeta_part ? lp.dbeta_part + (int64_t)blockIdx.x * lp.N : nullptr;
    if (g.win) { af.dl = sdl; af.wl = swl + (g.seg_col[l] - g.in_col[L]); }      // + dlogit x the logits weights of this segment
    stage8<1>(A, s, lds + g.dz_off[l], lds + g.dz_off[l - 1], lds + g.a_off[l - 1], lp.dz_out, lp.N, lp.N, nullptr, g.act, af, fb,
              lds, b0, g.batch, l == 1 && !want_dx);
    stamp();
  }
  if (want_dx) {
    Aff8 af{};
    af.wt = g.flags_wt;
    if (g.win) { af.dl = sdl; af.wl = swl + (g.seg_col[0] - g.in_col[L]); }
    stage8<2>(A, s, lds + g.dz_off[0], nullptr, nullptr, g.dx, g.ld_dx, g.K0, nullptr, 0, af, fb, lds, b0, g.batch, true);
  }
  stamp();
  if (g.stamps && (blockIdx.x == 0 || blockIdx.x == 100) && t == 0) {
    g.stamps[(blockIdx.x ? 32 : 0) + 29] = wall_clock64();
    g.stamps[(blockIdx.x ? 32 : 0) + 27] = (unsigned long long)(nstamp - 1);     // index of the last cycle stamp
  }
}

// LDS layout: [x | dz_{L-1} .. dz_0 aliasing x] [a_0] [a_1] ... [BN + bias tables]   (rows of 33 floats)
int64_t layout8(int32_t K0, const int32_t *N, int32_t L, int32_t *a_off, int32_t *dz_off, int32_t *tab_off,
                int64_t *regx_floats, int64_t *sum_n_out) {
  if (L < 1 || L > MAXL || K0 <= 0 || K0 % RT) return -1;
  int64_t sum_n = 0;
  for (int l = 0; l < L; ++l) {
    if (N[l] <= 0 || N[l] % RT || N[l] > 512) return -1;
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
  for (int l = 0; l < L; ++l) {
    if (tab_off) tab_off[l] = (int32_t)off;
    off += 3 * (int64_t)N[l];       // s_l, t_l, bias_l
  }
  if (regx_floats) *regx_floats = regx;
  if (sum_n_out) *sum_n_out = sum_n;
  const int64_t bytes = off * 4;
  return bytes <= DYN_LDS_MAX ? bytes : -1;
}

// Concatenating towers: the row tile mirrors the activation row (cols columns of P floats; segment s at seg_col[s]); the dz
// tiles take x's region once x is dead, in ASCENDING layer order -- the gradient of segment l is one product over the suffix
// [dz_l | .. | dz_{L-1}] --; behind the tile the per-layer tables (s, t, bias) and the column-wise affine of the whole row.
int64_t layout8w(const wd_chain_windows_t &w, int32_t K0, const int32_t *N, int32_t L, int32_t *a_off, int32_t *dz_off,
                 int32_t *tab_off, int32_t *x_off, int32_t *sall_off, int32_t *tall_off) {
  if (L < 1 || L > MAXL || K0 <= 0 || K0 % RT || w.cols <= 0 || w.cols % RT || w.k_logits <= 0 || w.k_logits > 1024) return -1;
  int64_t sum_n = 0;
  for (int l = 0; l < L; ++l) {
    if (N[l] <= 0 || N[l] % RT || N[l] > 512) return -1;
    if (w.seg_col[l + 1] < 0 || w.seg_col[l + 1] % RT || w.seg_col[l + 1] + N[l] > w.cols) return -1;
    sum_n += N[l];
  }
  if (w.seg_col[0] < 0 || w.seg_col[0] % RT || w.seg_col[0] + K0 > w.cols || sum_n > K0) return -1;
  for (int l = 0; l <= L; ++l)
    if (w.in_col[l] < 0 || w.in_col[l] % 8) return -1;
  if (w.in_col[L] + w.k_logits > w.cols) return -1;
  for (int l = 0; l < L; ++l)
    if (a_off) a_off[l] = (int32_t)((int64_t)w.seg_col[l + 1] * P);
  if (x_off) *x_off = (int32_t)((int64_t)w.seg_col[0] * P);
  int64_t off = (int64_t)w.seg_col[0] * P;
  for (int l = 0; l < L; ++l) {
    if (dz_off) dz_off[l] = (int32_t)off;
    off += (int64_t)N[l] * P;
  }
  off = (int64_t)w.cols * P;
  for (int l = 0; l < L; ++l) {
    if (tab_off) tab_off[l] = (int32_t)off;
    off += 3 * (int64_t)N[l];
  }
  if (sall_off) *sall_off = (int32_t)off;
  off += w.cols;
  if (tall_off) *tall_off = (int32_t)off;
  off += w.cols;
  const int64_t bytes = off * 4;
  return bytes <= DYN_LDS_MAX ? bytes : -1;
}

// the phases of one stage: 8 column tiles x 1 slice while 8 are left, then 4 x 2, 2 x 4, 1 x 8 for what remains.  The partial
// tiles of the split phases go to a scratch in `free` -- regions of the LDS tile that are dead while the stage runs ({float
// offset, floats} pairs) --, two of them to alternate between if there is room; no room at all: no split (some wavefronts idle).
struct Region { int64_t off, size; };
bool add_stage(Args8 &A, int &nph, int &s, const float *Wpk, int K, int ncols, const Region *free, int nfree) {
  const int KG = K / GK, ntiles = ncols / RT;
  const int64_t want = (int64_t)SCR_SLOTS * SLOT;
  int ia = -1, ib = -1;     // largest region for the scratch; a second one in what is left of it or in another region
  for (int i = 0; i < nfree; ++i)
    if (free[i].size >= SLOT && (ia < 0 || free[i].size > free[ia].size)) ia = i;
  // partial tiles the scratch can hold (a phase of 2^l2 tiles in ns slices needs (ns - 1) 2^l2): a stage with less dead LDS than
  // the seven a full split wants -- the forward of a concatenating tower, whose x stays alive -- splits less
  const int slots = ia >= 0 ? (int)(free[ia].size / SLOT < SCR_SLOTS ? free[ia].size / SLOT : SCR_SLOTS) : 0;
  A.scr_off[s] = ia >= 0 ? (int32_t)free[ia].off : -1;
  A.alt_off[s] = -1;
  if (ia >= 0 && slots == SCR_SLOTS) {
    if (free[ia].size >= 2 * want) A.alt_off[s] = (int32_t)(free[ia].off + want);
    else
      for (int i = 0; i < nfree; ++i)
        if (i != ia && free[i].size >= want && (ib < 0 || free[i].size > free[ib].size)) ib = i;
    if (ib >= 0) A.alt_off[s] = (int32_t)free[ib].off;
  }
  A.ph_first[s] = nph;
  int t0 = 0;
  while (t0 < ntiles) {
    const int rem = ntiles - t0;
    int l2, ns;
    if (rem >= 8) { l2 = 3; ns = 1; }
    else if (rem >= 4) { l2 = 2; ns = 2; }
    else if (rem >= 2) { l2 = 1; ns = 4; }
    else { l2 = 0; ns = 8; }
    while (ns > 1 && KG % ns) ns >>= 1;
    while (ns > 1 && ((ns - 1) << l2) > slots) ns >>= 1;
    if (nph >= MAXPH) return false;
    A.ph[nph++] = Phase8{Wpk, KG, t0, l2, ns, KG / ns};
    t0 += 1 << l2;
  }
  ++s;
  return true;
}

}  // namespace

namespace wd {

int64_t chain8_lds_bytes(int32_t K0, const int32_t *N, int32_t L, int32_t dx_cols) {
  (void)dx_cols;
  return layout8(K0, N, L, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int64_t chain8_windows_lds_bytes(const wd_chain_windows_t *w, int32_t K0, const int32_t *N, int32_t L) {
  return layout8w(*w, K0, N, L, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// adjacent dead regions as one (the scratch of a split phase wants SCR_SLOTS tiles in a row)
static int merge_regions(Region *r, int n) {
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && r[j].off < r[j - 1].off; --j) { const Region q = r[j]; r[j] = r[j - 1]; r[j - 1] = q; }
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (r[i].size <= 0) continue;
    if (m > 0 && r[m - 1].off + r[m - 1].size == r[i].off) r[m - 1].size += r[i].size;
    else r[m++] = r[i];
  }
  return m;
}

int chain8_launch(const ChainArgs &g0, wd_stream_t stream) {
  Args8 A{};
  A.g = g0;
  ChainArgs &g = A.g;
  const int L = g.L;
  int32_t N[MAXL];
  for (int l = 0; l < L; ++l) N[l] = g.layer[l].N;
  int64_t regx = 0, sum_n = 0;
  int64_t bytes;
  Region win_scr{0, 0};
  if (g.win) {
    wd_chain_windows_t w{};
    for (int l = 0; l <= L; ++l) { w.seg_col[l] = g.seg_col[l]; w.in_col[l] = g.in_col[l]; }
    w.k_logits = g.KL; w.cols = g.cols;
    bytes = layout8w(w, g.K0, N, L, g.a_off, g.dz_off, g.tab_off, &g.x_off, &g.sall_off, &g.tall_off);
    for (int l = 0; l <= L; ++l) g.in_off[l] = (int32_t)((int64_t)g.in_col[l] * P);
    if (bytes > 0) {
      // x stays alive through the forward, so its stages have (almost) no dead region for the partial tiles of a split phase:
      // what the LDS has left behind the tables is theirs (configs[3]: 4 slots -- F1 runs 4 tiles x 2 slices instead of 4 x 1 with four
      // wavefronts idle, F2 2 x 2 instead of 2 x 1)
      const int64_t spare = (DYN_LDS_MAX - bytes) / ((int64_t)SLOT * 4);
      win_scr = Region{bytes / 4, (spare < SCR_SLOTS ? spare : SCR_SLOTS) * SLOT};
      bytes += win_scr.size * 4;
    }
  } else {
    bytes = layout8(g.K0, N, L, g.a_off, g.dz_off, g.tab_off, &regx, &sum_n);
  }
  if (bytes <= 0) return 1;    // > 0: this kernel does not take the shape -- the caller falls back
  // 16-byte stores of the activations, dz and dx
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (g.ld_act % 4 || (g.dx && (g.ld_dx % 4 || !al16(g.dx)))) return 1;
  for (int l = 0; l < L; ++l)
    if (!al16(g.layer[l].a_out) || (g.train && !al16(g.layer[l].dz_out))) return 1;
  int nph = 0, s = 0;
  int nph_fwd = 0;
  if (g.win) {
    // F_l reads its window (x stays alive through the whole forward): only the outputs of LATER layers are dead
    for (int l = 0; l < L; ++l) {
      Region fr[MAXL + 1];
      int nf = 0;
      for (int j = l + 1; j < L; ++j) fr[nf++] = Region{g.a_off[j], (int64_t)N[j] * P};
      nf = merge_regions(fr, nf);
      if (win_scr.size > 0) {      // the dedicated scratch, unless a dead region is larger
        bool larger = false;
        for (int i = 0; i < nf; ++i) larger = larger || fr[i].size >= win_scr.size;
        if (!larger) { fr[0] = win_scr; nf = 1; }
      }
      if (g.layer[l].K <= 0 || g.layer[l].K % GK) return 1;
      if (!add_stage(A, nph, s, g.layer[l].Wpk, g.layer[l].K, N[l], fr, nf)) return 1;
    }
    nph_fwd = nph;
    if (g.train) {
      // the stage that produces dz_{l-1} reads [dz_l | .. | dz_{L-1}] and a_{l-1}: a_l .. a_{L-1} are dead, and so is what the x
      // region holds below dz_{l-1}
      int64_t kred = 0;
      for (int l = L - 1; l >= 1; --l) {
        kred += N[l];
        Region fr[MAXL + 1];
        int nf = 0;
        for (int j = l; j < L; ++j) fr[nf++] = Region{g.a_off[j], (int64_t)N[j] * P};
        if (l >= 2) fr[nf++] = Region{g.x_off, (int64_t)g.dz_off[l - 1] - g.x_off};
        nf = merge_regions(fr, nf);
        if (!add_stage(A, nph, s, g.layer[l].WTpk, (int)kred, N[l - 1], fr, nf)) return 1;
      }
      if (g.dx && g.dx_cols > 0) {
        kred += N[0];
        Region fr[MAXL];
        int nf = 0;
        for (int j = 0; j < L; ++j) fr[nf++] = Region{g.a_off[j], (int64_t)N[j] * P};
        nf = merge_regions(fr, nf);
        if (!add_stage(A, nph, s, g.layer[0].WTpk, (int)kred, g.dx_cols, fr, nf)) return 1;
      }
    }
  } else {
  const int64_t a_end = g.a_off[L - 1] + (int64_t)N[L - 1] * P, dz_end = g.dz_off[0] + (int64_t)N[0] * P;
  for (int l = 0; l < L; ++l) {
    // F_l reads x (l = 0) or a_{l-1} and writes a_l: the x region is dead from F_1 on, a_{l+1} .. are not written yet
    Region fr[2];
    int nf = 0;
    if (l >= 1) fr[nf++] = Region{0, regx};
    if (l + 1 < L) fr[nf++] = Region{g.a_off[l + 1], a_end - g.a_off[l + 1]};
    if (!add_stage(A, nph, s, g.layer[l].Wpk, l == 0 ? g.K0 : N[l - 1], N[l], fr, nf)) return 1;
  }
  nph_fwd = nph;
  if (g.train) {
    for (int l = L - 1; l >= 1; --l) {
      // B_l reads dz_l and a_{l-1}, writes dz_{l-1}: dz_{L-1} .. dz_{l+1} and a_l .. a_{L-1} are dead, dz_{l-2} .. dz_0 (and
      // what the x region has behind dz_0) not written yet
      Region fr[3];
      int nf = 0;
      if (g.dz_off[l] > 0) fr[nf++] = Region{0, g.dz_off[l]};
      if (l >= 2) fr[nf++] = Region{g.dz_off[l - 2], regx - g.dz_off[l - 2]};
      else if (regx > dz_end) fr[nf++] = Region{dz_end, regx - dz_end};
      fr[nf++] = Region{g.a_off[l], a_end - g.a_off[l]};
      if (!add_stage(A, nph, s, g.layer[l].WTpk, N[l], N[l - 1], fr, nf)) return 1;
    }
    if (g.dx && g.dx_cols > 0) {
      // dx reads dz_0: every a_l and dz_{L-1} .. dz_1 are dead
      Region fr[3];
      int nf = 0;
      if (g.dz_off[0] > 0) fr[nf++] = Region{0, g.dz_off[0]};
      if (regx > dz_end) fr[nf++] = Region{dz_end, regx - dz_end};
      fr[nf++] = Region{g.a_off[0], a_end - g.a_off[0]};
      if (!add_stage(A, nph, s, g.layer[0].WTpk, N[0], g.dx_cols, fr, nf)) return 1;
    }
  }
  }
  A.ph_first[s] = nph;
  A.nph = g.train ? nph : nph_fwd;
  static int64_t attr_bytes = 0;
  if (bytes > attr_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_tower_chain8), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)bytes);
    if (e != hipSuccess) {
      set_error("wd_tower_chain: hipFuncSetAttribute(%lld): %s", (long long)bytes, hipGetErrorString(e));
      return WD_ERR_LAUNCH;
    }
    attr_bytes = bytes;
  }
  hipLaunchKernelGGL(k_tower_chain8, dim3((unsigned)ceil_div(g.batch, (int64_t)RT)), dim3(NTHR), (size_t)bytes, as_stream(stream), A);
  return check_launch("wd_tower_chain");
}

}  // namespace wd
