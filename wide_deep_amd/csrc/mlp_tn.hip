// wide_deep_amd/csrc/mlp_tn.hip -- the weight-gradient products of a tower, G_l = in_l^T dz_l, as register-streamed MFMA products
// (round 5).  Replaces what tf.gradients emits for the kernels of python/lib/dnn.py:100-104 (tf.layers.dense) under the dnn-scope
// minimize of python/lib/joint.py:233-241: a reduction over the whole batch, `simple` and per-layer towers alike.
//
// Why not the LDS-tiled GEMM of mlp.hip (k_gemm_tn_group: 64 x 64 workgroup tile, one 32 x 32 accumulator per wavefront, operands
// staged through a double-buffered LDS slab; 38 us alone at the C2 shape = 0.43 of the fp32 MFMA peak): BOTH operands of a TN
// product are "output-contiguous" -- row b of the activations holds 64 consecutive output rows of G, row b of dz 64 consecutive
// output columns -- which is exactly the layout of an MFMA fragment.  v_mfma_f32_32x32x2_f32 takes A[i][k] from lane i + 32 k:
// with k = the example, lanes 0..31 read one 256-byte run of row b and lanes 32..63 of row b + 1, as ONE global_load_dwordx2 per
// operand whose two registers are the fragments of two 32-wide tiles (register r of lane i is output row 2 i + r: a permutation of
// the tile's rows that only the final store has to know).  So nothing is staged: a wavefront owns a 64 x 64 output tile (four
// accumulators), streams its two operands through a double-buffered register ring of 8 reduction steps per set, and issues
// 4 MFMAs per 2 loads; no LDS traffic, no barrier in the loop, one wait per set.
//
// A workgroup = one 64 x 64 tile over one slice of the batch; its four wavefronts take a quarter of the slice each and meet in LDS
// (register-major, no transposition) in a fixed order ((w0 + w2) + (w1 + w3)), so a slice costs ONE partial in HBM where four
// independent wavefronts would cost four.  The splits are laid over the XCDs in split-major order (workgroup i runs on XCD i % 8):
// every XCD gets a contiguous run of (split, tile) pairs, i.e. tiles that share operand panels over the same examples.
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2v __attribute__((ext_vector_type(2)));

constexpr int TS_MAXJ = WD_TN_GROUP_MAX;
constexpr int TS_MAXS = 32;      // slices of the batch
constexpr int U = 8;             // reduction steps (of two examples) per register set
constexpr int NSET = 3;          // register sets: two in flight while one is multiplied

struct TsJob {
  const float *A, *B;    // colsum: A only
  float *C;
  int32_t lda, ldb;      // floats
  int32_t M, N, K;       // output M x N, reduction K (examples); colsum: N columns, K rows
  int32_t tiles_n;       // 64 x 64 tiles per row of tiles
  int32_t pad_[3];
};

// Everything a workgroup needs to find its work sits in the first two cache lines of the kernel arguments (the prefix arrays) and
// in ONE 64-byte job record: a scan over the records themselves is a chain of dependent scalar-cache misses (12 us of a 43 us
// launch in the first version of this kernel, profiles/r5_products_stream.md).
struct TsArgs {
  int32_t tile_first[TS_MAXJ + 1];   // product jobs: prefix of their 64 x 64 tiles (entries past the last job: INT_MAX)
  int32_t cs_first[TS_MAXJ + 1];     // column-sum jobs: prefix of their workgroups (each padded to a multiple of 8)
  int32_t nprod, ncs;
  int32_t n_colsum_wg;          // workgroups [0, n_colsum_wg): column sums
  int32_t tiles;                // tiles of all product jobs = pairs per slice
  int32_t nsplit;               // slices of the batch (the same for every product job)
  int32_t pad_;
  TsJob prod[TS_MAXJ];
  TsJob cs[TS_MAXJ];
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// number of prefix entries 1 .. TS_MAXJ that are <= f: the job that owns flat index f (all entries loaded at once, no chain)
__device__ __forceinline__ int owner_of(const int32_t (&first)[TS_MAXJ + 1], int f) {
  int j = 0;
#pragma unroll
  for (int k = 1; k <= TS_MAXJ; ++k) j += (f >= first[k]) ? 1 : 0;
  return j;
}

// EXP: diagnostics (WD_TNS_EXP), bits: 1 no MFMAs, 2 no loads in the loop -- separate instantiations, the product kernel has none of it
template <int EXP, int PRIO>
__global__ void __launch_bounds__(256) k_tn_stream(TsArgs G) {
  __shared__ __attribute__((aligned(16))) float red[2][64][64];   // two partial tiles, register-major: [slot][register][lane]
  const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
  if ((int)blockIdx.x < G.n_colsum_wg) {
    // column sums (bias / BN gradients, the logits layer's per-tile partials, the loss): 64 columns per workgroup, 4 row groups
    // with 16 independent loads in flight each, combined in group order (fixed summation order)
    const int j = uni(owner_of(G.cs_first, (int)blockIdx.x));
    const TsJob &g = G.cs[j];
    const int rem = blockIdx.x - G.cs_first[j];
    float (*part)[64] = reinterpret_cast<float (*)[64]>(&red[0][0][0]);
    const int c = lane, q = wave;
    const int64_t n = (int64_t)rem * 64 + c;
    if ((int64_t)rem * 64 >= g.N) return;    // padding of the job's range
    float acc = 0.f;
    if (n < g.N) {
      for (int64_t k0 = q; k0 < g.K; k0 += 64) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = k0 + 4 * u < g.K ? g.A[(k0 + 4 * u) * g.lda + n] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
      }
    }
    part[q][c] = acc;
    __syncthreads();
    if (q == 0 && n < g.N) g.C[n] = part[0][c] + part[1][c] + part[2][c] + part[3][c];
    return;
  }
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);

  // ---- which (slice, job, tile): pairs are slice-major, every XCD (workgroup i runs on XCD i % 8) a contiguous run of them --------
  const int id = blockIdx.x - G.n_colsum_wg;
  const int xcd = id % wd::kXCDs, loc = id / wd::kXCDs;
  const int total = G.tiles * G.nsplit;
  const int f0 = (int)(((int64_t)xcd * total) / wd::kXCDs), f1 = (int)(((int64_t)(xcd + 1) * total) / wd::kXCDs);
  int f = f0 + loc;
  if (f >= f1) return;
  const int s = uni(f / G.tiles);
  f -= s * G.tiles;
  const int j = uni(owner_of(G.tile_first, f));
  f = uni(f - G.tile_first[j]);
  const TsJob &g = G.prod[j];
  const int M = uni(g.M), N = uni(g.N), K = uni(g.K), lda = uni(g.lda), ldb = uni(g.ldb), ns = uni(G.nsplit), tn = uni(g.tiles_n);
  const int m0 = (f / tn) * 64, n0 = (f % tn) * 64;

  // The reduction in SETS of U steps of two examples (2 U examples): the slice's range of sets, then this wavefront's quarter of
  // it.  The last set of the batch may be partial: rows past the batch lie past the buffer descriptors' ends and read as zeros.
  const int NT = (K + 2 * U - 1) / (2 * U);
  const int s0 = (int)(((int64_t)s * NT) / ns), s1 = (int)(((int64_t)(s + 1) * NT) / ns);
  const int t0 = s0 + (int)(((int64_t)wave * (s1 - s0)) / 4), t1 = s0 + (int)(((int64_t)(wave + 1) * (s1 - s0)) / 4);

  const int i = lane & 31, h = lane >> 5;
  // columns past the matrix: the address is clamped to the last even pair (legal, finite or not: a garbage operand ROW only
  // reaches its own output row / column, which is not stored)
  const int ca = min(m0 + 2 * i, (M - 1) & ~1), cb = min(n0 + 2 * i, (N - 1) & ~1);
  const int offA = (h * lda + ca) * 4, offB = (h * ldb + cb) * 4;
  // operands through buffer descriptors: address = base + per-lane offset (a VGPR that never changes) + scalar offset of the step
  // -- no 64-bit address arithmetic in vector registers, and anything at or past row K reads as 0
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.A), 0, (int)(((int64_t)K * lda) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.B), 0, (int)(((int64_t)K * ldb) * 4), 0x00020000);
  const int stepA = lda * 8, stepB = ldb * 8;   // bytes per reduction step (two rows)

  floatx16 acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.f;

  floatx2v ra[NSET][U], rb[NSET][U];
  // every load is unconditional (a load inside a branch makes the wait at the join vmcnt(0)): sets behind the wavefront's last one
  // are the next wavefront's (or zeros past the batch) and are not multiplied
  auto load_set = [&](int buf, int t) {
    const int sa = t * (U * stepA), sb = t * (U * stepB);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ra[buf][u] = __builtin_bit_cast(floatx2v, __builtin_amdgcn_raw_buffer_load_b64(rA, offA, sa + u * stepA, 0));
      rb[buf][u] = __builtin_bit_cast(floatx2v, __builtin_amdgcn_raw_buffer_load_b64(rB, offB, sb + u * stepB, 0));
    }
  };
  auto mm = [&](const floatx2v &a, const floatx2v &b) {
    const float ax = a.x, ay = a.y, bx = b.x, by = b.y;
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bx, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, by, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, bx, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, by, acc[1][1], 0, 0, 0);
  };
  auto compute_set = [&](int buf) {
    if (EXP & 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) acc[0][0][u] += fminf(ra[buf][u].x + rb[buf][u].y + ra[buf][u].y + rb[buf][u].x, 1.f);
      return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) mm(ra[buf][u], rb[buf][u]);
  };
#define WD_TNS_STAGE(ld, cp)                         \
  if (!(EXP & 2)) load_set(ld, t + 2);               \
  __builtin_amdgcn_sched_barrier(0);                 \
  compute_set(cp);                                   \
  __builtin_amdgcn_sched_barrier(0);                 \
  if (++t >= t1) break;
  if (t0 < t1) {
    load_set(0, t0);
    load_set(1, t0 + 1);
    for (int t = t0;;) {
      WD_TNS_STAGE(2, 0)
      WD_TNS_STAGE(0, 1)
      WD_TNS_STAGE(1, 2)
    }
  }
#undef WD_TNS_STAGE

  // ---- the four wavefronts' partial tiles meet: (w0 + w2) + (w1 + w3) ------------------------------------------------------------
  if (wave >= 2) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[wave - 2][(r * 2 + c) * 16 + q][lane] = acc[r][c][q];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[r][c][q] += red[wave][(r * 2 + c) * 16 + q][lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[0][(r * 2 + c) * 16 + q][lane] = acc[r][c][q];
  }
  __syncthreads();
  if (wave != 0) return;
  // output row of (register q, tile r) = m0 + 2 row_of(q, h) + r, columns n0 + 2 i + {0, 1}: one 8-byte store, 256-byte runs
  float *Cz = g.C + (int64_t)s * M * N;
  const int n = n0 + 2 * i;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + 2 * row_of(q, h) + r;
      const float v0 = acc[r][0][q] + red[0][(r * 2 + 0) * 16 + q][lane];
      const float v1 = acc[r][1][q] + red[0][(r * 2 + 1) * 16 + q][lane];
      if (m < M && n < N) {
        if (n + 1 < N) *reinterpret_cast<float2 *>(Cz + (int64_t)m * N + n) = make_float2(v0, v1);
        else Cz[(int64_t)m * N + n] = v0;
      }
    }
}

inline bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

}  // namespace

namespace wd {

// > 0: this group is not for the streamed kernel (the caller runs the LDS-tiled one); 0 / < 0: launched / error
int tn_stream_launch(const wd_tn_job_t *jobs, int32_t njobs, wd_stream_t stream) {
  // Off by default: alone the streamed kernel is the faster one (C2 shape: 35-38 us against 45 for the LDS-tiled kernel on the
  // same box), but in the step the products run beside the row update, and there it loses (0.165-0.168 against 0.161-0.164
  // ms/step: the update stretches from 50 to 60 us and the dense tail from 6 to 15) -- profiles/r5_products_stream.md.
  static const bool on = getenv("WD_TN_STREAM") && atoi(getenv("WD_TN_STREAM")) != 0;
  if (!on || njobs > TS_MAXJ) return 1;
  TsArgs G{};
  int ncs = 0, nsplit = 0, tiles = 0;
  for (int j = 0; j < njobs; ++j) {
    const wd_tn_job_t &q = jobs[j];
    if (!q.B) {    // column sums go first in the grid (a chain of dependent round trips: at its end they would finish last)
      if (!(q.A && q.Cpart && q.N > 0 && q.K > 0) || q.lda > (1 << 24) || q.N > (1 << 24) || q.K > (1 << 24)) return 1;
      TsJob &g = G.cs[G.ncs];
      g.A = q.A; g.C = q.Cpart; g.lda = (int32_t)q.lda; g.N = (int32_t)q.N; g.K = (int32_t)q.K;
      G.cs_first[G.ncs++] = ncs;
      ncs += (int)ceil_div(ceil_div(q.N, 64), 8) * 8;
      continue;
    }
    if (!(q.A && q.Cpart) || q.M <= 0 || q.N <= 0 || q.K < 2 || q.nsplit <= 0 || q.nsplit > TS_MAXS || q.append_ones) return 1;
    if (q.lda % 2 || q.ldb % 2 || !aligned8(q.A) || !aligned8(q.B) || !aligned8(q.Cpart) || q.N % 2) return 1;
    if ((q.M & 1) && q.M >= q.lda) return 1;         // an odd width reads (and ignores) column M of its last pair
    if (q.M > (1 << 24) || q.N > (1 << 24) || q.K * q.lda * 4 >= (int64_t(1) << 31) || q.K * q.ldb * 4 >= (int64_t(1) << 31)) return 1;
    if (nsplit && q.nsplit != nsplit) return 1;      // one slicing of the batch for the whole group
    nsplit = q.nsplit;
    TsJob &g = G.prod[G.nprod];
    g.A = q.A; g.B = q.B; g.C = q.Cpart; g.lda = (int32_t)q.lda; g.ldb = (int32_t)q.ldb;
    g.M = (int32_t)q.M; g.N = (int32_t)q.N; g.K = (int32_t)q.K;
    g.tiles_n = (int)ceil_div(q.N, 64);
    G.tile_first[G.nprod++] = tiles;
    tiles += (int)(ceil_div(q.M, 64) * ceil_div(q.N, 64));
  }
  if (G.nprod == 0) return 1;     // no product in the group
  for (int j = G.nprod; j <= TS_MAXJ; ++j) G.tile_first[j] = j == G.nprod ? tiles : INT32_MAX;
  for (int j = G.ncs; j <= TS_MAXJ; ++j) G.cs_first[j] = j == G.ncs ? ncs : INT32_MAX;
  G.n_colsum_wg = ncs; G.tiles = tiles; G.nsplit = nsplit;
  static const int pad_lds = getenv("WD_TNS_LDS") ? atoi(getenv("WD_TNS_LDS")) : 0;   // extra LDS bytes: workgroups per CU
  const int grid = ncs + kXCDs * (int)ceil_div((int64_t)tiles * nsplit, kXCDs);
  static const int exp = getenv("WD_TNS_EXP") ? atoi(getenv("WD_TNS_EXP")) : 0;
  static const int prio = getenv("WD_TNS_PRIO") ? atoi(getenv("WD_TNS_PRIO")) : 0;   // issue priority of the product wavefronts
  const dim3 gr((unsigned)grid), bl(256);
  const hipStream_t st = as_stream(stream);
  if (exp == 1) hipLaunchKernelGGL((k_tn_stream<1, 0>), gr, bl, (size_t)pad_lds, st, G);
  else if (exp == 2) hipLaunchKernelGGL((k_tn_stream<2, 0>), gr, bl, (size_t)pad_lds, st, G);
  else if (exp == 3) hipLaunchKernelGGL((k_tn_stream<3, 0>), gr, bl, (size_t)pad_lds, st, G);
  else if (prio == 3) hipLaunchKernelGGL((k_tn_stream<0, 3>), gr, bl, (size_t)pad_lds, st, G);
  else if (prio == 1) hipLaunchKernelGGL((k_tn_stream<0, 1>), gr, bl, (size_t)pad_lds, st, G);
  else hipLaunchKernelGGL((k_tn_stream<0, 0>), gr, bl, (size_t)pad_lds, st, G);
  return check_launch("wd_gemm_tn_splitk_group (streamed)");
}

}  // namespace wd
