// wide_deep_amd/csrc/onehot_path.hip -- the sparse side of the Criteo-shaped step (ONE id per bag, row-record tables).
//
// Two kernels that take the input layer and the bucketing of the NEXT batch off the critical path of the train step
// (python/lib/dnn.py:83-91 input_layer, python/lib/linear.py:29-36 linear_model, python/lib/joint.py:233-248 the sparse
// apply ops whose duplicate-summing this bucketing prepares):
//
//   k_bucket_onehot    occurrence (b, s) -> row-range bucket of slot s, in ONE launch.  With exactly one id per bag every
//                      slot owns B occurrences, so slot s' region of `pairs` is [s*B, (s+1)*B) -- a static offset -- and a
//                      workgroup (slot s, chunk r) needs nothing from other slots: it histograms the whole id column of its
//                      slot in LDS (which gives the bucket starts AND what the chunks in front of it put into every bucket),
//                      then scatters its own chunk.  (wd_sparse_bucketize -- three launches around a [chunks][buckets] count
//                      matrix -- took 22 + 30 + 31 us for the same 3.4 MB; it stays for ragged batches.)  The last
//                      workgroup to finish lists the buckets largest-first for k_bucket_update's launch order.
//   k_prefetch_onehot  x[b, out_col_s ..] = rec[row].emb, wv[b*S + s] = rec[row].w (the wide weight sits in the line fetched
//                      for the row), numeric columns normalised into x -- the input layer of a batch as its own launch at
//                      full occupancy.  pipeline.StepGraph runs it for batch t+1 beside the tower of batch t; the rows that
//                      the update of batch t rewrites afterwards are patched by that update (k_bucket_update, `next`), so
//                      the tower of batch t+1 reads exactly what a gather after the update would have read.
#include "common.h"

namespace {

constexpr int MAX_NB = 8192;          // buckets (wd_bucket_max)
constexpr int ONEHOT_CHUNKS = 4;      // workgroups per slot

// counter[cls] += (lanes of the wavefront with this cls), one LDS atomic per distinct class and wavefront; returns the lane's
// position (old value + rank among its peers) or -1.  (3328 same-address LDS atomics -- the bucket sizes of a uniform batch all
// fall into two classes -- serialise: 28 us of the 33 this kernel first took.)
__device__ __forceinline__ int wave_class_add(int32_t *counter, int cls, bool valid) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
  int pos = -1;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int c = __shfl(cls, leader, 64);
    const unsigned long long peers = __ballot(valid && cls == c);
    int base = 0;
    if (lane == leader) base = atomicAdd(&counter[c], __popcll(peers));
    base = __shfl(base, leader, 64);
    if (valid && cls == c) pos = base + __popcll(peers & ((1ull << lane) - 1ull));
    todo &= ~peers;
  }
  return pos;
}

// slot s, chunk r = blockIdx.x % CH: examples [r * per, (r+1) * per)
__global__ void __launch_bounds__(512)
k_bucket_onehot(const wd_slot_t *__restrict__ slots, int32_t S, const int32_t *__restrict__ ids, int64_t sb, int64_t ss,
                int64_t B, int32_t nb, int32_t *__restrict__ start, uint64_t *__restrict__ pairs,
                int32_t *__restrict__ ticket, int32_t *__restrict__ zero_word) {
  WD_SIDE_PRIO();
  extern __shared__ int32_t lds_i[];
  __shared__ int32_t wsum[8];
  __shared__ int32_t is_last;
  const int t = threadIdx.x;
  const int s = blockIdx.x / ONEHOT_CHUNKS, r = blockIdx.x % ONEHOT_CHUNKS;
  if (zero_word && blockIdx.x == 0 && t < 2) zero_word[t] = 0;     // the two counters of wd_bucket_sort (next launch)
  const wd_slot_t sl = slots[s];
  const int32_t nbk = (s + 1 < S ? slots[s + 1].bucket_base : nb) - sl.bucket_base;   // buckets of this slot
  int32_t *total = lds_i, *base = lds_i + nbk;     // counts of the whole column / of the chunks in front of this one
  for (int i = t; i < 2 * nbk; i += 512) lds_i[i] = 0;
  __syncthreads();
  const int64_t per = (B + ONEHOT_CHUNKS - 1) / ONEHOT_CHUNKS;
  const int64_t c0 = (int64_t)r * per, c1 = c0 + per < B ? c0 + per : B;
  // pass 1: the whole id column of the slot (stride S: 852 KB in all, L2-resident, read by the 4 chunks of the slot)
  constexpr int U = 8;
  for (int64_t b0 = t; b0 < B; b0 += 512 * U) {
    int32_t id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t b = b0 + 512 * u;
      id[u] = b < B ? ids[b * sb + s * ss] : -1;       // (sb, ss) = (S, 1) example-major, (1, B) slot-major
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t b = b0 + 512 * u;
      if (id[u] < 0) continue;
      const int32_t bk = id[u] >> sl.bucket_shift;
      atomicAdd(&total[bk], 1);
      if (b < c0) atomicAdd(&base[bk], 1);
    }
  }
  __syncthreads();
  // exclusive scan of total[0..nbk): thread t owns E consecutive counters
  const int E = (nbk + 511) / 512;
  int32_t local = 0;
  for (int e = 0; e < E; ++e) {
    const int i = t * E + e;
    if (i < nbk) local += total[i];
  }
  int32_t incl = local;
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t v = __shfl_up(incl, off, 64);
    if ((t & 63) >= off) incl += v;
  }
  if ((t & 63) == 63) wsum[t >> 6] = incl;
  __syncthreads();
  int32_t run = (int32_t)((int64_t)s * B) + incl - local;
  for (int w = 0; w < (t >> 6); ++w) run += wsum[w];
  for (int e = 0; e < E; ++e) {
    const int i = t * E + e;
    if (i < nbk) {
      const int32_t c = total[i];
      // (device-scope store: the last workgroup reads it in THIS launch; a __threadfence() instead writes back the whole L2
      // of the XCD -- tens of MB of dirty rows / activations of the step -- and cost 20 of this kernel's 33 us)
      if (r == 0) __hip_atomic_store(&start[sl.bucket_base + i], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      base[i] += run;          // first position of THIS chunk's occurrences in bucket i
      run += c;
    }
  }
  if (r == 0 && s == S - 1 && t == 511)
    __hip_atomic_store(&start[nb], (int32_t)((int64_t)S * B), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  // pass 2: scatter the chunk (arrival order inside a bucket is arbitrary: k_bucket_update sorts every bucket on (row, bag))
  for (int64_t b0 = c0 + t; b0 < c1; b0 += 512 * U) {
    int32_t id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t b = b0 + 512 * u;
      id[u] = b < c1 ? ids[b * sb + s * ss] : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t b = b0 + 512 * u;
      if (id[u] < 0) continue;
      const int32_t p = atomicAdd(&base[id[u] >> sl.bucket_shift], 1);
      pairs[p] = ((uint64_t)(uint32_t)(sl.row_base + id[u]) << 32) | (uint32_t)(b * S + s);
    }
  }
  if (!ticket) return;     // no launch order wanted (wd_row_update walks the sorted pairs, not the buckets)
  // ---- the last workgroup to get here lists the buckets largest-first (start[nb + 2 ..]: launch order of k_bucket_update; a
  // bucket holding a Zipf head row takes several times longer than the rest and must not start last).  Binned by
  // floor(log2(size)); the order inside a bin is whatever the LDS atomics give -- it only schedules workgroups.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // this wavefront's stores are complete (no cache write-back)
  __syncthreads();
  if (t == 0) {
    const int32_t n = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = n == (int32_t)gridDim.x - 1;
    if (is_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
  }
  __syncthreads();
  if (!is_last) return;
  __shared__ int32_t cls_cnt[33], cls_pos[33];
  if (t < 33) cls_cnt[t] = 0;
  __syncthreads();
  // bucket sizes from the starts the other workgroups wrote: device-scope loads (past this CU's L1), all in flight at once
  constexpr int PER = MAX_NB / 512;
  int32_t cl[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = t + 512 * q;
    cl[q] = -1;
    if (i < nb) {
      const int32_t c = __hip_atomic_load(&start[i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                        __hip_atomic_load(&start[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cl[q] = c > 0 ? 32 - __builtin_clz((unsigned)c) : 0;
    }
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) wave_class_add(cls_cnt, cl[q], cl[q] >= 0);
  __syncthreads();
  if (t == 0) {
    int32_t run2 = 0;
    for (int c = 32; c >= 0; --c) {
      cls_pos[c] = run2;
      run2 += cls_cnt[c];
    }
  }
  __syncthreads();
  int32_t *order = start + nb + 2;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int p = wave_class_add(cls_pos, cl[q], cl[q] >= 0);
    if (p >= 0) order[p] = t + 512 * q;
  }
}

// LANES = dim / 4 lanes per bag, BPG bags per lane group (independent id -> record chains).  One bag per lane group and no LDS
// staging of the slot descriptors: the launch is 213 k random 128-byte lines for a C2 batch, and what it needs is wavefronts in
// flight -- 13 per SIMD this way; four bags per group behind a barrier (3 per SIMD) was 12.5 us instead of 10.
template <int LANES, int BPG>
__global__ void __launch_bounds__(256)
k_prefetch_onehot(const float *__restrict__ rec, int32_t rec_stride, const wd_slot_t *__restrict__ rslots, int32_t S,
                  const int32_t *__restrict__ ids, int64_t B, float *__restrict__ x, int64_t ldx, float *__restrict__ wv,
                  const float *__restrict__ dense, int64_t ld_dense, const wd_dense_col_t *__restrict__ cols,
                  int32_t ncols, int32_t gather_blocks, unsigned long long *__restrict__ span, int32_t wt) {
  const int t = threadIdx.x;
  // Issue priority over the tower's wavefronts (WD_PREFETCH_PRIO=0: off).  The launch runs beside the tower of the previous
  // batch, whose two wavefronts per SIMD issue an MFMA / LDS stream without gaps; a handful of loads per wavefront here only
  // need to get OUT: 26 -> 13 us for the launch inside the step, the step itself unchanged (profiles/r4_gather_instep.txt)
  if (wt & 2) __builtin_amdgcn_s_setprio(3);
  // diagnostics (bench.py: the duration of this launch AS IT RUNS INSIDE the pipelined step): every workgroup stores the
  // chip-wide realtime clock (100 MHz) at its start and end, span[2 * block], [2 * block + 1] (one min / max word for all of
  // them -- 7500 same-address atomics -- stretched the launch from 11 to 86 us)
  if (span && t == 0) span[2 * blockIdx.x] = (unsigned long long)wall_clock64();
  if ((int)blockIdx.x >= gather_blocks) {
    // numeric columns (python/lib/build_estimator.py:61-68 normalizers): one thread per (example, column)
    const int64_t i = ((int64_t)blockIdx.x - gather_blocks) * 256 + t;
    if (i < B * ncols) {
      const int64_t b = i / ncols;
      const int j = (int)(i - b * ncols);
      const wd_dense_col_t c = cols[j];
      float v = dense[b * ld_dense + j];
      if (c.kind == 1) v = (v - c.p0) / (c.p1 - c.p0);
      else if (c.kind == 2) v = (v - c.p0) / c.p1;
      else if (c.kind == 3) v = logf(v);
      x[b * ldx + c.out_col] = v;
    }
  } else {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = t % LANES;
    const int64_t grp = ((int64_t)blockIdx.x * 256 + t) / LANES;
    const int64_t nwork = B * S;
    int64_t w[BPG];
    int32_t id[BPG], sidx[BPG];
#pragma unroll
    for (int q = 0; q < BPG; ++q) {
      w[q] = grp * BPG + q;
      const int64_t wc = w[q] < nwork ? w[q] : nwork - 1;
      id[q] = ids[wc];
      sidx[q] = (int32_t)(wc % S);
    }
    int64_t off[BPG];
    int32_t col[BPG];
#pragma unroll
    for (int q = 0; q < BPG; ++q) {          // (two fields of a 48-byte descriptor: L1 hits, requested with the ids)
      off[q] = rslots[sidx[q]].emb_off;
      col[q] = rslots[sidx[q]].out_col;
    }
    f4 r[BPG];
    float wq[BPG];
#pragma unroll
    for (int q = 0; q < BPG; ++q) {
      r[q] = (f4)(0.f);
      wq[q] = 0.f;
      if (id[q] >= 0) {
        const float *p = rec + off[q] + (int64_t)id[q] * rec_stride;
        r[q] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p) + lane);
        if (lane == 0) wq[q] = __builtin_nontemporal_load(p + 4 * LANES);     // {w, z, n, -} behind the row, same line
      }
    }
#pragma unroll
    for (int q = 0; q < BPG; ++q) {
      if (w[q] >= nwork) continue;
      const int64_t b = w[q] / S;
      wd::store4(reinterpret_cast<float4 *>(x + b * ldx + col[q] + 4 * lane), make_float4(r[q].x, r[q].y, r[q].z, r[q].w), wt & 1);
      if (lane == 0) wd::store1(&wv[w[q]], wq[q], wt & 1);
    }
  }
  if (span) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the end stamp is taken when this workgroup's stores have completed
    __syncthreads();
    if (t == 0) span[2 * blockIdx.x + 1] = (unsigned long long)wall_clock64();
  }
}

// ======================================================================================================================
// Sorted row lists: everything of the sparse update that needs only the IDS -- bucketing, the sort of every bucket on
// (row, bag), the detection of heavily repeated rows, and which rows of the previous batch this batch reads too -- runs
// beside the tower of the previous step (pipeline.StepGraph); what is left between two towers is k_row_update, a flat
// gather / read-modify-write over the sorted pairs with two dependent memory round trips.
//
//   k_bucket_sort   one workgroup per bucket: pairs sorted in place (rank sort up to 512 pairs, bitonic up to 1024 in LDS,
//                   larger buckets in place in HBM/L2); segments of more than LONG_SEG occurrences are listed for the
//                   cooperative reduction; and -- `prev` -- for every pair of the PREVIOUS batch in the same row range:
//                   patch[i] = position of that row's first pair in THIS batch's sorted list, or -1.
//   k_row_update    4 lanes per sorted position: a position that starts a row (its predecessor holds another key) sums the
//                   row's gradient over its occurrences in ascending bag order and applies Adagrad (row) / FTRL ({w, z, n});
//                   then stores the new row into the next batch's prefetched x tile wherever patch[] says that batch reads
//                   it.  Listed long segments: one workgroup each, 64 lane groups + a fixed-shape tree; the last workgroup
//                   does bias_weights.  The arithmetic is k_bucket_update's, operation for operation (bit-identical rows).
constexpr int ROW_LONG_SEG = 32;
constexpr int LONG_WORKERS = 256;

__device__ __forceinline__ uint32_t key_of(uint64_t p) { return (uint32_t)(p >> 32); }

template <typename PtrT, int NT>
__device__ __forceinline__ void bitonic_sort_nt(PtrT p, int m) {     // all comparators ascending (virtual +inf padding never moves)
  int lgP = 0;
  while ((1 << lgP) < m) ++lgP;
  const int half = (1 << lgP) >> 1;
  for (int lk = 1; lk <= lgP; ++lk) {
    const int k = 1 << lk, kh = k >> 1;
    for (int tq = threadIdx.x; tq < half; tq += NT) {
      const int blk = tq >> (lk - 1), off = tq & (kh - 1);
      const int i1 = (blk << lk) + off, i2 = (blk << lk) + k - 1 - off;
      if (i2 < m) {
        const uint64_t a = p[i1], b = p[i2];
        if (a > b) { p[i1] = b; p[i2] = a; }
      }
    }
    __syncthreads();
    for (int lj = lk - 2; lj >= 0; --lj) {
      const int j = 1 << lj;
      for (int tq = threadIdx.x; tq < half; tq += NT) {
        const int i1 = ((tq >> lj) << (lj + 1)) | (tq & (j - 1)), i2 = i1 + j;
        if (i2 < m) {
          const uint64_t a = p[i1], b = p[i2];
          if (a > b) { p[i1] = b; p[i2] = a; }
        }
      }
      __syncthreads();
    }
  }
}

// rank[q] of the lane's q-th element (index t + NT q) among arr[0 .. m): number of smaller elements (all distinct).  NQ = elements
// per lane that exist -- a compile-time bound, so that a 300-pair bucket in a 1024-pair workgroup pays for two, not four.
// TIES (ragged bags, round 6: an id may repeat INSIDE a bag -- two equal (row, bag) pairs): equal elements rank by their index.
template <typename T, int NQ, int NT, bool TIES>
__device__ __forceinline__ void rank_loop(const T *arr, int m, int t, int *r) {
  T x[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    x[q] = t + NT * q < m ? arr[t + NT * q] : (T)~(T)0;
    r[q] = 0;
  }
  for (int j = 0; j < m; ++j) {
    const T y = arr[j];
#pragma unroll
    for (int q = 0; q < NQ; ++q) r[q] += (y < x[q] || (TIES && y == x[q] && j < t + NT * q)) ? 1 : 0;
  }
}

template <typename T, int MAXQ, int NT, bool TIES>
__device__ __forceinline__ void rank_dispatch_t(const T *arr, int m, int nq, int t, int *r) {
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) r[q] = 0;
  if (MAXQ >= 4 && nq > 2) {
    if (nq == 3) rank_loop<T, (MAXQ >= 3 ? 3 : 1), NT, TIES>(arr, m, t, r);
    else rank_loop<T, (MAXQ >= 4 ? 4 : 1), NT, TIES>(arr, m, t, r);
  } else if (MAXQ >= 2 && nq == 2) {
    rank_loop<T, (MAXQ >= 2 ? 2 : 1), NT, TIES>(arr, m, t, r);
  } else {
    rank_loop<T, 1, NT, TIES>(arr, m, t, r);
  }
}

template <typename T, int MAXQ, int NT>
__device__ __forceinline__ void rank_dispatch(const T *arr, int m, int nq, int t, int *r, bool ties) {
  if (ties) rank_dispatch_t<T, MAXQ, NT, true>(arr, m, nq, t, r);
  else rank_dispatch_t<T, MAXQ, NT, false>(arr, m, nq, t, r);
}

struct SortArgs {
  const int32_t *start;
  uint64_t *pairs;
  int32_t *long_list;        // [0] long-segment count, [1] big-bucket count, [2 + 2k], [3 + 2k] = (position, length)
  int32_t *big_list;         // buckets with more than SMALL_CAP pairs: left to the second launch
  const int32_t *pstart;
  const uint64_t *ppairs;
  int2 *ppatch;
  int32_t long_cap, bag_bits, nb, S;
  int64_t batch;
  int32_t prio;              // wavefront priority of the two launches (WD_SORT_PRIO; they run beside the tower)
  int32_t ties;              // ragged bags: equal pairs exist (an id repeated inside a bag) -- rank them by index, no bitmap path
};

// One bucket, CAP pairs rank-sorted in LDS (2 x CAP x 8 bytes).  false: the bucket is larger (SMALL launch: the caller lists it).
template <int CAP, bool SMALL, int NT>
__device__ __forceinline__ bool sort_bucket(const SortArgs &g, int bkt, uint64_t *lds_pairs, uint64_t *lds_in, uint32_t *s_kmin,
                                            uint32_t *s_kmax, int32_t *s_dom) {
  const int32_t *__restrict__ start = g.start;
  uint64_t *__restrict__ pairs = g.pairs;
  int32_t *__restrict__ long_list = g.long_list;
  const int32_t *__restrict__ pstart = g.pstart;
  const uint64_t *__restrict__ ppairs = g.ppairs;
  int2 *__restrict__ ppatch = g.ppatch;
  const int32_t long_cap = g.long_cap, bag_bits = g.bag_bits;
  constexpr int SORT_CAP = CAP;
  const int t = threadIdx.x;
  const int32_t s0 = start[bkt];
  const int m = start[bkt + 1] - s0;
  if (SMALL && m > CAP) return false;
  int32_t a0 = 0, mA = 0;
  if (ppatch) {
    a0 = pstart[bkt];
    mA = pstart[bkt + 1] - a0;
  }
  uint64_t a_cur = 0, a_prev = ~0ull;
  if (t < mA) {       // the previous batch's (sorted) pairs of this row range: in flight during the sort
    a_cur = ppairs[a0 + t];
    if (t > 0) a_prev = ppairs[a0 + t - 1];
  }
  const uint64_t *sp = lds_pairs;
  if (m <= 0) {
  } else if (m <= SORT_CAP) {
    // rank sort: position of element i = number of elements ordered before it; every lane reads the same LDS word per step
    // (broadcast, conflict-free), no barriers inside.  O(m^2 / 256) per lane; the 900-pair bucket of a Zipf head row is what
    // bounds the launch, so the pairs of a bucket are first packed into 32 bits -- (row - first row of the bucket) above the
    // bag index -- whenever they fit (a 64-bit compare costs four times a 32-bit one); 64-bit pairs otherwise.
    uint32_t kmin = 0xffffffffu, kmax = 0;
    for (int i = t; i < m; i += NT) {
      const uint64_t p = pairs[s0 + i];
      lds_in[i] = p;
      kmin = min(kmin, key_of(p));
      kmax = max(kmax, key_of(p));
    }
    for (int off = 32; off >= 1; off >>= 1) {
      kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
      kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
    }
    if ((t & 63) == 0) { s_kmin[t >> 6] = kmin; s_kmax[t >> 6] = kmax; }
    __syncthreads();
    kmin = s_kmin[0]; kmax = s_kmax[0];
    for (int w = 1; w < NT / 64; ++w) { kmin = min(kmin, s_kmin[w]); kmax = max(kmax, s_kmax[w]); }
    int r[SORT_CAP / NT];
    bool ranked = false;
    uint64_t o_pair[(512 + NT - 1) / NT];      // dominant-row path: the pairs of OTHER rows this lane places, and where
    int o_rank[(512 + NT - 1) / NT];
    int n_other = 0;
    if (!SMALL && SORT_CAP >= 1024 && NT >= 256 && g.batch <= 16384 && !g.ties) {
      // ---- a bucket dominated by ONE row K (a Zipf head row: ~780 of ~900 pairs): O(m) instead of O(m^2).  The pairs of a row
      // are distinct EXAMPLES (one id per bag), so their order by bag is a rank in a bitmap over the examples:
      // rank = (pairs of smaller rows) + popcount(bits below example b); the few other pairs rank among themselves.
      uint32_t *bits = reinterpret_cast<uint32_t *>(lds_pairs);       // [512] words: 16384 examples
      uint32_t *wpre = bits + 512;                                    // [512] pairs of K in the words before
      uint64_t *others = lds_pairs + 512;                             // [512] pairs of other rows
      const int lane = t & 63;
      const uint32_t K = key_of(lds_in[m >> 1]);
      const int nw = (int)((g.batch + 31) >> 5);
      if (t < 3) s_dom[t] = 0;
      for (int i = t; i < 2 * 512; i += NT) bits[i] = 0;             // (bits and wpre)
      __syncthreads();
      uint64_t x[SORT_CAP / NT];
      int cK = 0, cLt = 0;
#pragma unroll
      for (int q = 0; q < SORT_CAP / NT; ++q) {
        const bool v = t + NT * q < m;
        x[q] = v ? lds_in[t + NT * q] : ~0ull;
        cK += __popcll(__ballot(v && key_of(x[q]) == K));
        cLt += __popcll(__ballot(v && key_of(x[q]) < K));
      }
      if (lane == 0) { atomicAdd(&s_dom[0], cK); atomicAdd(&s_dom[1], cLt); }
      __syncthreads();
      const int c = s_dom[0], nlt = s_dom[1];
      if (2 * c >= m && m - c <= 512) {
#pragma unroll
        for (int q = 0; q < SORT_CAP / NT; ++q) {
          if (t + NT * q >= m) continue;
          if (key_of(x[q]) == K) {
            const uint32_t b = (uint32_t)x[q] / (uint32_t)g.S;
            atomicOr(&bits[b >> 5], 1u << (b & 31));
          } else {
            others[atomicAdd(&s_dom[2], 1)] = x[q];
          }
        }
        __syncthreads();
        {   // exclusive prefix of the words' popcounts: two words per lane of the first 256 lanes
          const int w0 = 2 * t;
          const bool live = t < 256;
          const int v0 = live && w0 < nw ? __popc(bits[w0]) : 0, v1 = live && w0 + 1 < nw ? __popc(bits[w0 + 1]) : 0;
          int incl = v0 + v1;
          for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(incl, off, 64);
            if (lane >= off) incl += u;
          }
          if (live && lane == 63) s_dom[8 + (t >> 6)] = incl;
          __syncthreads();
          if (live) {
            int base = incl - (v0 + v1);
            for (int w = 0; w < (t >> 6); ++w) base += s_dom[8 + w];
            wpre[w0] = base;
            wpre[w0 + 1] = base + v0;
          }
        }
        __syncthreads();
        const int no = m - c;
#pragma unroll
        for (int q = 0; q < SORT_CAP / NT; ++q) {
          r[q] = -1;                                   // (pairs of other rows are placed by the lanes below)
          if (t + NT * q >= m) continue;
          if (key_of(x[q]) == K) {
            const uint32_t b = (uint32_t)x[q] / (uint32_t)g.S;
            r[q] = nlt + (int)wpre[b >> 5] + __popc(bits[b >> 5] & ((1u << (b & 31)) - 1u));
          }
        }
        // the other rows' pairs: lane i takes others[i] and counts the smaller ones among them (no <= 512)
        for (int i = t; i < no; i += NT) {
          const uint64_t xo = others[i];
          int rr = key_of(xo) > K ? c : 0;
          for (int j = 0; j < no; ++j) rr += others[j] < xo ? 1 : 0;
          o_pair[i / NT] = xo;
          o_rank[i / NT] = rr;
        }
        ranked = true;
        n_other = no;
      }
      __syncthreads();      // (the scratch in lds_pairs is dead from here on)
    }
    const int nq = (m + NT - 1) / NT;          // elements per lane that exist: the compare loop is instantiated for 1 .. CAP / NT
    if (ranked) {
    } else if (bag_bits < 32 && (uint64_t)(kmax - kmin) < (1ull << (32 - bag_bits))) {
      uint32_t *ck = reinterpret_cast<uint32_t *>(lds_pairs);       // (the sorted pairs are written behind the barrier below)
      for (int i = t; i < m; i += NT) {
        const uint64_t p = lds_in[i];
        ck[i] = ((key_of(p) - kmin) << bag_bits) | (uint32_t)p;
      }
      __syncthreads();
      rank_dispatch<uint32_t, SORT_CAP / NT, NT>(ck, m, nq, t, r, g.ties != 0);
    } else {
      rank_dispatch<uint64_t, SORT_CAP / NT, NT>(lds_in, m, nq, t, r, g.ties != 0);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SORT_CAP / NT; ++q)
      if (t + NT * q < m && r[q] >= 0) {
        const uint64_t p = lds_in[t + NT * q];
        lds_pairs[r[q]] = p;
        pairs[s0 + r[q]] = p;
      }
#pragma unroll
    for (int q = 0; q < (512 + NT - 1) / NT; ++q)
      if (t + NT * q < n_other) {
        lds_pairs[o_rank[q]] = o_pair[q];
        pairs[s0 + o_rank[q]] = o_pair[q];
      }
    __syncthreads();
  } else {
    bitonic_sort_nt<uint64_t *, NT>(pairs + s0, m);      // (more than CAP pairs in one row range: in place in HBM / L2)
    sp = pairs + s0;
  }
  // rows repeated more than ROW_LONG_SEG times: (position, length) -> long_list[1 + 2k ..]; long_list[0] counts (zeroed by
  // wd_bucket_onehot).  Arrival order of the entries is free: every segment is reduced on its own.
  for (int i = t; i < m; i += NT) {
    const uint32_t key = key_of(sp[i]);
    if (i > 0 && key_of(sp[i - 1]) == key) continue;
    if (i + ROW_LONG_SEG >= m || key_of(sp[i + ROW_LONG_SEG]) != key) continue;
    int lo = i + ROW_LONG_SEG, hi = m;       // upper bound of the key
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (key_of(sp[mid]) == key) lo = mid + 1; else hi = mid;
    }
    const int32_t q = atomicAdd(&long_list[0], 1);
    if (q < long_cap) {       // (cannot overflow when long_list[0] was zeroed: at most nnz / 33 such rows)
      long_list[2 + 2 * q] = s0 + i;
      long_list[3 + 2 * q] = lo - i;
    }
  }
  // the previous batch's rows that this batch reads too: (position of the row's first pair here, number of its pairs)
  for (int i = t; i < mA; i += NT) {
    const uint64_t a = i == t ? a_cur : ppairs[a0 + i];
    const uint64_t ap = i == t ? a_prev : ppairs[a0 + i - 1];
    int2 res = make_int2(-1, 0);
    const uint32_t key = key_of(a);
    if ((i == 0 || key_of(ap) != key) && m > 0) {
      int lo = 0, hi = m;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key_of(sp[mid]) < key) lo = mid + 1; else hi = mid;
      }
      if (lo < m && key_of(sp[lo]) == key) {
        int l2 = lo + 1, h2 = m;
        while (l2 < h2) {
          const int mid = (l2 + h2) >> 1;
          if (key_of(sp[mid]) == key) l2 = mid + 1; else h2 = mid;
        }
        res = make_int2(s0 + lo, l2 - lo);
      }
    }
    ppatch[a0 + i] = res;
  }
  return true;
}

constexpr int SMALL_CAP = 256;       // 4 KB of LDS: seven workgroups per CU beside the one-launch tower (512: three, and the launch took 47 us for Zipf ids instead of 29)
constexpr int BIG_CAP = 1024;        // 16 KB
constexpr int BIG_WORKERS = 256;

// first launch: one workgroup per bucket; buckets of more than SMALL_CAP pairs (Zipf head rows, "missing value" tokens) are
// listed for the second launch, whose workgroups carry the LDS a 1024-pair rank sort needs -- with it in every workgroup, one
// workgroup per CU fits beside the tower and the launch took 62 us for a uniform batch instead of 10.
__global__ void __launch_bounds__(256) k_bucket_sort_small(SortArgs g) {
  __shared__ uint64_t lds_pairs[SMALL_CAP];
  __shared__ uint64_t lds_in[SMALL_CAP];
  __shared__ uint32_t s_kmin[4], s_kmax[4];
  if (g.prio) __builtin_amdgcn_s_setprio(3);
  const int bkt = blockIdx.x;
  if (!sort_bucket<SMALL_CAP, true, 256>(g, bkt, lds_pairs, lds_in, s_kmin, s_kmax, nullptr) && threadIdx.x == 0)
    g.big_list[atomicAdd(&g.long_list[1], 1)] = bkt;
}

// (512 lanes, two pairs per lane; a 1024-lane workgroup does not fit beside the tower's wavefronts -- registers -- and waited for
// them to finish: 55 us for nothing)
__global__ void __launch_bounds__(512) k_bucket_sort_big(SortArgs g) {
  __shared__ uint64_t lds_pairs[BIG_CAP];
  __shared__ uint64_t lds_in[BIG_CAP];
  __shared__ uint32_t s_kmin[8], s_kmax[8];
  __shared__ int32_t s_dom[16];
  if (g.prio) __builtin_amdgcn_s_setprio(3);
  const int nbig = g.long_list[1];
  for (int q = blockIdx.x; q < nbig; q += BIG_WORKERS) {
    sort_bucket<BIG_CAP, false, 512>(g, g.big_list[q], lds_pairs, lds_in, s_kmin, s_kmax, s_dom);
    __syncthreads();
  }
}


// ---- ragged bags: one WAVEFRONT per bucket, the pairs sorted in REGISTERS (bitonic network over 64 lanes x NQ registers, partners
// reached by shuffles) -- no LDS at all, so the launch runs beside the window tower, whose row tile leaves 13 KB of a CU's LDS
// (k_bucket_sort_small there: three workgroups per CU, 160 us instead of 40; profiles/r6_c4_flat_ragged_ab.txt).  Equal pairs (an id
// repeated inside a bag) are bitwise identical: their order is free.  Buckets of more than 256 pairs go to k_bucket_sort_big.
__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int m) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
  return ((uint64_t)hi << 32) | lo;
}

// position p = lane + 64 q (striped: coalesced loads / stores); ascending over p
template <int NQ>
__device__ __forceinline__ void wave_bitonic(uint64_t (&x)[4], int lane) {
  constexpr int N = 64 * NQ;
#pragma unroll
  for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j >= 1; j >>= 1) {
      if (j >= 64) {            // partner in another register of the same lane
        const int dq = j >> 6;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if ((q & dq) == 0) {
            const int p = lane + 64 * q;
            const bool asc = (p & k) == 0;
            const uint64_t a = x[q], b = x[q | dq];
            const bool sw = asc ? a > b : a < b;
            x[q] = sw ? b : a;
            x[q | dq] = sw ? a : b;
          }
        }
      } else {                  // partner in another lane, same register
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int p = lane + 64 * q;
          const bool asc = (p & k) == 0;
          const bool low = (lane & j) == 0;            // this lane holds the lower position of the pair
          const uint64_t a = x[q], b = shfl_xor64(a, j);
          const bool keep_min = asc == low;
          x[q] = keep_min ? (a < b ? a : b) : (a > b ? a : b);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_bucket_sort_wave(SortArgs g) {
  if (g.prio) __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63;
  const int bkt = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bkt >= g.nb) return;
  const int32_t s0 = g.start[bkt];
  const int m = g.start[bkt + 1] - s0;
  if (m <= 0) return;
  if (m > 256) {
    if (lane == 0) g.big_list[atomicAdd(&g.long_list[1], 1)] = bkt;
    return;
  }
  uint64_t x[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) x[q] = lane + 64 * q < m ? g.pairs[s0 + lane + 64 * q] : ~0ull;
  if (m <= 64) wave_bitonic<1>(x, lane);
  else if (m <= 128) wave_bitonic<2>(x, lane);
  else wave_bitonic<4>(x, lane);
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (lane + 64 * q < m) g.pairs[s0 + lane + 64 * q] = x[q];
  // rows with more than ROW_LONG_SEG occurrences: (position, length) -> long_list.  Position p's predecessor is (lane - 1, q) or
  // (63, q - 1); the pair ROW_LONG_SEG = 32 positions ahead is (lane + 32, q) or (lane - 32, q + 1).
  bool any = false;
  uint32_t keyq[4];
  bool headlong[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    keyq[q] = key_of(x[q]);
    const uint32_t up = (uint32_t)__shfl_up((int)keyq[q], 1, 64);
    const uint32_t wrap = q > 0 ? (uint32_t)__shfl((int)keyq[q > 0 ? q - 1 : 0], 63, 64) : 0u;
    const uint32_t prev = lane > 0 ? up : wrap;
    const uint32_t f0 = (uint32_t)__shfl((int)keyq[q], (lane + 32) & 63, 64);
    const uint32_t f1 = q < 3 ? (uint32_t)__shfl((int)key_of(x[q < 3 ? q + 1 : 3]), (lane + 32) & 63, 64) : 0u;
    const int p = lane + 64 * q;
    const uint32_t far = lane < 32 ? f0 : f1;
    const bool head = p < m && (p == 0 || prev != keyq[q]);
    headlong[q] = head && p + ROW_LONG_SEG < m && (lane < 32 || q < 3) && far == keyq[q];
    any = any || headlong[q];
  }
  if (!__ballot(any)) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned long long todo = __ballot(headlong[q]);
    while (todo) {              // a wavefront-uniform loop over the long rows' heads: count the row's pairs among all positions
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const uint32_t K = (uint32_t)__shfl((int)keyq[q], src, 64);
      int cnt = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) cnt += __popcll(__ballot(lane + 64 * r < m && keyq[r] == K));
      if (lane == src) {
        const int32_t e = atomicAdd(&g.long_list[0], 1);
        if (e < g.long_cap) {
          g.long_list[2 + 2 * e] = s0 + lane + 64 * q;
          g.long_list[3 + 2 * e] = cnt;
        }
      }
    }
  }
}

struct RowUpd {
  float *rec, *accum, *bias;
  const wd_slot_t *slots;
  const float *dx, *dlogit;
  const uint64_t *pairs;
  const int32_t *long_list;
  const int2 *patch;          // [nnz] (position, count) in the next batch's sorted pairs, or NULL
  const uint64_t *npairs;     // the next batch's sorted pairs
  float *nx, *nwv;
  int64_t ldx, nldx, nnz, batch;
  int32_t rec_stride, dim, S, flat_blocks, long_cap;
  float lr_emb, lr_w, l1, l2;
  int32_t wt;                 // rows / patches stored write-through (common.h)
  // emit mode (wd_row_grad_presum, sharded engine: sender-side unique): the summed gradient of a row is not applied but
  // written as ONE record [dim floats | dlogit sum | ..] at emit_out[emit_pos[first occurrence] * emit_rs]
  float *emit_out;
  const int32_t *emit_pos;
  int32_t emit_rs;
  // ragged bags (round 6, wd_row_update_ragged): an occurrence carries dx / len(bag) (combiner='mean'), the number of pairs is a
  // device value (what wd_sparse_bucketize scattered: ids < 0 and the small-table columns are not in the list)
  const int32_t *bag_offs;
  const int32_t *nnz_dev;
  int64_t ld_dl;              // dlogit of example b at dlogit[b * ld_dl] (1; the sharded owner's gradient records: their stride)
};

__device__ __forceinline__ void ftrl_row(float &w, float &z, float &n, float g, float lr, float l1, float l2) {
  const float n_new = n + g * g;
  z += g - (sqrtf(n_new) - sqrtf(n)) / lr * w;
  const float quad = sqrtf(n_new) / lr + 2.0f * l2;
  const float sgn = z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f);
  const float pre = (sgn * l1 - z) / quad;
  w = fabsf(z) > l1 ? pre : 0.f;
  n = n_new;
}

__device__ __forceinline__ float4 adagrad_row4(float4 &a, float4 w, float4 g, float lr) {
  a.x += g.x * g.x; a.y += g.y * g.y; a.z += g.z * g.z; a.w += g.w * g.w;
  w.x -= lr * g.x / sqrtf(a.x);
  w.y -= lr * g.y / sqrtf(a.y);
  w.z -= lr * g.z / sqrtf(a.z);
  w.w -= lr * g.w / sqrtf(a.w);
  return w;
}

__global__ void __launch_bounds__(256)
k_row_update(RowUpd u) {
  constexpr int MAXS = 128;
  __shared__ int64_t s_acc_off[MAXS];     // accumulator offset of the slot's first row, minus row_base * dim
  __shared__ int32_t s_col[MAXS];
  __shared__ float4 red[256];
  __shared__ float redw[256];
  __shared__ int32_t seg_ex[256];
  __shared__ float seg_scale[256];
  __shared__ float new_row[20];           // long segments: the row's new value (16 floats at most) + its wide weight
  const int t = threadIdx.x;
  const int S = u.S, D = u.dim, LG = D >> 2;
  const int64_t RS = u.rec_stride;
  // new value of a row -> the next batch's prefetched input, wherever that batch reads the row (its pairs from `pj` on: all
  // occurrences of a row are adjacent in the sorted list).  `row` = the D new floats, `wnew` the new wide weight; the callers
  // spread the run over `nl` lanes (lane `l` takes every nl-th pair): a hot row is read hundreds of times by the next batch.
  // (`first`: the lane's first pair of the run, loaded by the caller with the row -- no extra dependent round trip)
  auto patch_run = [&](int2 pj, int l, int nl, const float *row, float wnew, int32_t out_col, uint64_t first, bool have_first) {
    for (int k = l; k < pj.y; k += nl) {
      const int32_t bag2 = (int32_t)(uint32_t)((have_first && k == l) ? first : u.npairs[pj.x + k]);
      float *dst = u.nx + (int64_t)(bag2 / S) * u.nldx + out_col;
      for (int c = 0; c < LG; ++c) wd::store4(reinterpret_cast<float4 *>(dst + 4 * c), make_float4(row[4 * c], row[4 * c + 1], row[4 * c + 2], row[4 * c + 3]), u.wt);
      wd::store1(&u.nwv[bag2], wnew, u.wt);
    }
  };
  if ((int)blockIdx.x == LONG_WORKERS) {     // (grid: long-segment workers, this one, flat blocks) bias_weights: g = sum_b dlogit[b] (fixed-shape tree), dense FTRL
    // One workgroup, batch / 256 values per lane: eight loads in flight per trip (one load -> wait -> add per value was 32
    // dependent round trips at batch 8192, in the LAST workgroup of the grid -- the tail of the whole launch).  Same order of adds.
    if (!u.bias) return;
    float acc = 0.f;
    for (int64_t i0 = t; i0 < u.batch; i0 += 256 * 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = i0 + 256 * k < u.batch ? u.dlogit[i0 + 256 * k] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + 256 * k < u.batch) acc += v[k];
    }
    redw[t] = acc;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
      if (t < st) redw[t] += redw[t + st];
      __syncthreads();
    }
    if (t == 0) {
      float w = u.bias[0], z = u.bias[1], n = u.bias[2];
      ftrl_row(w, z, n, redw[0], u.lr_w, u.l1, u.l2);
      u.bias[0] = w; u.bias[1] = z; u.bias[2] = n;
    }
    return;
  }
  for (int i = t; i < S; i += 256) {
    const wd_slot_t sl = u.slots[i];
    s_acc_off[i] = sl.emb_off - sl.row_base * (int64_t)D;
    s_col[i] = sl.out_col;
  }
  if ((int)blockIdx.x < LONG_WORKERS) {
    // ---- long segments (the first workgroups of the grid: they take longest): the whole workgroup per row, 64 lane groups in parallel + fixed-shape tree (k_bucket_update's) -----
    __syncthreads();
    const int gidx = t >> 2, gl = t & 3;
    const int nl = u.long_list[0] < u.long_cap ? u.long_list[0] : u.long_cap;
    for (int q = nl - 1 - (int)blockIdx.x; q >= 0; q -= LONG_WORKERS) {    // from the END of the list: the large buckets' rows (the
                                                                          // second launch of wd_bucket_sort), the longest, are listed last
      const int64_t i = u.long_list[2 + 2 * q], e = i + u.long_list[3 + 2 * q];
      const uint64_t p0 = u.pairs[i];
      const uint32_t key = key_of(p0);
      const int32_t sidx = (int32_t)(uint32_t)p0 % S;
      const int32_t out_col = s_col[sidx];
      const int2 pj = u.patch ? u.patch[i] : make_int2(-1, 0);
      const int64_t off = s_acc_off[sidx] + (int64_t)key * D;
      const int64_t eoff = (int64_t)key * RS;
      const int c = gl;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int64_t j0 = i; j0 < e; j0 += 256) {
        const int n = e - j0 < 256 ? (int)(e - j0) : 256;
        if (t < n) {
          const int32_t bag = (int32_t)(uint32_t)u.pairs[j0 + t];
          seg_ex[t] = bag / S;
          float sc = 1.0f;
          if (u.bag_offs) {
            const int32_t len = u.bag_offs[bag + 1] - u.bag_offs[bag];
            sc = len > 1 ? 1.0f / (float)len : 1.0f;
          }
          seg_scale[t] = sc;
        }
        __syncthreads();
        if (c < LG) {
          float4 d[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int idx = gidx + 64 * k;
            d[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < n) d[k] = *reinterpret_cast<const float4 *>(u.dx + (int64_t)seg_ex[idx] * u.ldx + out_col + 4 * c);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int idx = gidx + 64 * k;
            if (idx < n) {
              const float scale = seg_scale[idx];
              g.x += d[k].x * scale; g.y += d[k].y * scale; g.z += d[k].z * scale; g.w += d[k].w * scale;
            }
          }
        }
        __syncthreads();
      }
      red[t] = g;
      __syncthreads();
      for (int st = 32; st >= 1; st >>= 1) {
        if (gidx < st) {
          float4 a = red[t], b = red[t + 4 * st];
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
          red[t] = a;
        }
        __syncthreads();
      }
      const int32_t eslot = u.emit_out ? u.emit_pos[(int32_t)(uint32_t)p0] : -1;
      if (u.emit_out) {
        if (gidx == 0 && c < LG && eslot >= 0) *reinterpret_cast<float4 *>(u.emit_out + (int64_t)eslot * u.emit_rs + 4 * c) = red[t];
      } else if (gidx == 0 && c < LG) {
        g = red[t];
        const float gg[4] = {g.x, g.y, g.z, g.w};
        for (int k2 = 0; k2 < 4; ++k2) {
          const int64_t o = off + 4 * c + k2;
          const float a = u.accum[o] + gg[k2] * gg[k2];
          u.accum[o] = a;
          float wn = u.rec[eoff + 4 * c + k2];
          wn -= u.lr_emb * gg[k2] / sqrtf(a);
          u.rec[eoff + 4 * c + k2] = wn;
          new_row[4 * c + k2] = wn;
        }
      }
      __syncthreads();
      float gw = 0.f;
      for (int64_t j = i + t; j < e; j += 1024) {   // four independent (pair -> dlogit) chains per lane and round
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int64_t jj = j + 256 * k;
          v[k] = jj < e ? u.dlogit[(int64_t)((int32_t)(uint32_t)u.pairs[jj] / S) * u.ld_dl] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) gw += v[k];
      }
      redw[t] = gw;
      __syncthreads();
      for (int st = 128; st >= 1; st >>= 1) {
        if (t < st) redw[t] += redw[t + st];
        __syncthreads();
      }
      if (u.emit_out) {
        if (t == 0 && eslot >= 0) u.emit_out[(int64_t)eslot * u.emit_rs + D] = redw[0];
      } else if (t == 0) {
        float4 r = *reinterpret_cast<float4 *>(u.rec + eoff + D);
        ftrl_row(r.x, r.y, r.z, redw[0], u.lr_w, u.l1, u.l2);
        *reinterpret_cast<float4 *>(u.rec + eoff + D) = r;
        new_row[16] = r.x;
      }
      __syncthreads();
      if (!u.emit_out && pj.y > 0) patch_run(pj, t, 256, new_row, new_row[16], out_col, 0, false);      // the whole workgroup over the run
      __syncthreads();
    }
    return;
  }
  // ---- flat part: 4 lanes per sorted position ------------------------------------------------------------------------
  const int gl = t & 3;
  const int64_t i = (((int64_t)blockIdx.x - LONG_WORKERS - 1) * 256 + t) >> 2;
  uint64_t pr = 0, prv = ~0ull, nxt = ~0ull, far = ~0ull;
  int2 pj = make_int2(-1, 0);
  int64_t nnz = u.nnz;
  if (u.nnz_dev) {          // ragged: the grid covers the capacity, the list holds *nnz_dev pairs
    const int64_t nd = *u.nnz_dev;
    nnz = nd < nnz ? nd : nnz;
  }
  if (i < nnz) {
    pr = u.pairs[i];
    if (i > 0) prv = u.pairs[i - 1];
    if (i + 1 < nnz) nxt = u.pairs[i + 1];
    if (i + ROW_LONG_SEG < nnz) far = u.pairs[i + ROW_LONG_SEG];
    if (u.patch) pj = u.patch[i];
  }
  __syncthreads();          // slot tables
  if (i >= nnz) return;
  const uint32_t key = key_of(pr);
  if (key_of(prv) == key) return;            // not the first occurrence of its row
  if (key_of(far) == key) return;            // more than ROW_LONG_SEG occurrences: listed by k_bucket_sort, reduced above
  int64_t e = i + 1;
  uint32_t mybag[ROW_LONG_SEG / 4];
  if (key_of(nxt) == key) {
    // sorted: the row's pairs end inside (i + 1, i + ROW_LONG_SEG].  The group's four lanes load the ROW_LONG_SEG pairs from i on
    // in ONE round (lane gl: i + gl, + 4, ...) and count the ones that carry the key -- a prefix; the bag indices the gradient
    // rounds below need arrive with them.  (Rounds 3-5: a binary search, five dependent round trips per row with more than one
    // occurrence before its first gradient row was requested: with skewed ids a third of the rows.)
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < ROW_LONG_SEG / 4; ++k) {
      const int64_t idx = i + gl + 4 * k;
      const uint64_t p = idx < nnz ? u.pairs[idx] : ~0ull;
      const bool mine = key_of(p) == key;
      mybag[k] = mine ? (uint32_t)p : 0u;
      cnt += mine ? 1 : 0;
    }
    cnt += __shfl_xor(cnt, 1, 64);
    cnt += __shfl_xor(cnt, 2, 64);
    e = i + cnt;
  }
  const int32_t bag0 = (int32_t)(uint32_t)pr;
  const int32_t sidx = bag0 % S;
  const int32_t out_col = s_col[sidx];
  const bool lane_emb = gl < LG;
  const int64_t off = s_acc_off[sidx] + (int64_t)key * D + 4 * gl;
  const int64_t eoff = (int64_t)key * RS + 4 * gl;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), w = a, r = a;
  uint64_t tgt0 = 0;
  if (gl < pj.y) tgt0 = u.npairs[pj.x + gl];       // the lane's first patch target: in flight with the row
  const bool emit = u.emit_out != nullptr;
  const int32_t eslot = emit ? u.emit_pos[bag0] : -1;
  if (lane_emb && !emit) {
    a = *reinterpret_cast<float4 *>(u.accum + off);
    w = *reinterpret_cast<float4 *>(u.rec + eoff);
  }
  if (gl == 0 && !emit) r = *reinterpret_cast<float4 *>(u.rec + (int64_t)key * RS + D);
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  float gw = 0.f;
  if (e - i == 1) {
    const int64_t b = bag0 / S;
    float scale = 1.0f;                      // one id per bag: the mean of one; ragged: dx / len(bag)
    if (u.bag_offs) {
      const int32_t len = u.bag_offs[bag0 + 1] - u.bag_offs[bag0];
      scale = len > 1 ? 1.0f / (float)len : 1.0f;
    }
    if (lane_emb) {
      const float4 d = *reinterpret_cast<const float4 *>(u.dx + b * u.ldx + out_col + 4 * gl);
      g.x += d.x * scale; g.y += d.y * scale; g.z += d.z * scale; g.w += d.w * scale;
    }
    if (gl == 0) gw += u.dlogit[b * u.ld_dl];
  } else {
    // 2 .. 32 occurrences: their bag indices arrived with the count above (lane gl of the group holds pairs i + gl, + 4, ...; a bag
    // travels to the other lanes by shuffle), then four gradient rows per round -- 1 + ceil(n / 4) dependent round trips (eight
    // per round cost the launch its occupancy: slower with uniform and with skewed ids); the adds keep ascending bag order
    // (bit-identical sums)
    const int n = (int)(e - i);
    const int l0 = (t & 63) & ~3;
#pragma unroll
    for (int c = 0; c < ROW_LONG_SEG / 4; ++c) {
      if (4 * c >= n) break;
      int32_t bag[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) bag[k] = (int32_t)__shfl((int)mybag[c], l0 + k, 64);
      float4 d[4];
      float v[4], scale[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        v[k] = 0.f;
        scale[k] = 1.0f;
        if (4 * c + k < n) {
          if (lane_emb) d[k] = *reinterpret_cast<const float4 *>(u.dx + (int64_t)(bag[k] / S) * u.ldx + out_col + 4 * gl);
          if (gl == 0) v[k] = u.dlogit[(int64_t)(bag[k] / S) * u.ld_dl];
          if (u.bag_offs) {
            const int32_t len = u.bag_offs[bag[k] + 1] - u.bag_offs[bag[k]];
            scale[k] = len > 1 ? 1.0f / (float)len : 1.0f;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * c + k < n) {
          g.x += d[k].x * scale[k]; g.y += d[k].y * scale[k]; g.z += d[k].z * scale[k]; g.w += d[k].w * scale[k];
          gw += v[k];
        }
    }
  }
  if (emit) {
    if (eslot >= 0) {
      if (lane_emb) *reinterpret_cast<float4 *>(u.emit_out + (int64_t)eslot * u.emit_rs + 4 * gl) = g;
      if (gl == 0) u.emit_out[(int64_t)eslot * u.emit_rs + D] = gw;
    }
    return;
  }
  float4 wn = w;
  if (lane_emb) {
    wn = adagrad_row4(a, w, g, u.lr_emb);
    wd::store4(reinterpret_cast<float4 *>(u.accum + off), a, u.wt);
    wd::store4(reinterpret_cast<float4 *>(u.rec + eoff), wn, u.wt);
  }
  if (gl == 0) {
    ftrl_row(r.x, r.y, r.z, gw, u.lr_w, u.l1, u.l2);
    wd::store4(reinterpret_cast<float4 *>(u.rec + (int64_t)key * RS + D), r, u.wt);
  }
  if (pj.y > 0) {     // every lane of the group gets the whole new row (shuffles) and takes every 4th pair of the run
    float row[16];
    const int l0 = (t & 63) & ~3;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      row[4 * c + 0] = __shfl(wn.x, l0 + c, 64); row[4 * c + 1] = __shfl(wn.y, l0 + c, 64);
      row[4 * c + 2] = __shfl(wn.z, l0 + c, 64); row[4 * c + 3] = __shfl(wn.w, l0 + c, 64);
    }
    const float wnew = __shfl(r.x, l0, 64);
    patch_run(pj, gl, 4, row, wnew, out_col, tgt0, true);
  }
}

}  // namespace

extern "C" int wd_row_grad_presum(const wd_slot_t *slots, int32_t S, int64_t batch, const float *dx, int64_t ldx,
                                  const float *dlogit, int32_t dim, const uint64_t *pairs, const int32_t *long_list,
                                  int32_t long_capacity, const int32_t *pos, float *out, int32_t row_stride, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(slots && dx && dlogit && pairs && long_list && pos && out, "null pointer");
  WD_REQUIRE(S > 0 && S <= 128 && (dim == 4 || dim == 8 || dim == 16) && row_stride >= dim + 1 && row_stride % 4 == 0,
             "record = [dim | dlogit ..], dim in {4, 8, 16}, row_stride % 4 == 0, S <= 128");
  WD_REQUIRE(ldx % 4 == 0, "ldx % 4 == 0");
  RowUpd u{};
  u.ld_dl = 1;
  u.slots = slots; u.dx = dx; u.dlogit = dlogit; u.pairs = pairs; u.long_list = long_list; u.long_cap = long_capacity;
  u.ldx = ldx; u.nnz = batch * S; u.batch = batch; u.rec_stride = row_stride; u.dim = dim; u.S = S;
  u.emit_out = out; u.emit_pos = pos; u.emit_rs = row_stride;
  u.flat_blocks = (int32_t)wd::ceil_div(u.nnz * 4, 256);
  hipLaunchKernelGGL(k_row_update, dim3((unsigned)(u.flat_blocks + LONG_WORKERS + 1)), dim3(256), 0, wd::as_stream(stream), u);
  return wd::check_launch("wd_row_grad_presum");
}

extern "C" int wd_bucket_onehot(const wd_slot_t *slots, int32_t S, const int32_t *ids, int32_t ids_slot_major, int64_t batch,
                                int32_t *bucket_start, uint64_t *pairs, int32_t nbuckets, int32_t max_slot_buckets,
                                int32_t *ticket, int32_t *zero_word, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(slots && ids && bucket_start && pairs, "null pointer");
  WD_REQUIRE(S > 0 && nbuckets > 0 && nbuckets <= MAX_NB && max_slot_buckets > 0 && max_slot_buckets <= nbuckets,
             "bad bucket geometry");
  WD_REQUIRE(batch * S < ((int64_t)1 << 31), "batch * S must fit 31 bits");
  hipLaunchKernelGGL(k_bucket_onehot, dim3((unsigned)(S * ONEHOT_CHUNKS)), dim3(512), (size_t)max_slot_buckets * 8,
                     wd::as_stream(stream), slots, S, ids, ids_slot_major ? (int64_t)1 : (int64_t)S,
                     ids_slot_major ? batch : (int64_t)1, batch, nbuckets, bucket_start, pairs, ticket, zero_word);
  return wd::check_launch("wd_bucket_onehot");
}

extern "C" int64_t wd_prefetch_onehot_blocks(int64_t batch, int32_t S, int32_t dim, int32_t ncols) {
  if (batch <= 0 || S <= 0 || dim <= 0) return 0;
  return wd::ceil_div(batch * S * (dim / 4), 256) + (ncols > 0 ? wd::ceil_div(batch * ncols, 256) : 0);
}

extern "C" int wd_prefetch_onehot(const float *rec, int32_t rec_stride, int32_t dim, const wd_slot_t *rec_slots, int32_t S,
                                  const int32_t *ids, int64_t batch, float *x, int64_t ldx, float *wide_vals,
                                  const float *dense, int64_t ld_dense, const wd_dense_col_t *dense_cols, int32_t ncols,
                                  void *span, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(rec && rec_slots && ids && x && wide_vals, "null pointer");
  WD_REQUIRE(S > 0 && S <= 128, "1 <= S <= 128");
  WD_REQUIRE((dim == 4 || dim == 8 || dim == 16) && rec_stride % 4 == 0 && rec_stride >= dim + 4, "record = [dim | w z n -], dim in {4, 8, 16}");
  WD_REQUIRE(ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "x must be 16-byte aligned with ldx % 4 == 0");
  WD_REQUIRE(ncols == 0 || (dense && dense_cols), "numeric columns need dense + descriptors");
  hipStream_t st = wd::as_stream(stream);
  constexpr int BPG = 1;     // bags per lane group: 2 / 4 (more requests per wavefront, fewer wavefronts) are slower alone AND
                             // beside the tower (profiles/r4_gather_instep.txt)
  static const int prio = (getenv("WD_PREFETCH_PRIO") && atoi(getenv("WD_PREFETCH_PRIO")) == 0) ? 0 : 2;
  const int lanes = dim / 4;
  const int64_t groups = wd::ceil_div(batch * S, BPG);
  const int gb = (int)wd::ceil_div(groups * lanes, 256);
  const int db = ncols > 0 ? (int)wd::ceil_div(batch * ncols, 256) : 0;
#define WD_LAUNCH_PF(L)                                                                                                     \
  hipLaunchKernelGGL((k_prefetch_onehot<L, BPG>), dim3((unsigned)(gb + db)), dim3(256), 0, st, rec, rec_stride, rec_slots, \
                     S, ids, batch, x, ldx, wide_vals, dense, ld_dense, dense_cols, ncols, gb,                              \
                     static_cast<unsigned long long *>(span), (int32_t)((wd::wt_mask() & WD_WT_PREFETCH ? 1 : 0) | prio))
  if (lanes == 4) WD_LAUNCH_PF(4);
  else if (lanes == 2) WD_LAUNCH_PF(2);
  else WD_LAUNCH_PF(1);
#undef WD_LAUNCH_PF
  return wd::check_launch("wd_prefetch_onehot");
}

extern "C" int wd_bucket_sort(const int32_t *bucket_start, uint64_t *pairs, int32_t nbuckets, int32_t *long_list,
                              int32_t long_capacity, int32_t *big_list, int64_t batch, int32_t S,
                              const int32_t *prev_bucket_start, const uint64_t *prev_pairs, int32_t *prev_patch,
                              wd_stream_t stream) {
  WD_REQUIRE(bucket_start && pairs && long_list && big_list && nbuckets > 0 && nbuckets <= MAX_NB, "null pointer / bad bucket geometry");
  WD_REQUIRE(!prev_patch || (prev_bucket_start && prev_pairs), "prev_patch needs the previous batch's buckets");
  WD_REQUIRE(long_capacity > 0 && batch > 0 && S > 0, "long_capacity, batch, S");
  const int64_t nnz = batch * S;
  SortArgs g{};
  g.S = S; g.batch = batch;
  g.start = bucket_start; g.pairs = pairs; g.long_list = long_list; g.big_list = big_list; g.pstart = prev_bucket_start;
  g.ppairs = prev_pairs; g.ppatch = reinterpret_cast<int2 *>(prev_patch); g.long_cap = long_capacity; g.nb = nbuckets;
  static const int sort_prio = getenv("WD_SORT_PRIO") ? atoi(getenv("WD_SORT_PRIO")) : 1;
  g.prio = sort_prio;
  g.bag_bits = 1;
  while (g.bag_bits < 32 && ((int64_t)1 << g.bag_bits) < nnz) ++g.bag_bits;
  hipStream_t st = wd::as_stream(stream);
  hipLaunchKernelGGL(k_bucket_sort_small, dim3((unsigned)nbuckets), dim3(256), 0, st, g);
  hipLaunchKernelGGL(k_bucket_sort_big, dim3(BIG_WORKERS), dim3(512), 0, st, g);
  return wd::check_launch("wd_bucket_sort");
}

// ragged bags (multi-hot batches on row records, round 6): the buckets of wd_sparse_bucketize sorted in place (equal pairs -- an id
// repeated inside a bag -- allowed), long rows listed; no patch lists (no prefetched input layer on this path)
extern "C" int wd_bucket_sort_ragged(const int32_t *bucket_start, uint64_t *pairs, int32_t nbuckets, int32_t *long_list,
                                     int32_t long_capacity, int32_t *big_list, int64_t batch, int32_t S, int64_t nnz_capacity,
                                     wd_stream_t stream) {
  WD_REQUIRE(bucket_start && pairs && long_list && big_list && nbuckets > 0 && nbuckets <= MAX_NB, "null pointer / bad bucket geometry");
  WD_REQUIRE(long_capacity > 0 && batch > 0 && S > 0 && nnz_capacity > 0, "long_capacity, batch, S, nnz_capacity");
  SortArgs g{};
  g.S = S; g.batch = batch;
  g.start = bucket_start; g.pairs = pairs; g.long_list = long_list; g.big_list = big_list; g.long_cap = long_capacity; g.nb = nbuckets;
  static const int sort_prio = getenv("WD_SORT_PRIO") ? atoi(getenv("WD_SORT_PRIO")) : 1;
  g.prio = sort_prio;
  g.ties = 1;
  g.bag_bits = 1;
  while (g.bag_bits < 32 && ((int64_t)1 << g.bag_bits) < batch * S) ++g.bag_bits;
  hipStream_t st = wd::as_stream(stream);
  if (hipMemsetAsync(long_list, 0, 8, st) != hipSuccess) {      // the two counters wd_bucket_onehot zeroes on the one-id-per-bag path
    wd::set_error("wd_bucket_sort_ragged: hipMemsetAsync failed");
    return WD_ERR_LAUNCH;
  }
  static const int wave_sort = getenv("WD_SORT_WAVE") ? atoi(getenv("WD_SORT_WAVE")) : 1;
  if (wave_sort) hipLaunchKernelGGL(k_bucket_sort_wave, dim3((unsigned)wd::ceil_div(nbuckets, 4)), dim3(256), 0, st, g);
  else hipLaunchKernelGGL(k_bucket_sort_small, dim3((unsigned)nbuckets), dim3(256), 0, st, g);
  hipLaunchKernelGGL(k_bucket_sort_big, dim3(BIG_WORKERS), dim3(512), 0, st, g);
  return wd::check_launch("wd_bucket_sort_ragged");
}

extern "C" int wd_row_update_ragged(float *rec, int32_t rec_stride, int32_t dim, float *emb_accum, float *bias_wzn,
                                    const wd_slot_t *slots, int32_t S, int64_t batch, const int32_t *bag_offs, const float *dx,
                                    int64_t ldx, const float *dlogit, int64_t ld_dlogit, float lr_emb, float lr_wide, float l1,
                                    float l2, const uint64_t *pairs, int64_t nnz_capacity, const int32_t *nnz_dev,
                                    const int32_t *long_list, int32_t long_capacity, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(rec && emb_accum && slots && dx && dlogit && pairs && long_list && bag_offs && nnz_dev, "null pointer");
  WD_REQUIRE(ld_dlogit >= 1 && (ld_dlogit == 1 || !bias_wzn), "ld_dlogit >= 1; the bias update reads a contiguous dlogit");
  WD_REQUIRE(S > 0 && S <= 128 && (dim == 4 || dim == 8 || dim == 16) && rec_stride % 4 == 0 && rec_stride >= dim + 4,
             "record = [dim | w z n -], dim in {4, 8, 16}, S <= 128");
  WD_REQUIRE(nnz_capacity > 0 && nnz_capacity < ((int64_t)1 << 31), "nnz_capacity");
  RowUpd u{};
  u.ld_dl = 1;
  u.rec = rec; u.accum = emb_accum; u.bias = bias_wzn; u.slots = slots; u.dx = dx; u.dlogit = dlogit; u.pairs = pairs;
  u.long_list = long_list; u.long_cap = long_capacity; u.ldx = ldx; u.nnz = nnz_capacity; u.batch = batch; u.rec_stride = rec_stride;
  u.dim = dim; u.S = S; u.lr_emb = lr_emb; u.lr_w = lr_wide; u.l1 = l1; u.l2 = l2;
  u.bag_offs = bag_offs; u.nnz_dev = nnz_dev; u.ld_dl = ld_dlogit;
  u.flat_blocks = (int32_t)wd::ceil_div(u.nnz * 4, 256);
  u.wt = wd::wt_mask() & WD_WT_ROW_UPDATE ? 1 : 0;
  hipLaunchKernelGGL(k_row_update, dim3((unsigned)(u.flat_blocks + LONG_WORKERS + 1)), dim3(256), 0, wd::as_stream(stream), u);
  return wd::check_launch("wd_row_update_ragged");
}

extern "C" int wd_row_update(float *rec, int32_t rec_stride, int32_t dim, float *emb_accum, float *bias_wzn,
                             const wd_slot_t *slots, int32_t S, int64_t batch, const float *dx, int64_t ldx,
                             const float *dlogit, float lr_emb, float lr_wide, float l1, float l2, const uint64_t *pairs,
                             const int32_t *long_list, int32_t long_capacity, const int32_t *patch,
                             const wd_apply_next_t *next, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(rec && emb_accum && slots && dx && dlogit && pairs && long_list, "null pointer");
  WD_REQUIRE(S > 0 && S <= 128 && (dim == 4 || dim == 8 || dim == 16) && rec_stride % 4 == 0 && rec_stride >= dim + 4,
             "record = [dim | w z n -], dim in {4, 8, 16}, S <= 128");
  WD_REQUIRE(!patch || (next && next->pairs && next->x && next->wide_vals && next->ldx > 0),
             "patch needs the next batch's sorted pairs, x tile and wide-weight list");
  RowUpd u{};
  u.ld_dl = 1;
  u.rec = rec; u.accum = emb_accum; u.bias = bias_wzn; u.slots = slots; u.dx = dx; u.dlogit = dlogit; u.pairs = pairs;
  u.long_list = long_list; u.long_cap = long_capacity; u.patch = reinterpret_cast<const int2 *>(patch); u.ldx = ldx; u.nnz = batch * S; u.batch = batch; u.rec_stride = rec_stride;
  u.dim = dim; u.S = S; u.lr_emb = lr_emb; u.lr_w = lr_wide; u.l1 = l1; u.l2 = l2;
  if (patch) { u.npairs = next->pairs; u.nx = next->x; u.nwv = next->wide_vals; u.nldx = next->ldx; }
  u.flat_blocks = (int32_t)wd::ceil_div(u.nnz * 4, 256);
  u.wt = wd::wt_mask() & WD_WT_ROW_UPDATE ? 1 : 0;
  static const int pad_lds = getenv("WD_UPD_LDS") ? atoi(getenv("WD_UPD_LDS")) : 0;   // diagnostics: extra LDS bytes = fewer workgroups per CU
  hipLaunchKernelGGL(k_row_update, dim3((unsigned)(u.flat_blocks + LONG_WORKERS + 1)), dim3(256), (size_t)pad_lds, wd::as_stream(stream), u);
  return wd::check_launch("wd_row_update");
}
