// wide_deep_amd/csrc/dist_exchange.hip -- device side of the row-sharded multi-GPU step.
//
// The reference shards big variables over parameter servers and moves rows over TF gRPC
// (python/lib/joint.py:140-143, python/train.py:202-225).  Here every GPU owns the rows id % W == rank
// (local row = id / W) of every table and a step exchanges
//     A  requested local rows        int32  [W][cap]                 requester -> owner
//     B  embedding rows + wide value float  [W][cap][RS]             owner -> requester
//     C  per-occurrence gradients    float  [W][cap][RS]             requester -> owner
// with RCCL all-to-all over xGMI.  All three use EQUAL, FIXED per-peer segments of `cap` entries (unused entries
// carry row -1), so no split size ever has to travel to the host: the whole step stays asynchronous on the stream.
// A peer receiving more than `cap` requests raises a device flag that the host checks lazily.
//
// Kernels (requester side / owner side):
//   k_route_count / k_route_scan / k_route_scatter   occurrence -> (owner, slot in the owner's segment); deterministic:
//                       ranks follow (bag-batch, position-in-bag, wave, lane) order, never atomics arrival order
//   k_owner_gather      B payload: emb[lrow, 0..D) and wide[lrow].w for every received request
//   k_grad_pack         C payload: dx[b, col..col+D) / len(bag) and dlogit[b] per occurrence
// Pooling of B on the requester is wd_embag_fwd / wd_wide_fwd with a row stride; the owner update of C is
// wd_sparse_bwd_fused on the received (row, gradient) list.
#include "common.h"

namespace {

constexpr int MAX_W = 16;       // ranks per node
constexpr int MAX_CHUNKS = 128; // workgroups of the routing kernels

// chunk c = bags [c*bags_per_chunk, ...).  Pass 1 (count) and pass 3 (scatter) walk the chunk identically.
// Per (batch of 256 bags, position k in the bag): lanes vote per owner, waves are ordered through LDS.
template <bool SCATTER>
__global__ void __launch_bounds__(256)
k_route(const wd_slot_t *__restrict__ slots, int32_t S, int32_t W, const int32_t *__restrict__ ids,
        const int32_t *__restrict__ bag_offs, int64_t nbags, int64_t bags_per_chunk, int32_t cap,
        int32_t *__restrict__ cntm /* [chunks][W] counts (pass 1) / exclusive chunk prefixes (pass 3) */,
        int32_t *__restrict__ send_rows, int32_t *__restrict__ pos) {
  WD_SIDE_PRIO();
  __shared__ int32_t wcount[4][MAX_W];
  __shared__ int32_t running[MAX_W];
  __shared__ int32_t any_left;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (!SCATTER) {   // count pass: every workgroup also clears its share of the send segments (unused entries = row -1)
    const int64_t n = (int64_t)W * cap;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t i1 = ((int64_t)blockIdx.x + 1) * per < n ? ((int64_t)blockIdx.x + 1) * per : n;
    for (int64_t i = (int64_t)blockIdx.x * per + t; i < i1; i += 256) send_rows[i] = -1;
  }
  if (t < MAX_W) running[t] = (SCATTER && t < W) ? cntm[(int64_t)blockIdx.x * W + t] : 0;
  __syncthreads();
  const int64_t b0 = (int64_t)blockIdx.x * bags_per_chunk;
  const int64_t b1 = b0 + bags_per_chunk < nbags ? b0 + bags_per_chunk : nbags;
  for (int64_t base = b0; base < b1; base += 256) {
    const int64_t bag = base + t;
    int32_t j0 = 0, j1 = 0;
    int64_t rbase = 0;
    if (bag < b1) {
      const wd_slot_t &sl = slots[bag % S];
      j0 = bag_offs[bag];
      j1 = (sl.flags & WD_SLOT_F_SMALL) ? j0 : bag_offs[bag + 1];     // a replicated column (small_tables.hip): nothing to request
      rbase = sl.row_base;
    }
    for (int32_t k = 0;; ++k) {
      if (t == 0) any_left = 0;
      __syncthreads();
      const bool live = j0 + k < j1;
      if (live) any_left = 1;
      int32_t id = 0, o = -1;
      if (live) {
        id = ids[j0 + k];
        o = id % W;
      }
      // per-owner vote inside the wave
      int32_t my_rank = 0;
      for (int w = 0; w < W; ++w) {
        const unsigned long long m = __ballot(o == w);
        if (o == w) my_rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[wave][w] = __popcll(m);
      }
      __syncthreads();
      if (!any_left) break;
      if (live) {
        int32_t before = running[o];
        for (int w2 = 0; w2 < wave; ++w2) before += wcount[w2][o];
        const int32_t p = before + my_rank;   // index inside the owner's segment
        if (SCATTER) {
          if (p < cap) {
            send_rows[(int64_t)o * cap + p] = (int32_t)(rbase + id / W);
            pos[j0 + k] = o * cap + p;
          } else {
            pos[j0 + k] = -1;
          }
        }
      }
      __syncthreads();
      if (t < W) running[t] += wcount[0][t] + wcount[1][t] + wcount[2][t] + wcount[3][t];
      __syncthreads();
    }
  }
  if (!SCATTER && t < W) cntm[(int64_t)blockIdx.x * W + t] = running[t];
}

// one workgroup: exclusive prefix over the chunks per owner, overflow check.  One wavefront-sized lane group per owner walks
// the chunk counts 64 at a time (two rounds for MAX_CHUNKS = 128): a wave scan instead of a chain of dependent loads.
__global__ void __launch_bounds__(1024)
k_route_scan(int32_t *__restrict__ cntm, int32_t nchunks, int32_t W, int32_t cap, int32_t *__restrict__ send_rows,
             int32_t *__restrict__ overflow, int32_t *__restrict__ peer_counts) {
  WD_SIDE_PRIO();
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;   // wavefront w = owner w (W <= MAX_W = 16 wavefronts)
  if (w < W) {
    int32_t run = 0;
    for (int c0 = 0; c0 < nchunks; c0 += 64) {
      const int c = c0 + lane;
      const int32_t v = c < nchunks ? cntm[(int64_t)c * W + w] : 0;
      int32_t incl = v;
      for (int off = 1; off < 64; off <<= 1) {
        const int32_t u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
      }
      if (c < nchunks) cntm[(int64_t)c * W + w] = run + incl - v;
      run += __shfl(incl, 63, 64);
    }
    if (lane == 0) {
      peer_counts[w] = run;
      if (run > cap) atomicMax(overflow, run);
    }
  }
  if (nchunks == 0) {   // empty batch: no count pass ran
    const int64_t n = (int64_t)W * cap;
    for (int64_t i = t; i < n; i += 1024) send_rows[i] = -1;
  }
}

__global__ void __launch_bounds__(256)
k_fill_i32(int32_t *__restrict__ p, int32_t v, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) p[i] = v;
}

// 4 lanes x float4 per (request, 16 floats); lane group loops over D/16 chunks.  Requests with row -1 are skipped
// (their slots are never read by the requester).
__global__ void __launch_bounds__(256)
k_owner_gather(const float *__restrict__ emb, int64_t n_emb_rows, int32_t D, const float *__restrict__ wide,
               const int32_t *__restrict__ rows, int64_t n, float *__restrict__ out, int32_t RS) {
  WD_SIDE_PRIO();
  const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
  const int lane = threadIdx.x & 3;
  if (r >= n) return;
  const int32_t lrow = rows[r];
  if (lrow < 0) return;
  float *o = out + r * RS;
  if (emb && lrow < n_emb_rows) {
    const float4 *src = reinterpret_cast<const float4 *>(emb + (int64_t)lrow * D);
    for (int c = lane; c < (D >> 2); c += 4) *reinterpret_cast<float4 *>(o + 4 * c) = src[c];
  }
  if (wide && lane == 0) o[emb ? D : 0] = wide[(int64_t)lrow * 4];
}

// row-record layout on the owner (engine.py: rec[row] = [emb (D f32) | w z n - | pad]): a request is ONE copy of the first
// `nvec` float4 of the record -- the embedding row and the wide line behind it arrive in the same 128-byte line.
__global__ void __launch_bounds__(256)
k_owner_gather_rec(const float *__restrict__ rec, int32_t rec_stride, int32_t nvec, const int32_t *__restrict__ rows,
                   int64_t n, float *__restrict__ out, int32_t RS) {
  WD_SIDE_PRIO();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = i / nvec;
  const int c = (int)(i - r * nvec);
  if (r >= n) return;
  const int32_t lrow = rows[r];
  if (lrow < 0) return;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(rec + (int64_t)lrow * rec_stride) + c);
  *(reinterpret_cast<f4 *>(out + r * RS) + c) = v;
}

// per bag (b, s): every occurrence j gets out[pos[j]*RS + 0..D) = dx[b, col..] * (1/len) and, for wide slots,
// out[pos[j]*RS + D] = dlogit[b].  D = width of the exchanged record = the LARGEST embedding dim; a slot of a smaller dim
// fills the rest with zeros (the owner keeps its rows padded to D: zero gradient, the pad columns never move)
__global__ void __launch_bounds__(256)
k_grad_pack(const wd_slot_t *__restrict__ slots, int32_t S, const int32_t *__restrict__ bag_offs,
            const int32_t *__restrict__ pos, int64_t nbags, const float *__restrict__ dx, int64_t ldx,
            const float *__restrict__ dlogit, int32_t D, int32_t RS, float *__restrict__ out) {
  WD_SIDE_PRIO();
  const int64_t bag = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
  const int lane = threadIdx.x & 3;
  if (bag >= nbags) return;
  const int64_t b = bag / S;
  const wd_slot_t sl = slots[bag - b * S];
  const int32_t j0 = bag_offs[bag], j1 = bag_offs[bag + 1];
  if (j1 <= j0 || (sl.flags & WD_SLOT_F_SMALL)) return;      // (a replicated column's gradient is all-reduced, not exchanged)
  const bool is_emb = dx && sl.kind == WD_SLOT_EMBEDDING;
  const int dq = is_emb ? (sl.dim >> 2) : 0;
  const float scale = (j1 - j0) > 1 ? 1.0f / (float)(j1 - j0) : 1.0f;
  const float dl = (dlogit && sl.wide) ? dlogit[b] : 0.f;
  for (int32_t j = j0; j < j1; ++j) {
    const int32_t p = pos[j];
    if (p < 0) continue;
    float *o = out + (int64_t)p * RS;
    if (dx) {
      for (int c = lane; c < (D >> 2); c += 4) {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < dq) {
          g = *reinterpret_cast<const float4 *>(dx + b * ldx + sl.out_col + 4 * c);
          g.x *= scale; g.y *= scale; g.z *= scale; g.w *= scale;
        }
        *reinterpret_cast<float4 *>(o + 4 * c) = g;
      }
    }
    if (dlogit && lane == 0) o[dx ? D : 0] = dl;
  }
}


// ---- sender-side unique (wd_route_unique): one request per DISTINCT (owner, local row) of the batch ---------------------------
// Input: the batch's occurrences sorted on (key, occurrence) -- wd_bucket_onehot + wd_bucket_sort with requester-side slot
// descriptors whose row space is  key = W * local_row_base(slot) + id,  so that  owner = key % W  and  local row = key / W
// with no table lookup.  A position whose predecessor holds another key starts a row ("head"): heads get the next entry of
// their owner's segment in sorted order (deterministic: wave votes ordered through LDS, chunk prefixes from k_route_scan), every
// other occurrence of the row points at the same entry.  With skewed ids a batch asks for less than half as many rows as it has
// occurrences (Zipf(1.05), 8192 x 26: 46 %), and sends one pre-summed gradient per row (wd_row_grad_presum).
__device__ __forceinline__ uint32_t uq_key(uint64_t p) { return (uint32_t)(p >> 32); }

template <bool SCATTER>
__global__ void __launch_bounds__(256)
k_route_unique(const uint64_t *__restrict__ pairs, int64_t n, int64_t per_chunk, int32_t W, int32_t cap,
               int32_t *__restrict__ cntm, int32_t *__restrict__ send_rows, int32_t *__restrict__ pos) {
  WD_SIDE_PRIO();
  __shared__ int32_t wcount[4][MAX_W];
  __shared__ int32_t running[MAX_W];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (!SCATTER) {   // count pass: every workgroup also clears its share of the send segments (unused entries = row -1)
    const int64_t m = (int64_t)W * cap;
    const int64_t per = (m + gridDim.x - 1) / gridDim.x;
    const int64_t i1 = ((int64_t)blockIdx.x + 1) * per < m ? ((int64_t)blockIdx.x + 1) * per : m;
    for (int64_t i = (int64_t)blockIdx.x * per + t; i < i1; i += 256) send_rows[i] = -1;
  }
  if (t < MAX_W) running[t] = (SCATTER && t < W) ? cntm[(int64_t)blockIdx.x * W + t] : 0;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * per_chunk;
  const int64_t i1 = i0 + per_chunk < n ? i0 + per_chunk : n;
  for (int64_t base = i0; base < i1; base += 256) {
    const int64_t i = base + t;
    bool head = false;
    uint64_t pr = 0;
    if (i < i1) {
      pr = pairs[i];
      head = i == 0 || uq_key(pairs[i - 1]) != uq_key(pr);
    }
    const uint32_t key = uq_key(pr);
    const int32_t o = head ? (int32_t)(key % (uint32_t)W) : -1;
    int32_t my_rank = 0;
    for (int w = 0; w < W; ++w) {
      const unsigned long long m = __ballot(o == w);
      if (o == w) my_rank = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wcount[wave][w] = __popcll(m);
    }
    __syncthreads();
    if (SCATTER && head) {
      int32_t before = running[o];
      for (int w2 = 0; w2 < wave; ++w2) before += wcount[w2][o];
      const int32_t p = before + my_rank;   // index inside the owner's segment
      const int32_t occ = (int32_t)(uint32_t)pr;
      if (p < cap) {
        send_rows[(int64_t)o * cap + p] = (int32_t)(key / (uint32_t)W);
        pos[occ] = o * cap + p;
      } else {
        pos[occ] = -1;
      }
    }
    __syncthreads();
    if (t < W) running[t] += wcount[0][t] + wcount[1][t] + wcount[2][t] + wcount[3][t];
    __syncthreads();
  }
  if (!SCATTER && t < W) cntm[(int64_t)blockIdx.x * W + t] = running[t];
}

// every occurrence that is not the first of its row: the entry of the row's head (first position holding the key: lower bound)
__global__ void __launch_bounds__(256)
k_route_unique_fill(const uint64_t *__restrict__ pairs, int64_t n, int32_t *__restrict__ pos) {
  WD_SIDE_PRIO();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || i == 0) return;
  const uint64_t pr = pairs[i];
  const uint32_t key = uq_key(pr);
  if (uq_key(pairs[i - 1]) != key) return;      // a head: done by the scatter pass
  int64_t lo = 0, hi = i - 1;                    // first index in [0, i - 1] holding `key` (pairs[i - 1] does)
  // galloping backwards first: most rows have a handful of occurrences
  int64_t step = 1;
  while (i - 1 - step >= 0 && uq_key(pairs[i - 1 - step]) == key) step <<= 1;
  lo = i - 1 - step >= 0 ? i - 1 - step + 1 : 0;
  hi = i - 1 - (step >> 1);
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (uq_key(pairs[mid]) == key) hi = mid; else lo = mid + 1;
  }
  pos[(int32_t)(uint32_t)pr] = pos[(int32_t)(uint32_t)pairs[lo]];
}

}  // namespace

extern "C" int32_t wd_route_chunks(void) { return MAX_CHUNKS; }

extern "C" int wd_route_build(const wd_slot_t *local_slots, int32_t S, int32_t world, const int32_t *ids,
                              const int32_t *bag_offs, int64_t batch, int32_t cap, int32_t *send_rows, int32_t *pos,
                              int32_t *workspace, int32_t *peer_counts, int32_t *overflow, wd_stream_t stream) {
  WD_REQUIRE(local_slots && ids && bag_offs && send_rows && pos && workspace && peer_counts && overflow, "null pointer");
  WD_REQUIRE(world >= 1 && world <= MAX_W && S > 0 && cap > 0, "bad geometry");
  hipStream_t st = wd::as_stream(stream);
  const int64_t nbags = batch * S;
  const int64_t bags_per_chunk = nbags > 0 ? wd::ceil_div(wd::ceil_div(nbags, MAX_CHUNKS), 256) * 256 : 256;
  const int nchunks = nbags > 0 ? (int)wd::ceil_div(nbags, bags_per_chunk) : 0;
  if (nchunks > 0)
    hipLaunchKernelGGL((k_route<false>), dim3(nchunks), dim3(256), 0, st, local_slots, S, world, ids, bag_offs, nbags,
                       bags_per_chunk, cap, workspace, send_rows, pos);
  hipLaunchKernelGGL(k_route_scan, dim3(1), dim3(1024), 0, st, workspace, nchunks, world, cap, send_rows, overflow,
                     peer_counts);
  if (nchunks > 0)
    hipLaunchKernelGGL((k_route<true>), dim3(nchunks), dim3(256), 0, st, local_slots, S, world, ids, bag_offs, nbags,
                       bags_per_chunk, cap, workspace, send_rows, pos);
  return wd::check_launch("wd_route_build");
}


extern "C" int wd_route_unique(const uint64_t *sorted_pairs, int64_t n, int32_t world, int32_t cap, int32_t *send_rows,
                               int32_t *pos, int32_t *workspace, int32_t *peer_counts, int32_t *overflow, wd_stream_t stream) {
  WD_REQUIRE(sorted_pairs && send_rows && pos && workspace && peer_counts && overflow, "null pointer");
  WD_REQUIRE(world >= 1 && world <= MAX_W && cap > 0 && n >= 0 && n < ((int64_t)1 << 31), "bad geometry");
  hipStream_t st = wd::as_stream(stream);
  const int64_t per_chunk = n > 0 ? wd::ceil_div(wd::ceil_div(n, MAX_CHUNKS), 256) * 256 : 256;
  const int nchunks = n > 0 ? (int)wd::ceil_div(n, per_chunk) : 0;
  if (nchunks > 0)
    hipLaunchKernelGGL((k_route_unique<false>), dim3(nchunks), dim3(256), 0, st, sorted_pairs, n, per_chunk, world, cap,
                       workspace, send_rows, pos);
  hipLaunchKernelGGL(k_route_scan, dim3(1), dim3(1024), 0, st, workspace, nchunks, world, cap, send_rows, overflow,
                     peer_counts);
  if (nchunks > 0) {
    hipLaunchKernelGGL((k_route_unique<true>), dim3(nchunks), dim3(256), 0, st, sorted_pairs, n, per_chunk, world, cap,
                       workspace, send_rows, pos);
    hipLaunchKernelGGL(k_route_unique_fill, dim3((unsigned)wd::ceil_div(n, (int64_t)256)), dim3(256), 0, st, sorted_pairs, n, pos);
  }
  return wd::check_launch("wd_route_unique");
}

extern "C" int wd_owner_gather(const float *emb, int64_t n_emb_rows, int32_t dim, const float *wide,
                               const int32_t *rows, int64_t n, float *out, int32_t row_stride, wd_stream_t stream) {
  if (n <= 0) return WD_OK;
  WD_REQUIRE(rows && out && (emb || wide), "null pointer");
  WD_REQUIRE(!emb || (dim > 0 && dim % 4 == 0 && row_stride % 4 == 0), "dim and row_stride must be multiples of 4");
  hipLaunchKernelGGL(k_owner_gather, dim3((unsigned)wd::ceil_div(n * 4, 256)), dim3(256), 0, wd::as_stream(stream), emb,
                     n_emb_rows, dim, wide, rows, n, out, row_stride);
  return wd::check_launch("wd_owner_gather");
}

extern "C" int wd_owner_gather_rec(const float *rec, int32_t rec_stride, int32_t dim, const int32_t *rows, int64_t n,
                                   float *out, int32_t row_stride, wd_stream_t stream) {
  if (n <= 0) return WD_OK;
  WD_REQUIRE(rec && rows && out, "null pointer");
  WD_REQUIRE(dim > 0 && dim % 4 == 0 && rec_stride % 4 == 0 && rec_stride >= dim + 4 && row_stride >= dim + 4 && row_stride % 4 == 0,
             "record = [dim | w z n -], exchanged row = [dim | w ...], 16-byte aligned");
  const int nvec = dim / 4 + 1;
  hipLaunchKernelGGL(k_owner_gather_rec, dim3((unsigned)wd::ceil_div(n * nvec, 256)), dim3(256), 0, wd::as_stream(stream),
                     rec, rec_stride, nvec, rows, n, out, row_stride);
  return wd::check_launch("wd_owner_gather_rec");
}

extern "C" int wd_grad_pack(const wd_slot_t *slots, int32_t S, const int32_t *bag_offs, const int32_t *pos,
                            int64_t batch, const float *dx, int64_t ldx, const float *dlogit, int32_t dim,
                            int32_t row_stride, float *out, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(slots && bag_offs && pos && out && (dx || dlogit), "null pointer");
  WD_REQUIRE(!dx || (dim > 0 && dim % 4 == 0 && row_stride % 4 == 0), "dim and row_stride must be multiples of 4");
  const int64_t nbags = batch * S;
  hipLaunchKernelGGL(k_grad_pack, dim3((unsigned)wd::ceil_div(nbags * 4, 256)), dim3(256), 0, wd::as_stream(stream),
                     slots, S, bag_offs, pos, nbags, dx, ldx, dlogit, dim, row_stride, out);
  return wd::check_launch("wd_grad_pack");
}

extern "C" int wd_fill_i32(int32_t *p, int32_t v, int64_t n, wd_stream_t stream) {
  if (n <= 0) return WD_OK;
  int blocks = (int)std::min<int64_t>(wd::ceil_div(n, 256), 2048);
  hipLaunchKernelGGL(k_fill_i32, dim3(blocks), dim3(256), 0, wd::as_stream(stream), p, v, n);
  return wd::check_launch("wd_fill_i32");
}
