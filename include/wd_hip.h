/*
 * include/wd_hip.h -- C ABI of the MI355X (gfx950) Wide&Deep train-step hot path.
 *
 * The reference (Lapis-Hong/wide_deep) has no FFI boundary of its own: the hot
 * path is Python that selects TensorFlow ops.  Each entry point below replaces
 * the TF op(s) that one reference call site selects; the citation is the
 * reference file:line that wires that op in.
 *
 * Conventions
 *   - plain C ABI, no torch / STL types; every pointer is a DEVICE pointer
 *     unless its name ends in _host.
 *   - the caller owns every buffer; nothing is allocated behind the caller's
 *     back (workspace sizes come from the *_workspace_bytes queries).
 *   - every launch goes to the hipStream_t passed as `stream` (void* here so
 *     that the header needs no HIP include); calls are asynchronous.
 *   - return value: WD_OK (0) or a negative WD_ERR_*; wd_last_error() gives a
 *     thread-local message for the last failure.
 *   - ids and CSR offsets are int32 on the device (SURVEY 8(d) byte contract).
 *
 * Device batch layout ("bag CSR", example-major): with S categorical slots
 * and a batch of B examples, bag(b, s) = b*S + s; bag_offs[B*S + 1] are the
 * CSR offsets into ids[nnz]; ids are slot-local row numbers in [0, num_buckets).
 */
#ifndef WD_HIP_H_
#define WD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WD_OK 0
#define WD_ERR_INVALID (-1)
#define WD_ERR_LAUNCH (-2)
#define WD_ERR_WORKSPACE (-3)
#define WD_ERR_UNSUPPORTED (-4)

#define WD_MAX_CROSS_KEYS 8

typedef void *wd_stream_t; /* hipStream_t */

const char *wd_last_error(void);
/* sha256 (hex) over the sources the library was built from (csrc/build.sh); the Python loader compares it with the sources it
 * finds next to itself and refuses a stale binary. */
const char *wd_build_stamp(void);
int wd_abi_version(void);

/* Per-slot descriptor (device array of S entries, built once by the host). */
typedef struct wd_slot {
  int64_t emb_off;     /* element offset of row 0 of this slot's table in the flat embedding buffer; -1: none */
  int64_t row_base;    /* first row of this slot in the fused categorical row space (wide table / sort keys) */
  int32_t num_buckets; /* rows of this slot */
  int32_t dim;         /* embedding dim D (0 if the slot has no embedding) */
  int32_t out_col;     /* first column of this slot in the deep input matrix x (-1: not in deep input) */
  int32_t kind;        /* WD_SLOT_* */
  int32_t wide;        /* 1: slot contributes to the wide (linear) logit */
  /* bucket geometry of the fused sparse backward: bucket(id) = bucket_base + (id >> bucket_shift).  Slots with few
   * rows get bucket_shift 0 (ONE row per bucket: its occurrences need no sort), big tables ~64 occurrences a bucket. */
  int32_t bucket_shift;
  int32_t bucket_base;
  int32_t flags;       /* WD_SLOT_F_*; 0 in the tables every entry point but the ones named there takes */
} wd_slot_t;

/* wd_slot_t.flags bit 0: the column is handled by wd_small_tables_fwd / _bwd (csrc/small_tables.hip); wd_wide_fwd and
 * wd_sparse_bucketize treat its bags as empty when they are handed a slot table that carries the flag. */
#define WD_SLOT_F_SMALL 1

#define WD_SLOT_NONE 0      /* wide-only categorical column (bucketized, cross with is_deep=0) */
#define WD_SLOT_EMBEDDING 1 /* embedding_column(combiner='mean')   build_estimator.py:90-97,157 */
#define WD_SLOT_INDICATOR 2 /* indicator_column (multi-hot counts) build_estimator.py:108,118 */

/* ---- a4: categorical_column_with_hash_bucket -> string_to_hash_bucket_fast ------------------
 * (python/lib/build_estimator.py:86-88).  Tokens are packed in `bytes`, token t is
 * bytes[tok_offs[t] .. tok_offs[t+1]).  out_fp[t] = FarmHash Fingerprint64(token t). */
int wd_fingerprint64(const uint8_t *bytes, const int32_t *tok_offs, int64_t ntok, uint64_t *out_fp,
                     wd_stream_t stream);
/* The same with the token count on the DEVICE: tokens [0, *ntok_dev] are hashed (the last one is the batch's trailing ''), the
 * grid covers ntok_capacity -- for a featurizer captured into a hipGraph over fixed-capacity buffers. */
int wd_fingerprint64_dyn(const uint8_t *bytes, const int32_t *tok_offs, int64_t ntok_capacity, const int32_t *ntok_dev,
                         uint64_t *out_fp, wd_stream_t stream);

/* ids[t] = Fingerprint64(token t) % num_buckets of the token's slot.  Tokens are in bag order
 * (example-major); token_bag_offs = bag CSR over tokens (NULL: exactly one token per bag, bag t = token t);
 * the slot of bag g is g % S.  Fuses fingerprint + modulo for the pure hash-bucket configs. */
int wd_hash_bucket(const uint8_t *bytes, const int32_t *tok_offs, int64_t ntok, const int32_t *token_bag_offs,
                   int64_t nbags, const wd_slot_t *slots, int32_t S, int32_t *out_ids, wd_stream_t stream);
/* One token per bag (nbags = batch * S tokens): as above, plus a SLOT-MAJOR copy out_ids_cols[s * batch + b] = ids[b * S + s]
 * (the id column of every slot contiguous: what wd_bucket_onehot reads). */
int wd_hash_bucket_cols(const uint8_t *bytes, const int32_t *tok_offs, int64_t nbags, const wd_slot_t *slots, int32_t S,
                        int32_t *out_ids, int32_t *out_ids_cols, wd_stream_t stream);

/* Emit the ids of ONE slot from precomputed fingerprints of one feature's tokens:
 * feature CSR feat_offs[B+1] into fp[]; writes ids[bag_offs[b*S+slot] + j] = fp % num_buckets. */
int wd_emit_hash_slot(const uint64_t *fp, const int32_t *feat_offs, int64_t batch, uint64_t num_buckets,
                      const int32_t *bag_offs, int32_t S, int32_t slot, int32_t *ids, wd_stream_t stream);

/* Emit already-integer ids (identity / bucketized / host-looked-up vocab) of ONE slot. */
int wd_emit_int_slot(const int32_t *vals, const int32_t *feat_offs, int64_t batch, const int32_t *bag_offs, int32_t S,
                     int32_t slot, int32_t *ids, wd_stream_t stream);

/* ---- a5: crossed_column -> SparseCross (hashed)  (python/lib/build_estimator.py:138-155) ----
 * Key k of the cross: vals[k] (uint64: Fingerprint64 of a string token, or the int64 id of an
 * identity / bucketized key) with CSR offs[k][batch+1].  Per example the cartesian product is
 * enumerated with the LAST key varying fastest; h = hash_key; h = FingerprintCat64(h, v_k);
 * id = h % num_buckets (0 buckets: % INT64_MAX).  Output goes to slot `slot` of the bag CSR;
 * bag_offs must already give that bag prod_k(count_k) entries. */
typedef struct wd_cross_keys {
  const uint64_t *vals[WD_MAX_CROSS_KEYS];
  const int32_t *offs[WD_MAX_CROSS_KEYS];
  int32_t nkeys;
} wd_cross_keys_t;

int wd_cross_hash(const wd_cross_keys_t *keys_host, int64_t batch, uint64_t hash_key, uint64_t num_buckets,
                  const int32_t *bag_offs, int32_t S, int32_t slot, int32_t *ids, wd_stream_t stream);

/* ---- a4-a6 for a WHOLE batch on the device (python/lib/build_estimator.py:83-158: every categorical column of the conf) ----
 * Instead of one emit launch per column with host-prepared value arrays (wd_emit_*_slot / wd_cross_hash), the host uploads
 * the parsed batch as it is -- token bytes + offsets of all string features, integer / float features as
 * [feature][example] matrices -- and four launches build the example-major bag CSR of ids:
 *   wd_fingerprint64 (all tokens) -> wd_feat_vocab_lookup (per vocabulary_list column) -> wd_feat_lens -> wd_feat_offsets
 *   -> wd_feat_emit.
 * Column semantics (same as the per-column entry points; reference lines in the a4-a6 sections above):
 *   HASH      id = Fingerprint64(token) % num_buckets, one per token
 *   VOCAB     index in vocabulary_list, out-of-vocabulary tokens dropped
 *   IDENTITY  v if 0 <= v < num_buckets else 0; v == -1 (missing) gives no id
 *   BUCKET    number of boundaries <= normalizer(x): the wide twin of a continuous column (quirk C.5: boundaries are raw)
 *   CROSS     cartesian product over the keys, LAST key fastest, h = hash_key; h = FingerprintCat64(h, v_k); id = h % num_buckets;
 *             a STRING key contributes its tokens' fingerprints -- with `lmax` (the reference's padded_batch, quirk C.16) every
 *             example contributes lmax[feature] values, the missing ones being Fingerprint64('') --, an IDENTITY key its id
 *             (none if -1), a BUCKET key bucketize(raw x). */
#define WD_FEAT_HASH 0
#define WD_FEAT_VOCAB 1
#define WD_FEAT_IDENTITY 2
#define WD_FEAT_BUCKET 3
#define WD_FEAT_CROSS 4
#define WD_FEAT_KEY_STRING 0
#define WD_FEAT_KEY_IDENTITY 1
#define WD_FEAT_KEY_BUCKET 2
typedef struct wd_feat_key {
  int32_t kind;          /* WD_FEAT_KEY_* */
  int32_t src;           /* string feature index / row of `ints` / row of `floats` */
  int32_t num_buckets;   /* IDENTITY: range of valid ids */
  int32_t nbound;        /* BUCKET: boundaries[bound_off .. bound_off + nbound) of wd_feat_batch_t.bounds */
  int32_t bound_off;
  int32_t pad_;
} wd_feat_key_t;
typedef struct wd_feat_slot {
  int32_t kind;          /* WD_FEAT_* */
  int32_t src;           /* HASH / VOCAB: string feature index; IDENTITY: row of `ints`; BUCKET: row of `floats` */
  int32_t num_buckets;
  int32_t nbound, bound_off;
  int32_t norm_kind;     /* BUCKET: 0 none, 1 (x - p0) / (p1 - p0), 2 (x - p0) / p1 (log: applied by the host) */
  float p0, p1;
  int32_t nkeys;         /* CROSS */
  int32_t pad_;
  uint64_t hash_key;
  wd_feat_key_t keys[WD_MAX_CROSS_KEYS];
} wd_feat_slot_t;
typedef struct wd_feat_batch {
  const uint64_t *fp;        /* [T + 1] Fingerprint64 of every token; entry T = the trailing '' token (cross padding) */
  const int32_t *tok_val;    /* [T] vocabulary index per token (wd_feat_vocab_lookup), -1 = not in the list; NULL: no VOCAB column */
  const int32_t *ex_offs;    /* [F][batch + 1] token ranges per string feature and example, relative to tok_base[f] */
  const int32_t *tok_base;   /* [F] first token of feature f in fp / tok_val */
  const int32_t *lmax;       /* [F] longest token list of feature f in THIS batch (tf_dense cross padding); NULL: ragged */
  const int64_t *ints;       /* [NI][batch] */
  const float *floats;       /* [NF][batch] */
  const float *bounds;       /* boundaries of all bucketized columns / keys */
  int64_t batch;
  int32_t S;                 /* slots per example (= rows of the slot table) */
  int32_t empty_index;       /* T */
  const int32_t *ntok_dev;   /* optional: T on the DEVICE (overrides empty_index) -- a featurizer captured into a hipGraph whose
                              * batches differ in token count (fixed-capacity buffers, wd_fingerprint64_dyn) */
} wd_feat_batch_t;
/* vocabulary_list of one string feature for wd_feat_vocab_lookup_all (device pointers; nvocab 0: the feature has none) */
typedef struct wd_feat_vocab {
  const uint8_t *bytes;
  const int32_t *offs;       /* [nvocab + 1] */
  int32_t nvocab;
  int32_t pad_;
} wd_feat_vocab_t;
int wd_feat_vocab_lookup(const uint8_t *bytes, const int32_t *tok_offs, int64_t tok_begin, int64_t n,
                         const uint8_t *vocab_bytes, const int32_t *vocab_offs, int32_t nvocab, int32_t *tok_val,
                         wd_stream_t stream);
/* The same for EVERY vocabulary_list column in one launch: token t belongs to the feature f with tok_base[f] <= t < tok_base[f] +
 * tok_n[f] (device arrays [nfeat]) and is looked up in vocab_table_dev[f].  ntok: tokens (grid size; with ntok_dev non-NULL the
 * capacity, the count is read from the device). */
int wd_feat_vocab_lookup_all(const uint8_t *bytes, const int32_t *tok_offs, int64_t ntok, const int32_t *ntok_dev,
                             const int32_t *tok_base, const int32_t *tok_n, int32_t nfeat, const wd_feat_vocab_t *vocab_table_dev,
                             int32_t *tok_val, wd_stream_t stream);
/* lens[b * S + s] = ids column s gives example b; block_stats[2 k], [2 k + 1] = sum of the lengths of pairs [256 k, 256 k + 256)
 * and "one of them is not 1" (wd_feat_offsets_workspace_bytes(batch * S) bytes). */
int wd_feat_lens(const wd_feat_slot_t *slots_dev, const wd_feat_batch_t *batch, int32_t *lens, int32_t *block_stats,
                 wd_stream_t stream);
int64_t wd_feat_offsets_workspace_bytes(int64_t n);
/* Bag CSR of the n = batch * S bags: offs[0] = 0, offs[i + 1] = lens[0] + ... + lens[i] (offs[n] = ids of the batch -- it stays on
 * the device).  flags (optional, device int32[2], plain stores -- nothing to zero): flags[0] = offs[n] > ids_capacity,
 * flags[1] = some bag does not hold exactly one id (the engine's one-id-per-bag fast paths need to know). */
int wd_feat_offsets(const int32_t *lens, const int32_t *block_stats, int64_t n, int32_t *offs, int64_t ids_capacity, int32_t *flags,
                    wd_stream_t stream);
/* ids of every bag, one lane per OUTPUT id (a crossed column's bag is the product of its keys' counts: 25-125 ids per example at
 * BASELINE configs[3]); the grid is sized by the batch, not by the id count, so the count never has to reach the host:
 * `ids` holds `ids_capacity` entries, ids beyond it are dropped (wd_feat_offsets raised flags[0]).  At most 1024 categorical columns. */
int wd_feat_emit(const wd_feat_slot_t *slots_dev, const wd_feat_batch_t *batch, const int32_t *bag_offs, int32_t *ids,
                 int64_t ids_capacity, wd_stream_t stream);

/* ---- a8: tf.feature_column.input_layer (python/lib/dnn.py:83-90) ---------------------------
 * Fused multi-slot embedding-bag gather: for every slot g in group_slots (all of one dim D):
 *   x[b, out_col_g .. +D) = mean_{ids in bag(b,g)} emb[emb_off_g + id*D ..]   (empty bag -> 0). */
int wd_embag_fwd(const float *emb, const wd_slot_t *slots, int32_t S, const int32_t *group_slots, int32_t ngroup,
                 int32_t dim, const int32_t *ids, const int32_t *bag_offs, int64_t batch, float *x, int64_t ldx,
                 wd_stream_t stream);
/* Round 6, row records and ragged bags: wd_embag_fwd_strided on the record table that also leaves the wide sum of every bag
 * (python/lib/linear.py:29-36) -- wide_vals[b * S + s] = sum over the bag of rec[row][dim], the word behind the row in the line the gather
 * fetched anyway; wd_wide_sum: wide_logit[b] = bias + the sums of b's bags (columns that are not wide or carry WD_SLOT_F_SMALL skipped).
 * Replaces wd_wide_fwd's second pass over the same lines. */
int wd_embag_fwd_wide(const float *rec, int64_t rec_stride, const wd_slot_t *rec_slots, int32_t S, const int32_t *group_slots,
                      int32_t ngroup, int32_t dim, const int32_t *ids, const int32_t *bag_offs, int64_t batch, float *x, int64_t ldx,
                      float *wide_vals, wd_stream_t stream);
int wd_wide_sum(const float *wide_vals, const float *bias, const wd_slot_t *slots, int32_t S, int64_t batch, float *out,
                wd_stream_t stream);
/* Same with an explicit row stride (floats, multiple of 4, >= dim): row i of a slot starts at emb_off + i*row_stride.
 * Used to pool rows that arrived through the all-to-all exchange (rows are then `row_stride` apart); ids < 0 are
 * skipped (they still count in the mean's denominator, like an all-zero row). */
int wd_embag_fwd_strided(const float *emb, int64_t row_stride, const wd_slot_t *slots, int32_t S,
                         const int32_t *group_slots, int32_t ngroup, int32_t dim, const int32_t *ids,
                         const int32_t *bag_offs, int64_t batch, float *x, int64_t ldx, wd_stream_t stream);

/* Same result for a CONTIGUOUS slot range [slot0, slot0 + ngroup) (the engine orders the slots of one dim together):
 * no group_slots indirection, slot metadata staged in LDS, two bags per lane group.  dim in {4,8,16,32,64,128}.
 * bag_offs == NULL declares a strictly one-id-per-bag batch (bag_offs would be 0,1,2,...): ids[bag] is read directly
 * and the CSR offsets are never touched (one dependent memory round trip less). */
int wd_embag_fwd_range(const float *emb, const wd_slot_t *slots, int32_t S, int32_t slot0, int32_t ngroup, int32_t dim,
                       const int32_t *ids, const int32_t *bag_offs, int64_t batch, float *x, int64_t ldx,
                       wd_stream_t stream);

/* indicator_column slots: x[b, out_col + id] = multiplicity of id in bag(b, slot). */
int wd_indicator_fwd(const wd_slot_t *slots, int32_t S, const int32_t *group_slots, int32_t ngroup,
                     const int32_t *ids, const int32_t *bag_offs, int64_t batch, float *x, int64_t ldx,
                     wd_stream_t stream);

/* numeric_column(normalizer_fn) (python/lib/build_estimator.py:61-68,121-136):
 * x[b, out_col[j]] = f_j(dense[b, j]); kind 0: identity, 1: (v-p0)/(p1-p0) min_max, 2: (v-p0)/p1 standard, 3: log. */
typedef struct wd_dense_col {
  float p0, p1;
  int32_t kind;
  int32_t out_col;
} wd_dense_col_t;
int wd_dense_fwd(const float *dense, int64_t ld_dense, const wd_dense_col_t *cols, int32_t ncols, int64_t batch,
                 float *x, int64_t ldx, wd_stream_t stream);

/* The sparse forward of a step in ONE launch: wd_embag_fwd_range (one dim group) + wd_wide_fwd (wide may be NULL;
 * AoS table, stride 4) + wd_dense_fwd (ncols may be 0) share a grid -- they are independent, and every kernel boundary
 * on this part costs a cold L2.  bag_offs is always required (the wide part walks the CSR); one_id_per_bag != 0
 * additionally lets the gather skip it. */
int wd_input_layer_fwd(const float *emb, const wd_slot_t *slots, int32_t S, int32_t slot0, int32_t ngroup, int32_t dim,
                       const int32_t *ids, const int32_t *bag_offs, int32_t one_id_per_bag, int64_t batch, float *x,
                       int64_t ldx, const float *dense, int64_t ld_dense, const wd_dense_col_t *dense_cols, int32_t ncols,
                       const float *wide, const float *bias, float *wide_out, wd_stream_t stream);


/* ---- a7: tf.feature_column.linear_model(sparse_combiner='sum') (python/lib/linear.py:29-36) --
 * wide state is array-of-structs: wide[row*4 + {0,1,2}] = {w, z (Ftrl_1 "linear"), n (Ftrl "accum")}.
 * out[b] = bias[0] + sum over wide slots s, ids in bag(b,s) of wide[(row_base_s + id) * wide_stride]
 * (wide_stride 4 for the AoS table; 1 when summing weights that arrived through the all-to-all exchange). */
int wd_wide_fwd(const float *wide, int32_t wide_stride, const float *bias, const wd_slot_t *slots, int32_t S,
                const int32_t *ids, const int32_t *bag_offs, int64_t batch, float *out, wd_stream_t stream);

/* ---- a11: head, sigmoid CE, SUM reduction, weight column (python/lib/joint.py:216-222,402-406) --
 * logit = dnn_logit (may be NULL) + wide_logit (may be NULL); loss_sum[0] += sum_b w_b*CE; dlogit[b] = w_b*(p-y);
 * prob[b] = sigmoid(logit).  weights may be NULL.  loss_sum must be zeroed by the caller. */
int wd_bce_sum_fwd_bwd(const float *dnn_logit, const float *wide_logit, const float *labels, const float *weights,
                       int64_t batch, float *logit, float *prob, float *dlogit, float *loss_sum, wd_stream_t stream);

/* The same batch-SUM loss (python/lib/joint.py:264-269 -> head loss) from the STORED logits, summed in a
 * fixed order: loss_sum[0] = sum_b w_b * CE(logit_b, y_b) (stored, not accumulated).  The head kernels add their partial
 * sums to loss_sum with a float atomic (arrival order); callers that need a bit-reproducible loss pass loss_sum = NULL
 * there and launch this instead (one workgroup). */
int wd_bce_loss_sum(const float *logit, const float *labels, const float *weights, int64_t batch, float *loss_sum,
                    wd_stream_t stream);

/* ---- a12: sparse optimizer apply (python/lib/joint.py:224-262, utils/model_util.py:84-90) ----
 * Step 1: keys[j] = row_base[slot(j)] + ids[j], vals[j] = bag(j); stable radix sort by key.
 * Workspace query then sort. */
size_t wd_sort_workspace_bytes(int64_t nnz, int32_t key_bits);
int wd_build_sort_keys(const wd_slot_t *slots, int32_t S, const int32_t *ids, const int32_t *bag_offs, int64_t nbags,
                       int64_t nnz, uint32_t *keys, int32_t *vals, wd_stream_t stream);
int wd_sort_pairs(const uint32_t *keys_in, const int32_t *vals_in, uint32_t *keys_out, int32_t *vals_out, int64_t nnz,
                  int32_t key_bits, void *workspace, size_t workspace_bytes, wd_stream_t stream);

/* Step 2 (embedding rows, one launch per dim group): for every unique key whose slot has dim D,
 * g = sum over its occurrences of dx[b, out_col..+D) / len(bag)   (duplicates summed first, TF IndexedSlices),
 * then tf.train.AdagradOptimizer: acc += g*g; row -= lr*g/sqrt(acc). */
int wd_embag_bwd_adagrad(float *emb, float *emb_accum, const wd_slot_t *slots, int32_t S, int32_t dim,
                         const uint32_t *keys_sorted, const int32_t *vals_sorted, int64_t nnz,
                         const int32_t *bag_offs, const float *dx, int64_t ldx, float lr, wd_stream_t stream);

/* Step 3 (wide rows): g = sum of dlogit[b] over the occurrences of the row, then tf.train.FtrlOptimizer
 * (lr_power -0.5): n' = n+g^2; z += g-(sqrt(n')-sqrt(n))/lr*w; w = |z|>l1 ? (sign(z)*l1-z)/(sqrt(n')/lr+2*l2) : 0. */
int wd_wide_bwd_ftrl(float *wide, const wd_slot_t *slots, int32_t S, const uint32_t *keys_sorted,
                     const int32_t *vals_sorted, int64_t nnz, const float *dlogit, float lr, float l1, float l2,
                     wd_stream_t stream);

/* bias_weights of the linear model: g = sum_b dlogit[b]; dense FTRL on {w,z,n} = bias[0..2]. */
int wd_bias_ftrl(float *bias_wzn, const float *dlogit, int64_t batch, float lr, float l1, float l2,
                 wd_stream_t stream);

/* Fused sparse backward (the path the engine uses; same arithmetic as steps 1-3 above in FOUR short launches, no
 * device-wide sort): occurrences are bucketed by row range (bucket = slot.bucket_base + (id >> slot.bucket_shift); the
 * placement inside a bucket is STABLE, i.e. in ascending bag order), each multi-row bucket is sorted on (row, bag) by
 * one workgroup in LDS (single-row buckets need no sort), duplicates are summed in ascending bag order and Adagrad
 * (embedding rows, lr_emb) / FTRL (wide rows and bias_wzn, lr_wide, l1, l2) are applied in the same kernel.
 * emb / wide / bias_wzn may be NULL (deep-only / wide-only).  Workspaces (caller-owned, no initialisation needed):
 * bucket_cnt[(2 * wd_bucket_chunks() + 1) * nbuckets], bucket_start[2 * nbuckets + 2] (bucket starts, then the order in
 * which the update kernel takes the buckets: largest first), rank[nnz], pairs[nnz] (uint64).
 * nbuckets = sum over slots of ceil(num_buckets / 2^bucket_shift) <= wd_bucket_max().  dlogit of example b is dlogit[b * ld_dlogit]; ids < 0
 * are padding and are skipped; an embedding slot updates only keys below row_base + num_buckets. */
int32_t wd_bucket_max(void);
int32_t wd_bucket_chunks(void);
/* The two phases of wd_sparse_bwd_fused as separate entry points: bucketize needs only the ids (the engine runs it on a
 * side stream, concurrently with the tower), apply needs dx / dlogit. */
int wd_sparse_bucketize(const wd_slot_t *slots, int32_t S, const int32_t *ids, const int32_t *bag_offs, int64_t batch,
                        int64_t nnz, int32_t *bucket_cnt, int32_t *bucket_start, int32_t *rank, uint64_t *pairs,
                        int32_t nbuckets, wd_stream_t stream);
int wd_sparse_apply(float *emb, float *emb_accum, float *wide, float *bias_wzn, const wd_slot_t *slots, int32_t S,
                    const int32_t *bag_offs, int64_t batch, const float *dx, int64_t ldx, const float *dlogit,
                    int64_t ld_dlogit, float lr_emb, float lr_wide, float l1, float l2, const int32_t *bucket_start,
                    uint64_t *pairs, int32_t nbuckets, wd_stream_t stream);
/* wd_sparse_apply on the ROW-RECORD layout: one table of `rec_stride`-float records indexed by the FUSED row
 * (slot.row_base + id), record = [embedding row (dim floats) | w, z, n, - | pad]; the Adagrad accumulator keeps the flat
 * per-slot layout of wd_sparse_apply (slot.emb_off).  An update then touches two random lines per row instead of three, and
 * the forward finds the wide weight in the line of the embedding row (wd_chain_input_t.wide_in_row, wd_wide_fwd with
 * wide = rec + dim and stride rec_stride, wd_embag_fwd_strided with slot.emb_off = row_base * rec_stride). */
/* `next` (may be NULL), for the pipelined step of one-id-per-bag batches (wide_deep_amd/pipeline.py):
 *   unsorted_buckets != 0: `pairs` came from wd_bucket_onehot (arrival order inside a bucket): one-row buckets are sorted too;
 *   bucket_start != NULL: the input layer of the NEXT batch has already been gathered (wd_prefetch_onehot into x / wide_vals)
 *   from the tables as they were BEFORE this update; `bucket_start` / `pairs` are the next batch's buckets (same geometry).
 *   Every row rewritten here that the next batch holds too is stored again -- x[b'*ldx + out_col + 0..dim) = new row,
 *   wide_vals[bag'] = new w -- so that the next forward reads exactly what a gather AFTER this update would have read
 *   (python/lib/joint.py:233-262: the reference applies both optimizers before the next session.run reads a variable). */
typedef struct wd_apply_next {
  const int32_t *bucket_start;
  const uint64_t *pairs;
  float *x;
  float *wide_vals;
  int64_t ldx;
  int32_t unsorted_buckets, pad_;
} wd_apply_next_t;
int wd_sparse_apply_rec(float *rec, int32_t rec_stride, int32_t dim, float *emb_accum, float *bias_wzn,
                        const wd_slot_t *slots, int32_t S, const int32_t *bag_offs, int64_t batch, const float *dx,
                        int64_t ldx, const float *dlogit, int64_t ld_dlogit, float lr_emb, float lr_wide, float l1, float l2,
                        const int32_t *bucket_start, uint64_t *pairs, int32_t nbuckets, const wd_apply_next_t *next,
                        wd_stream_t stream);
/* One-id-per-bag batches (bag (b, s) holds exactly ids[b*S + s] >= 0): the bucketing of wd_sparse_bucketize in ONE launch
 * (csrc/onehot_path.hip).  Every slot owns `batch` occurrences, so slot s fills pairs[s*batch, (s+1)*batch); inside a bucket
 * the pairs are in ARRIVAL order (wd_sparse_apply_rec: wd_apply_next_t.unsorted_buckets).  bucket_start as above
 * ([2 * nbuckets + 2]); max_slot_buckets = the largest per-slot bucket count (LDS of the launch); ticket: one int32 that is
 * 0 before the first call (the last workgroup of a launch lists the buckets largest-first and resets it).
 * ids_slot_major != 0: `ids` is the slot-major copy of wd_hash_bucket_cols (ids[s * batch + b]; coalesced column reads).
 * ticket NULL: no launch order is written.  zero_word (may be NULL): two int32 this launch sets to 0 (wd_bucket_sort's counters). */
int wd_bucket_onehot(const wd_slot_t *slots, int32_t S, const int32_t *ids, int32_t ids_slot_major, int64_t batch,
                     int32_t *bucket_start, uint64_t *pairs, int32_t nbuckets, int32_t max_slot_buckets, int32_t *ticket,
                     int32_t *zero_word, wd_stream_t stream);
/* The id-only part of the sparse update, off the critical path of the step (csrc/onehot_path.hip; python/lib/joint.py:233-248:
 * the duplicate-summing of the IndexedSlices gradient these lists prepare).
 * wd_bucket_sort: every bucket of wd_bucket_onehot sorted in place on (row, bag), two launches (buckets of up to 256 pairs; the
 *   larger ones, listed in big_list[nbuckets], with the LDS a 1024-pair sort needs); long_list[0] and [1] (zeroed by the caller,
 *   e.g. wd_bucket_onehot's zero_word) count the rows with more than 32 occurrences and the large buckets, long_list[2 + 2k],
 *   [3 + 2k] = position and length of such a row (long_capacity entries: batch * S / 32 + 1 always suffice); prev_* (all three or none): for every sorted position i of the PREVIOUS
 *   batch (same bucket geometry) prev_patch[2i], [2i + 1] = position of the first pair of the same row in THIS batch (-1: none)
 *   and the number of its pairs (prev_patch: 2 * batch * S int32, 8-byte aligned).
 * wd_row_update: Adagrad (embedding rows, accumulators flat per slot as in wd_sparse_apply_rec) + FTRL ({w, z, n} behind the
 *   row, and bias_wzn) over the sorted pairs of a one-id-per-bag batch on the row-record table: duplicates of a row summed in
 *   ascending bag order, the optimizer applied once per row -- the arithmetic of wd_sparse_apply_rec, bit for bit -- as a
 *   flat launch (4 lanes per sorted position) with two dependent memory round trips.  patch (may be NULL) = the array the NEXT
 *   batch's wd_bucket_sort filled for this one: rows the next batch reads too are stored into its prefetched input again
 *   (next->pairs = its SORTED pairs, next->x, next->wide_vals, next->ldx; next->bucket_start unused). */
int wd_bucket_sort(const int32_t *bucket_start, uint64_t *pairs, int32_t nbuckets, int32_t *long_list, int32_t long_capacity,
                   int32_t *big_list, int64_t batch, int32_t S, const int32_t *prev_bucket_start,
                   const uint64_t *prev_pairs, int32_t *prev_patch, wd_stream_t stream);
int wd_row_update(float *rec, int32_t rec_stride, int32_t dim, float *emb_accum, float *bias_wzn, const wd_slot_t *slots,
                  int32_t S, int64_t batch, const float *dx, int64_t ldx, const float *dlogit, float lr_emb, float lr_wide,
                  float l1, float l2, const uint64_t *pairs, const int32_t *long_list, int32_t long_capacity,
                  const int32_t *patch, const wd_apply_next_t *next, wd_stream_t stream);
/* The same two steps for RAGGED bags on row records (round 6; multi-hot batches: BASELINE configs[3]): the buckets that
 * wd_sparse_bucketize scattered (bucket_start[nbuckets] = the number of pairs: ids < 0 and the WD_SLOT_F_SMALL columns are not in
 * the list) are sorted in place -- equal (row, bag) pairs, an id repeated inside a bag, keep their order -- and the long rows
 * listed (the two counters long_list[0], [1] are zeroed here); the update walks them as a flat launch over nnz_capacity positions
 * of which *nnz_dev (device; pass bucket_start + nbuckets) exist, every occurrence carrying dx / len(bag) (combiner='mean',
 * python/lib/dnn.py:83-90) -- wd_sparse_apply_rec's arithmetic, summation order included, without its sort inside the update
 * (configs[3]: 259 -> ~150 us in the step, the sort beside the tower).  ld_dlogit: dlogit of example b at dlogit[b * ld_dlogit]
 * (1; the sharded owner's received gradient records [dim | dlogit | ..]: their stride, with bias_wzn NULL). */
int wd_bucket_sort_ragged(const int32_t *bucket_start, uint64_t *pairs, int32_t nbuckets, int32_t *long_list,
                          int32_t long_capacity, int32_t *big_list, int64_t batch, int32_t S, int64_t nnz_capacity,
                          wd_stream_t stream);
int wd_row_update_ragged(float *rec, int32_t rec_stride, int32_t dim, float *emb_accum, float *bias_wzn, const wd_slot_t *slots,
                         int32_t S, int64_t batch, const int32_t *bag_offs, const float *dx, int64_t ldx, const float *dlogit,
                         int64_t ld_dlogit, float lr_emb, float lr_wide, float l1, float l2, const uint64_t *pairs,
                         int64_t nnz_capacity, const int32_t *nnz_dev, const int32_t *long_list, int32_t long_capacity,
                         wd_stream_t stream);
/* The input layer of a one-id-per-bag batch on the row-record tables, as its own launch (python/lib/dnn.py:83-91,
 * python/lib/linear.py:29-36): x[b*ldx + out_col_s + 0..dim) = rec[row(b,s)][0..dim), wide_vals[b*S + s] = rec[row(b,s)][dim]
 * (the wide weight, NOT summed: wd_tower_chain adds them up with the bias, wd_chain_opts_t.wide_vals), numeric columns as
 * wd_dense_fwd.  rec_slots: slot descriptors with emb_off = row_base * rec_stride.  dim in {4, 8, 16}.
 * span (diagnostics, may be NULL): device uint64[wd_prefetch_onehot_blocks(..)][2], the chip-wide 100 MHz realtime clock at the
 * start and at the end of every workgroup of the launch. */
int64_t wd_prefetch_onehot_blocks(int64_t batch, int32_t S, int32_t dim, int32_t ncols);
int wd_prefetch_onehot(const float *rec, int32_t rec_stride, int32_t dim, const wd_slot_t *rec_slots, int32_t S,
                       const int32_t *ids, int64_t batch, float *x, int64_t ldx, float *wide_vals, const float *dense,
                       int64_t ld_dense, const wd_dense_col_t *dense_cols, int32_t ncols, void *span, wd_stream_t stream);
int wd_sparse_bwd_fused(float *emb, float *emb_accum, float *wide, float *bias_wzn, const wd_slot_t *slots, int32_t S,
                        const int32_t *ids, const int32_t *bag_offs, int64_t batch, int64_t nnz, const float *dx,
                        int64_t ldx, const float *dlogit, int64_t ld_dlogit, float lr_emb, float lr_wide, float l1, float l2,
                        int32_t *bucket_cnt, int32_t *bucket_start, int32_t *rank, uint64_t *pairs, int32_t nbuckets,
                        wd_stream_t stream);

/* ---- any optimizer the reference accepts (python/lib/utils/model_util.py:84-90: Adagrad, Adam, Ftrl, RMSProp, SGD =
 * tf.train.{Adagrad,Adam,Ftrl,RMSProp,GradientDescent}Optimizer, TF 1.x kernels), per scope as python/lib/joint.py:224-262
 * applies them: `dnn_optimizer` on embeddings + tower variables, `linear_optimizer` on wide weights + bias.
 * Slots a / b of a variable (same shape; wide lines are {w, a, b, -}):
 *   SGD      -                              var -= lr g
 *   Adagrad  b = accumulator                b += g^2; var -= lr g / sqrt(b)
 *   Ftrl     a = linear, b = accumulator    (as wd_wide_bwd_ftrl; p0 = l1, p1 = l2; p2 = learning_rate_power <= 0: the TF default
 *            -0.5 takes sqrt(accumulator), anything else accumulator^(-p2), FtrlCompute's general branch -- set it
 *            explicitly, a zeroed struct means a FIXED learning rate; p3 = l2_shrinkage_regularization_strength: the linear
 *            slot moves by g + 2 p3 var, the accumulator by the plain g^2 -- TF's FtrlCompute with shrinkage)
 *   RMSProp  a = rms (init 1), b = momentum a += (g^2 - a)(1 - p0); b = p1 b + lr g / sqrt(a + p2); var -= b
 *            (p0 = decay, p1 = momentum, p2 = epsilon; centered = False)
 *   RMSProp centered (WD_OPT_RMSPROP_CENTERED, TF ApplyCenteredRMSProp / SparseApplyCenteredRMSProp): third slot
 *            c = mean gradient (init 0): c += (g - c)(1 - p0); b = p1 b + lr g / sqrt(a - c^2 + p2); var -= b.
 *            c of an embedding table / a dense buffer = wd_opt_t.slot_c (same shape as the variable); of a wide line its
 *            4th float; of the bias bias[3].
 *   Adam     a = m, b = v                   p0 = beta1, p1 = beta2, p2 = epsilon, lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t);
 *            `pow` = device {beta1^t, beta2^t} of the step being applied (wd_adam_tick multiplies it afterwards).
 *            Dense variables: ApplyAdam.  Sparsely updated variables: AdamOptimizer._apply_sparse_shared -- EVERY row
 *            moves (m *= beta1, v *= beta2, var -= lr_t m / (sqrt(v) + eps)); rows with a gradient are done by
 *            wd_sparse_apply_opt, which marks them in `touched` (bit per fused row), the rest by wd_adam_untouched,
 *            which also clears the bitmap. */
#define WD_OPT_SGD 0
#define WD_OPT_ADAGRAD 1
#define WD_OPT_FTRL 2
#define WD_OPT_RMSPROP 3
#define WD_OPT_ADAM 4
#define WD_OPT_ADAM_DENSE 5   /* internal */
#define WD_OPT_RMSPROP_CENTERED 6
typedef struct wd_opt {
  int32_t kind;
  float lr;
  float p0, p1, p2;
  float p3;                 /* Ftrl: l2_shrinkage_regularization_strength (0: none) */
  const float *pow;
  float *slot_c;            /* centered RMSProp: mean-gradient slot of the variable this call updates; else NULL */
} wd_opt_t;
int wd_sparse_apply_opt(float *emb, float *emb_a, float *emb_b, float *wide, float *bias, const wd_slot_t *slots,
                        int32_t S, const int32_t *bag_offs, int64_t batch, const float *dx, int64_t ldx,
                        const float *dlogit, int64_t ld_dlogit, const wd_opt_t *emb_opt, const wd_opt_t *wide_opt,
                        const int32_t *bucket_start, uint64_t *pairs, int32_t nbuckets, uint32_t *touched,
                        wd_stream_t stream);
int wd_opt_dense(float *w, float *slot_a, float *slot_b, const float *g, int64_t n, const wd_opt_t *opt,
                 wd_stream_t stream);
/* max_rows = largest slot (grid sizing), total_rows = size of the fused row space (bitmap length in bits) */
int wd_adam_untouched(float *emb, float *emb_m, float *emb_v, float *wide, const wd_slot_t *slots, int32_t S,
                      int64_t max_rows, int64_t total_rows, uint32_t *touched, const wd_opt_t *emb_opt,
                      const wd_opt_t *wide_opt, wd_stream_t stream);
int wd_adam_tick(float *pow, float beta1, float beta2, wd_stream_t stream);

/* ---- multi-GPU exchange (replaces the PS-partitioned variables of python/lib/joint.py:140-143, train.py:202-225):
 * rows are sharded owner = id % world, local row = row_base_local[slot] + id / world.  All exchange buffers have
 * `world` equal segments of `cap` entries so that the all-to-all split sizes are static (no host sync).
 * wd_route_build: per occurrence j -> pos[j] = owner * cap + p (or -1 when the owner's segment overflowed) and
 *   send_rows[pos[j]] = local row; unused entries are -1.  peer_counts[world] = requests per owner;
 *   *overflow = max over owners of a count > cap (0: none; caller zeroes it once).  workspace: wd_route_chunks()*world ints.
 * wd_owner_gather: out[r*row_stride + 0..dim) = emb[rows[r]*dim ..] when rows[r] < n_emb_rows (emb may be NULL) and
 *   out[r*row_stride + (emb ? dim : 0)] = wide[rows[r]*4] (wide may be NULL); rows[r] < 0 skipped.
 * wd_owner_gather_rec: the same on the owner's row-record table (rec[row] = [emb dim f32 | w z n - | pad], rec_stride floats):
 *   out[r*row_stride + 0..dim+4) = rec[rows[r]*rec_stride + 0..dim+4) -- one copy of dim/4 + 1 float4 per request, the wide
 *   weight arrives in the line fetched for the row (python/lib/joint.py:140-143: the PS-side read of a partitioned variable).
 * wd_grad_pack: out[pos[j]*row_stride + 0..dim) = dx[b, out_col..] / len(bag) (zeros for non-embedding slots), and
 *   out[pos[j]*row_stride + (dx ? dim : 0)] = dlogit[b] for wide slots (0 otherwise). */
int32_t wd_route_chunks(void);
int wd_route_build(const wd_slot_t *local_slots, int32_t S, int32_t world, const int32_t *ids, const int32_t *bag_offs,
                   int64_t batch, int32_t cap, int32_t *send_rows, int32_t *pos, int32_t *workspace,
                   int32_t *peer_counts, int32_t *overflow, wd_stream_t stream);
int wd_owner_gather(const float *emb, int64_t n_emb_rows, int32_t dim, const float *wide, const int32_t *rows, int64_t n,
                    float *out, int32_t row_stride, wd_stream_t stream);
int wd_owner_gather_rec(const float *rec, int32_t rec_stride, int32_t dim, const int32_t *rows, int64_t n, float *out,
                        int32_t row_stride, wd_stream_t stream);
/* Sender-side unique of the row exchange (sharded engine, one id per bag): ONE request per distinct (owner, local row) of the
 * batch instead of one per occurrence.  sorted_pairs = the batch's (key << 32 | occurrence) pairs sorted ascending
 * (wd_bucket_onehot + wd_bucket_sort with requester-side slot descriptors: key = world * local_row_base(slot) + id, so that
 * owner = key % world and local row = key / world).  Fills the [world][cap] request segments (unused entries -1) with the
 * distinct rows in key order, pos[occurrence] = entry of the occurrence's row (-1: segment overflow), peer_counts and the
 * overflow flag like wd_route_build.  workspace: wd_route_chunks() * world ints. */
int wd_route_unique(const uint64_t *sorted_pairs, int64_t n, int32_t world, int32_t cap, int32_t *send_rows, int32_t *pos,
                    int32_t *workspace, int32_t *peer_counts, int32_t *overflow, wd_stream_t stream);
/* ... and the gradient records that go with it: out[pos[first occurrence of a row] * row_stride + ..] = [sum of dx[b, col_s ..]
 * over the row's occurrences (ascending bag order) | sum of dlogit[b]] -- wd_grad_pack + the owner's per-row sum of the
 * requester's share, done before the wire (python/lib/joint.py:233-262: the gradient of a row is the sum over its
 * occurrences).  long_list as written by wd_bucket_sort. */
int wd_row_grad_presum(const wd_slot_t *slots, int32_t S, int64_t batch, const float *dx, int64_t ldx, const float *dlogit,
                       int32_t dim, const uint64_t *pairs, const int32_t *long_list, int32_t long_capacity, const int32_t *pos,
                       float *out, int32_t row_stride, wd_stream_t stream);
int wd_grad_pack(const wd_slot_t *slots, int32_t S, const int32_t *bag_offs, const int32_t *pos, int64_t batch,
                 const float *dx, int64_t ldx, const float *dlogit, int32_t dim, int32_t row_stride, float *out,
                 wd_stream_t stream);
int wd_fill_i32(int32_t *p, int32_t v, int64_t n, wd_stream_t stream);

/* ---- a9: dense tower (python/lib/dnn.py:92-234) ---------------------------------------------
 * fp32 MFMA GEMMs, row-major.  C[M,N] = epi(A op B):
 *   NN: C = A[M,K] B[K,N]        (forward; epilogue: + bias[N], activation)
 *   NT: C (+)= A[M,K] B[N,K]^T   (input gradient)
 *   TN: Cpart[split][M,N] = A[K,M]^T B[K,N] over K-slices (weight gradient; partials reduced by wd_mlp_finalize) */
#define WD_ACT_NONE 0
#define WD_ACT_RELU 1
#define WD_ACT_SIGMOID 2
#define WD_ACT_TANH 3
#define WD_ACT_RELU6 4
#define WD_ACT_LEAKY_RELU 5
#define WD_ACT_ELU 6
#define WD_ACT_SELU 7
#define WD_ACT_SOFTPLUS 8
#define WD_ACT_SOFTSIGN 9

/* bias (may be NULL) is the sum of `bias_parts` vectors bias[p*N + n] (the partial sums wd_fold_affine emits). */
int wd_gemm_nn_bias_act(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, int32_t bias_parts,
                        int32_t act, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, wd_stream_t stream);
int wd_gemm_nt(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, int64_t M, int64_t N,
               int64_t K, int32_t accumulate, wd_stream_t stream);
/* NT with the activation derivative fused into the epilogue (simple connection mode):
 *   C[m,n] = (A B^T)[m,n] * act'(act_src[m*ld_act + n])   = dz of the producing layer, no separate elementwise pass. */
int wd_gemm_nt_actbwd(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, int64_t M,
                      int64_t N, int64_t K, const float *act_src, int64_t ld_act, int32_t act, wd_stream_t stream);
/* Cpart[split][(M + append_ones)][N] = A[Kslice, M]^T B[Kslice, N]; with append_ones the extra output row M is
 * ones^T B = the column sums of B (the bias gradient).  The caller provides nsplit*(M+append_ones)*N floats. */
int wd_gemm_tn_splitk(const float *A, int64_t lda, const float *B, int64_t ldb, float *Cpart, int64_t M, int64_t N,
                      int64_t K, int32_t nsplit, int32_t append_ones, wd_stream_t stream);

/* wd_gemm_tn_splitk for several layers in ONE launch (the weight-gradient products of a whole tower): job j is exactly
 * wd_gemm_tn_splitk(A, lda, B, ldb, Cpart, M, N, K, nsplit, append_ones). */
#define WD_TN_GROUP_MAX 20
typedef struct wd_tn_job {
  const float *A;
  const float *B;    /* NULL: column-sum job -- Cpart[n] = sum over the K rows of A [K][lda] (n < N), in row order */
  float *Cpart;
  int64_t lda, ldb, M, N, K;
  int32_t nsplit, append_ones;
} wd_tn_job_t;
int wd_gemm_tn_splitk_group(const wd_tn_job_t *jobs, int32_t njobs, wd_stream_t stream);

/* Dense parameters live in ONE flat fp32 buffer P (kernels [K,N] row-major, biases, BN gamma/beta); the
 * gradient buffer Gflat has the same layout.  gamma_idx[k] / beta_idx[k] give, for input column k of a
 * layer, the index in P of the BN gamma / beta of the unit that produced that column (-1: raw input column).
 *
 * BN is the inference-mode affine of SURVEY App. C.1 (python/lib/dnn.py:113-114), folded into the consumer:
 *   s[k] = gamma*inv (or 1), t[k] = beta (or 0), inv = 1/sqrt(1+eps);
 *   Wf[k,n] = s[k]*W[k,n];  bf[n] = b[n] + sum_k t[k]*W[k,n], emitted as WD_FOLD_PARTS partial vectors
 *   bf[p*N + n] (K split into WD_FOLD_PARTS chunks; fixed summation order, no atomics). */
#define WD_FOLD_PARTS 16
int wd_fold_affine(const float *P, int64_t w_off, int64_t b_off, const int32_t *gamma_idx, const int32_t *beta_idx,
                   float inv, float *Wf, float *bf, float *s, float *t, int64_t K, int64_t N, wd_stream_t stream);

/* Per-layer descriptor (device array) for the all-layers-in-one-launch variants below. */
typedef struct wd_mlp_layer {
  int64_t w_off, b_off;     /* kernel [K,N] / bias [N] offsets in P and Gflat */
  int64_t K, N;
  const int32_t *gamma_idx; /* [K] or NULL */
  const int32_t *beta_idx;  /* [K] or NULL */
  float *Wf, *bf, *s, *t;   /* folded outputs (as wd_fold_affine) */
  const float *Gpart;       /* split-K partials of this layer (as wd_mlp_finalize) */
  int32_t nsplit;
  int32_t pad_;
  /* optional IEEE-half copies of the folded kernel for the fp16-input tower (NULL: not written): */
  uint16_t *WfT_h;          /*   transposed kernel WfT_h [N][ld_wft_h] (operand of wd_hgemm_nn) */
  int64_t ld_wft_h;
  const int64_t *cat_off;   /*   [K] element offsets into wcat (< 0: skip): row k of this layer is written to */
  uint16_t *wcat;           /*   wcat[cat_off[k] + n], n < N -- the operand of the segment-gradient GEMM (wd_hgemm_nt) */
} wd_mlp_layer_t;

/* wd_fold_affine for every layer of every tower in ONE launch; also zero-fills up to two small buffers
 * (the step's loss accumulator / flat gradient buffer) so a step needs no separate fill launches. */
int wd_fold_affine_all(const float *P, const wd_mlp_layer_t *layers_dev, int32_t nlayers, int64_t max_n, float inv,
                       float *zero_a, int64_t zero_a_n, float *zero_b, int64_t zero_b_n, wd_stream_t stream);

/* Activation `crelu` (python/lib/utils/model_util.py:52: tf.nn.crelu = concat(relu(z), relu(-z)) on the feature axis): the
 * host runs the layer as a relu layer of width 2N with kernel [W | -W] [K][2N] and bias [b | -b]; after the weight gradient of
 * that 2N-wide layer is final in Gflat this ties it back: g = G[k][n] - G[k][N+n]; G[k][n] = g; G[k][N+n] = -g (same for the
 * bias), N = width of the TF variable. */
int wd_crelu_tie(float *Gflat, int64_t w_off, int64_t b_off, int64_t K, int64_t N, wd_stream_t stream);

/* wd_mlp_finalize for every layer in ONE launch.  Only valid when every BN gamma/beta feeds exactly one consumer
 * layer (connected_mode `simple`): the affine gradients are then stored, not accumulated, and Gflat needs no zeroing. */
int wd_mlp_finalize_all(const wd_mlp_layer_t *layers_dev, int32_t nlayers, int64_t max_k, const float *P, float inv,
                        float *Gflat, wd_stream_t stream);
/* wd_mlp_finalize_all + wd_adagrad_dense in one launch: every parameter is updated (accum += g^2; w -= lr g / sqrt(accum))
 * where its gradient becomes final; Gflat is still written.  Only when nothing sits between the two (single GPU: no
 * all-reduce of Gflat) and the dnn optimizer is Adagrad. */
int wd_mlp_finalize_adagrad_all(const wd_mlp_layer_t *layers_dev, int32_t nlayers, int64_t max_k, float *P, float *Pacc,
                                float inv, float *Gflat, float lr, wd_stream_t stream);

/* ---- logits layer + head fused (python/lib/dnn.py:226-232, python/lib/joint.py:216-222,264-269) ----
 * dnn_logit[b] = a[b, 0..K) . wf + sum(bf parts); logit = dnn_logit + wide_logit (may be NULL); sigmoid CE SUM into
 * loss_sum (+=), prob, dlogit = w*(p-y).  Backward of the logits layer in the same launch:
 *   out[b*ld_out + k] = dlogit[b]*wf[k]  (times act'(a[b,k]) when act != 0, i.e. out = dz of the last hidden layer),
 *   Gpart[blk*(K+1) + k] = sum over the block's examples of a[b,k]*dlogit[b], [.. + K] = sum dlogit  (split-K
 *   partials in wd_mlp_finalize's layout with nsplit = wd_logits_head_blocks(batch, K)).
 * labels NULL: forward only (predict); out / Gpart may be NULL. */
int64_t wd_logits_head_blocks(int64_t batch, int64_t K);
int wd_logits_head(const float *a, int64_t ld_a, int64_t K, const float *wf, const float *bf, int32_t bias_parts,
                   const float *wide_logit, const float *labels, const float *weights, int64_t batch,
                   float *dnn_logit, float *logit, float *prob, float *dlogit, float *loss_sum, float *out,
                   int64_t ld_out, int32_t act, float *Gpart, wd_stream_t stream);

/* ---- whole `simple` tower in one launch (csrc/mlp_chain.hip) -------------------------------------------------------
 * python/lib/dnn.py:92-141 (dense -> activation -> BN per hidden layer; BN = the inference affine of SURVEY App. C.1, applied
 * as written -- nothing is folded), dnn.py:226-232 (logits), python/lib/joint.py:216-222, 264-269 (joint logit, sigmoid CE) for
 * one row tile of 32 examples per workgroup (csrc/mlp_chain8.hip: 512 lanes, two wavefronts per SIMD), with s_l = gamma_l inv,
 * t_l = beta_l:
 *   a_l = act(bn_{l-1} W_l + b_l),  bn_l = s_l a_l + t_l  (bn_{-1} = x),  dnn_logit = bn_{L-1} . w_logits + b, head as
 *   wd_logits_head, and with labels  d(bn_{L-1}) = dlogit w_logits^T,  dz_l = d(bn_l) s_l act'(a_l),  d(bn_{l-1}) = dz_l W_l^T,
 *   dx = dz_0 W_0^T.
 * bn_l is written to layers[l].a_out (row stride ld_act), dz_l to layers[l].dz_out [batch][N_l] -- the operands of the
 * weight-gradient products wd_gemm_tn_splitk_group, which stay a separate launch -- the first dx_cols columns of dx to
 * dx[b*ld_dx + k], the logits-layer gradient partials to Gpart_logits ([tile][K_L + 1], nsplit = wd_tower_chain_blocks(batch,
 * row_tile)), and per-tile column sums for the bias / BN gradients (wd_chain_layer_t).  labels NULL: forward only.  Shapes: K0
 * and every N_l multiples of the row tile and wd_tower_chain_lds_bytes(K0, N, L, row_tile) > 0 (else the call fails: use the
 * per-layer GEMMs).  inv = 1 / sqrt(1 + eps). */
#define WD_CHAIN_MAX_LAYERS 6
typedef struct wd_chain_layer {
  const float *Wpk;  /* MFMA-packed kernel W [K][N] (layout below, written by wd_chain_tail) */
  const float *WTpk; /* MFMA-packed transposed kernel; may be NULL without labels */
  const float *bias; /* [N] (the raw variable; NULL: none) */
  const float *gamma, *beta;  /* [N] BN affine of this layer's output (python/lib/dnn.py:113-114), both NULL: no BN */
  float *a_out;      /* [batch] rows of ld_act: bn_l = gamma inv act(..) + beta -- what the next layer reads (the operand of its
                        weight-gradient product); the kernel keeps the un-normalised activation for act' on chip */
  float *dz_out;     /* [batch][N] gradient of the loss wrt this layer's pre-activation */
  /* optional [wd_tower_chain_blocks(batch, row_tile)][N] per-row-tile column sums, reduced by column-sum jobs of
   * wd_gemm_tn_splitk_group: db_part of dz (bias gradient), dgamma_part of d(bn) * a (x inv = gamma gradient), dbeta_part of d(bn) */
  float *db_part, *dgamma_part, *dbeta_part;
  int32_t K, N;
} wd_chain_layer_t;
/* A packed operand B [R reduction rows][C columns] (Wpk: R = K, C = N; WTpk = pack(W^T): R = N, C = K) is stored as
 *   pk[((c/32) * (R/8)  + r/8)  * 256 + ((r%2) * 32 + c%32) * 4 + (r%8)/2 ]  =  B[r][c]
 * i.e. one 16-byte load per lane feeds four consecutive v_mfma_f32_32x32x2_f32 steps of a 32-column tile.
 *
 * wd_chain_tail -- the dense tail of a step on the one-launch tower (python/lib/joint.py:233-241: Adagrad on the dnn scope).
 * Nothing is folded, so every dense parameter is on its own: per parameter the split-K partials of its gradient are summed in
 * split order (kernels: Gpart[z][(K | K+1)][N]; bias: db_sum[N] or the appended row K of the partials; BN: inv * dgamma_sum,
 * dbeta_sum -- finished column sums), mode & WD_TAIL_GRAD stores the gradient to Gflat, & WD_TAIL_UPDATE takes the Adagrad step
 * (without GRAD: from Gflat, e.g. after an all-reduce), & WD_TAIL_PACK rewrites Wpk / WTpk from the (updated) kernel.
 * One launch, one thread per parameter, no ordering between workgroups. */
#define WD_TAIL_GRAD 1
#define WD_TAIL_UPDATE 2
#define WD_TAIL_PACK 4
typedef struct wd_tail_layer {
  int64_t w_off, b_off, gamma_off, beta_off;   /* offsets in P / Gflat; gamma_off / beta_off < 0: none */
  int64_t K, N;
  const float *Gpart;        /* nsplit partials */
  const float *db_sum;       /* [N] or NULL (then the partials carry the bias-gradient row K) */
  const float *dgamma_sum, *dbeta_sum;   /* [N] (when gamma_off / beta_off >= 0) */
  float *Wpk, *WTpk;         /* NULL: not packed (the logits layer) */
  int32_t nsplit, pk_tile;   /* pk_tile: 32 (the only packing) */
  /* concatenating towers (wd_chain_windows_t): the segments of this layer's window, in window order -- rows
   * [seg_k0[q], seg_k0[q] + seg_w[q]) of the kernel go to the pull operand seg_wt[q] of that segment (reduction length
   * seg_kred[q]), kernel column n to its reduction row seg_red0[q] + n.  nseg = 0: WTpk = pack(W^T). */
  int32_t nseg, pad_;
  int32_t seg_k0[WD_CHAIN_MAX_LAYERS + 1], seg_w[WD_CHAIN_MAX_LAYERS + 1], seg_red0[WD_CHAIN_MAX_LAYERS + 1],
      seg_kred[WD_CHAIN_MAX_LAYERS + 1];
  float *seg_wt[WD_CHAIN_MAX_LAYERS + 1];
} wd_tail_layer_t;
int wd_chain_tail(const wd_tail_layer_t *layers, int32_t nlayers, float *P, float *Pacc, float *Gflat, float inv, float lr,
                  int32_t mode, wd_stream_t stream);
/* Optional (wd_chain_opts_t.input): fuse the input layer into the call (one-id-per-bag batches, the Criteo shape): the kernel
 * then builds its x tile itself -- x[b, out_col_s ..] = emb[emb_off_s + ids[b*S + s]*dim ..] for the slots
 * [slot0, slot0+ngroup) (id < 0: zeros), the numeric columns (wd_dense_fwd), and the wide logit
 * bias[0] + sum_s wide[(row_base_s + id)*4] (wd_wide_fwd; NULL: none) -- writes x to x_out (the weight-gradient GEMMs
 * read it) and the wide logit to wide_out, and ignores the x / wide_logit arguments of wd_tower_chain.  It replaces the
 * wd_input_layer_fwd launch (python/lib/dnn.py:88-90, python/lib/linear.py:29-36). */
#define WD_CHAIN_MAX_SLOTS 128
typedef struct wd_chain_input {
  const float *emb;
  const wd_slot_t *slots;
  const int32_t *ids;              /* [batch * S], one id per bag */
  const float *wide;               /* AoS {w, .., .., -} table or NULL */
  const float *wide_bias;
  float *wide_out;                 /* [batch] or NULL */
  const float *dense;              /* [batch][ld_dense] raw numeric features */
  const wd_dense_col_t *cols;
  float *x_out;                    /* [batch][ld_act] */
  int64_t ld_dense;
  int32_t S, slot0, ngroup, dim, ncols;
  /* rows that arrived through the exchange instead of the tables (sharded engine): ids = positions in a buffer whose
   * rows are row_stride floats apart (0: dim; slots then carry emb_off 0) and -- wide_in_row != 0 -- hold their wide
   * weight at [dim] (`wide` = the same buffer) */
  int32_t row_stride;
  int32_t wide_in_row, pad_;
} wd_chain_input_t;
/* Per-call options of wd_tower_chain (pass NULL for none; nothing is remembered between calls):
 *   input      fused input layer, above
 *   loss_part  store each row tile's loss to loss_part[tile] (wd_tower_chain_blocks(batch, row_tile) floats, plain stores) instead of
 *              adding it atomically to loss_sum -- the caller sums them in tile order (a column-sum job of
 *              wd_gemm_tn_splitk_group): a reproducible loss that needs no zeroed accumulator
 *   stamps     diagnostics: device uint64[192]; workgroups 0 and 100 write shader-clock stamps (start, x tile in LDS, after
 *              each forward layer, head, after each gradient stage, end) to [0..31] / [32..63]
 *   row_tile   examples per workgroup: 0 or 32 (v_mfma_f32_32x32x2_f32; the 16-row variant of rounds 1-3 is gone)
 *   tile_stamps diagnostics: device uint64[2 * wd_tower_chain_blocks]: every workgroup stores the constant-rate realtime
 *              clock (100 MHz, chip-wide) at its start and when its x tile is complete in LDS -- bench.py derives the
 *              in-step gather span from them */
/* Concatenating towers in the one launch (python/lib/dnn.py:155-193: dnn_connected_mode 'dense' -- every layer reads
 * [x | h_0 | .. | h_{l-1}] -- and 'resnet' -- [h_{l-1} | .. | h_0 | x], a concat, not an add: SURVEY App. C.8).  The activation row
 * holds every segment once, in an order that makes each layer's input ONE contiguous column window; the row tile in LDS mirrors
 * the row.  seg_col[0] = first column of x, seg_col[l + 1] = of hidden layer l's output; layer l reads layers[l].K columns from
 * in_col[l] on, the logits layer k_logits columns from in_col[L] on (w_logits: that many weights, in window order);
 * layers[l].Wpk = pack(W_l) with its K = the window width.  layers[l].WTpk = the PULL operand of segment l (x for l = 0, else
 * the output of hidden layer l - 1): pack(B) with B[r][c] = W_j[row of column c of the segment in layer j's window][n],
 * r = (N_l + .. + N_{j-1}) + n over the consumers j = l .. L-1 -- the gradient of a segment is one product over
 * [dz_l | .. | dz_{L-1}], plus dlogit x the segment's logits weights (added in the epilogue).  wd_chain_tail writes these
 * operands from wd_tail_layer_t.seg_*.  Needs N_0 + .. + N_{L-1} <= K0 (the dz tiles take the x region once x is dead), every
 * width a multiple of 32, x / wide_logit given (no fused input layer, no wide_vals). */
typedef struct wd_chain_windows {
  int32_t seg_col[WD_CHAIN_MAX_LAYERS + 1];
  int32_t in_col[WD_CHAIN_MAX_LAYERS + 1];
  int32_t k_logits;
  int32_t cols;          /* columns of the row tile: >= the end of every segment, multiple of 32 */
} wd_chain_windows_t;

typedef struct wd_chain_opts {
  const wd_chain_input_t *input;
  float *loss_part;
  void *stamps;
  void *tile_stamps;
  int32_t row_tile;
  int32_t flags;       /* A/B switches -- bit 2: plain instead of write-through stores of the HBM outputs; bit 3: s_setprio 3 for
                          the launch's wavefronts */
  /* wide logit from a per-occurrence weight list (wd_prefetch_onehot): wide_logit[b] = wide_bias[0] + sum_s wide_vals[b*wide_S + s],
   * slots in order; replaces the wide_logit argument (input must be NULL); also stored to wide_out when that is not NULL */
  const float *wide_vals;
  const float *wide_bias;
  float *wide_out;
  int32_t wide_S, pad_;
  const wd_chain_windows_t *windows;   /* NULL: every layer reads its predecessor only ('simple') */
} wd_chain_opts_t;
int64_t wd_tower_chain_lds_bytes(int32_t K0, const int32_t *N, int32_t L, int32_t row_tile);   /* -1: unsupported shape */
int64_t wd_tower_chain_windows_lds_bytes(const wd_chain_windows_t *windows, int32_t K0, const int32_t *N, int32_t L);
int64_t wd_tower_chain_blocks(int64_t batch, int32_t row_tile);   /* ceil(batch / row_tile) */
int wd_tower_chain(const float *x, int64_t ld_act, int32_t K0, const wd_chain_layer_t *layers, int32_t L, int32_t act,
                   float inv, const float *w_logits, const float *b_logits, const float *wide_logit,
                   const float *labels, const float *weights, int64_t batch, float *dnn_logit, float *logit,
                   float *prob, float *dlogit, float *loss_sum, float *Gpart_logits, float *dx, int64_t ld_dx,
                   int32_t dx_cols, const wd_chain_opts_t *opts, wd_stream_t stream);

/* ---- fp16-input MFMA tower (BASELINE configs[4]; csrc/mlp_half.hip).  wd_half_t = IEEE binary16 bit pattern.
 * Operands are half, reduction-contiguous; accumulation, bias, split-K partials and gradient accumulators are fp32.
 *   wd_hgemm_nn:  C = act(A WT^T + bias) written as half C [M][ldc] AND transposed CT [N][ldct] (CT may be NULL)
 *                 A [M][lda] (k contiguous), WT [N][ldw] (k contiguous) = WfT_h of wd_fold_affine_all
 *   wd_hgemm_nt:  X = dZ W^T with dZ [M][lddz] (n contiguous), W [N=K_l rows][ldw] = Wf_h; either
 *                 C32 != NULL: C32[m][n] (+)= X (fp32 gradient accumulator), or
 *                 C32 == NULL: Ch / CT = half(X * act'(act_src)) and its transpose (dz of the producing layer)
 *   wd_hgemm_tn_splitk: Gpart[z][(K+1)][N] = [AT ; 1] dZT^T over batch slices; AT [K][ldat], dZT [N][lddzt] (batch
 *                 contiguous); row K = column sums of dZ (bias gradient), as wd_gemm_tn_splitk
 *   wd_cast_transpose_h: dst[r][c] = half(src[r][c] * (act_h ? act'(act_h[r][c]) : 1)) and dstT[c][r] (either NULL)
 *   wd_logits_head_h: wd_logits_head with a half activation window
 * Operand loads are unconditional 16-byte loads clamped to the last 8-half vector of a row's reduction range (16-byte aligned
 * operands with pitches that are multiples of 8): up to 7 halfs behind the range are read and discarded, so an operand buffer
 * must extend at least 14 bytes past the reduction range of its LAST row (pitch >= the range rounded up to 8 does it). */
typedef uint16_t wd_half_t;
int wd_hgemm_nn(const wd_half_t *A, int64_t lda, const wd_half_t *WT, int64_t ldw, const float *bias, int32_t bias_parts,
                int32_t act, wd_half_t *C, int64_t ldc, wd_half_t *CT, int64_t ldct, int64_t M, int64_t N, int64_t K,
                wd_stream_t stream);
int wd_hgemm_nt(const wd_half_t *dZ, int64_t lddz, const wd_half_t *W, int64_t ldw, int64_t M, int64_t N, int64_t K,
                float *C32, int64_t ldc32, int32_t accumulate, wd_half_t *Ch, int64_t ldch, wd_half_t *CT, int64_t ldct,
                const wd_half_t *act_src, int64_t ld_act, int32_t act, wd_stream_t stream);
int wd_hgemm_tn_splitk(const wd_half_t *AT, int64_t ldat, const wd_half_t *dZT, int64_t lddzt, float *Gpart, int64_t K,
                       int64_t N, int64_t batch, int32_t nsplit, wd_stream_t stream);
int wd_cast_transpose_h(const float *src, int64_t ld_src, int64_t rows, int64_t cols, const wd_half_t *act_h,
                        int64_t ld_act, int32_t act, wd_half_t *dst, int64_t ld_dst, wd_half_t *dstT, int64_t ld_dstT,
                        wd_stream_t stream);
int wd_logits_head_h(const wd_half_t *a_h, int64_t ld_a, int64_t K, const float *wf, const float *bf, int32_t bias_parts,
                     const float *wide_logit, const float *labels, const float *weights, int64_t batch,
                     float *dnn_logit, float *logit, float *prob, float *dlogit, float *loss_sum, float *out,
                     int64_t ld_out, int32_t act, float *Gpart, wd_stream_t stream);

/* dz = da * act'(a), the derivative expressed through the activation output a. */
int wd_act_bwd(const float *da, int64_t ldda, const float *a, int64_t lda, int32_t act, float *dz, int64_t lddz,
               int64_t M, int64_t N, wd_stream_t stream);

/* ---- dropout (python/lib/dnn.py:111-112 and the same lines of every connected mode: tf.layers.dropout(net, rate,
 * training=True) after the activation and before BN, TRAIN mode only; ret = x / keep_prob * keep).
 * keep(b, n) is a counter-based function of seed_step = device {seed, step}, the layer and b*N + n:
 *     z = seed + step*0x632BE59BD9B4E019 + (layer+1)*0x9E3779B97F4A7C15 + (b*N+n)*0xD1B54A32D192ED03   (mod 2^64)
 *     z = (z ^ z>>30) * 0xBF58476D1CE4E5B9;  z = (z ^ z>>27) * 0x94D049BB133111EB;  z ^= z>>31
 *     keep = (z >> 40) / 2^24 >= rate
 * so no mask is stored: wd_dropout_fwd rewrites the activations a [M][N] in place, wd_act_bwd_dropout forms
 * dz = da / keep_prob * keep * act'(a_drop * keep_prob) from the same function, wd_counter_tick advances `step`
 * (on the device: a captured graph draws a new mask on every replay). */
int wd_dropout_fwd(float *a, int64_t lda, int64_t M, int64_t N, float rate, const int64_t *seed_step, int32_t layer,
                   wd_stream_t stream);
int wd_act_bwd_dropout(const float *da, int64_t ldda, const float *a_drop, int64_t lda, int32_t act, float *dz,
                       int64_t lddz, int64_t M, int64_t N, float rate, const int64_t *seed_step, int32_t layer,
                       wd_stream_t stream);
int wd_counter_tick(int64_t *seed_step, wd_stream_t stream);

/* Reduce the split-K partials G = sum_split Gpart ([K+1, N], row K = db) and un-fold the affine:
 *   Gflat[w_off + k*N + n] = s[k]*G[k,n] + t[k]*db[n];   Gflat[b_off + n] = db[n];
 *   Gflat[gamma_idx[k]] += inv * sum_n W[k,n]*G[k,n];     Gflat[beta_idx[k]] += sum_n W[k,n]*db[n]. */
int wd_mlp_finalize(const float *Gpart, int32_t nsplit, const float *P, int64_t w_off, int64_t b_off, const float *s,
                    const float *t, const int32_t *gamma_idx, const int32_t *beta_idx, float inv, float *Gflat,
                    int64_t K, int64_t N, wd_stream_t stream);

/* tf.train.AdagradOptimizer dense apply over a flat parameter buffer. */
int wd_adagrad_dense(float *w, float *accum, const float *g, int64_t n, float lr, wd_stream_t stream);

/* Diagnostic (bench-only): n random 64-byte row reads from `table` (row = 16 floats) with `per` independent rows in
 * flight per 4-lane group; writes one float per wavefront to out.  Measures the random-gather ceiling of the part. */
int wd_diag_gather64(const float *table, const int32_t *ids, int64_t n, int32_t per, float *out, wd_stream_t stream);
/* Diagnostic (bench-only): the gather of n rows (16 floats at records rec_stride floats apart) into out[n][16] with the request
 * issued in eight different ways (csrc/common.hip lists them: 4 / 8 / 1 lanes per row, nontemporal loads / stores, LDS-DMA,
 * read side alone, 4 rows in flight per lane group): the table behind profiles/r3_gather_modes.txt. */
int wd_diag_gather_modes(const float *rec, int64_t rec_stride, const int32_t *ids, int64_t n, int32_t mode, float *out,
                         wd_stream_t stream);

/* Diagnostic (bench-only): n items, item j touching bytes[s] bytes at base[s] + ids[j] * stride_bytes[s] in each of nseg <= 3
 * tables (16-byte pieces, <= 256 bytes per item, `per` items in flight per 16-lane group); rmw != 0 writes every piece back.
 * Prices table-row layouts (scripts/bench_layouts.py): separate embedding / accumulator / wide lines against one record. */
int wd_diag_access(float *const *base, const int64_t *stride_bytes, const int32_t *bytes, int32_t nseg, const int32_t *ids,
                   int64_t n, int32_t per, int32_t rmw, float *out, wd_stream_t stream);

/* misc plumbing */
int wd_fill_f32(float *p, float v, int64_t n, wd_stream_t stream);

/* ---- a7 / a8 / a12 for categorical columns whose whole table fits in LDS and whose bags are long: crossed columns over
 * multi-valued keys (python/lib/build_estimator.py:138-155: a bag holds the product of its keys' counts), their
 * embedding_column(combiner='mean') and linear_model weight, Adagrad / Ftrl with IndexedSlices semantics (joint.py:224-262).
 * small_idx[nsmall]: slot numbers (device); max_rows / max_dim over those slots, max_rows * (max_dim + 2) <=
 * WD_SMALL_MAX_FLOATS, max_dim <= 16.  rec_stride 0: separate tables (emb flat at the slot's emb_off, wide [rows][4] = {w, z, n, -});
 * rec_stride > 0 (round 6): row records -- `emb` = the record table, row r of a slot at (row_base + r) * rec_stride, `wide` = the
 * {w, z, n, -} part of record 0 (same stride); the Adagrad accumulator stays flat (the slot's emb_off) in both.
 *   wd_small_tables_fwd: x[b][out_col ..] = mean of the bag's rows; wide_logit[b] += sum of the bag's wide weights -- call it
 *     BEHIND wd_wide_fwd on the same stream (that launch writes bias + the other columns).
 *   wd_small_tables_bwd: per slot, count x (dx / len | dlogit) summed per row in ascending example order (integer histograms in
 *     LDS, no sort, no float atomics), then Adagrad (embedding rows) / Ftrl ({w, z, n}) on the rows the batch holds.
 *     ws: wd_small_tables_ws_floats(nsmall, max_rows, max_dim, max_batch) floats. */
#define WD_SMALL_MAX_FLOATS 8192
#define WD_SMALL_BAGS_PER_SLICE 64
int64_t wd_small_tables_ws_floats(int32_t nsmall, int32_t max_rows, int32_t max_dim, int64_t max_batch);
int wd_small_tables_fwd(const float *emb, const float *wide, const wd_slot_t *slots, int32_t S, const int32_t *small_idx,
                        int32_t nsmall, int32_t max_rows, int32_t max_dim, const int32_t *ids, const int32_t *bag_offs,
                        int64_t batch, float *x, int64_t ldx, float *wide_logit, int32_t rec_stride, wd_stream_t stream);
int wd_small_tables_bwd(float *emb, float *emb_accum, float *wide_wzn, const wd_slot_t *slots, int32_t S,
                        const int32_t *small_idx, int32_t nsmall, int32_t max_rows, int32_t max_dim, const int32_t *ids,
                        const int32_t *bag_offs, int64_t batch, const float *dx, int64_t ldx, const float *dlogit, float lr_emb,
                        float lr_wide, float l1, float l2, float *ws, int64_t ws_floats, int32_t rec_stride, wd_stream_t stream);
/* The same update in two halves, for tables REPLICATED on every rank of a row-sharded model (a 200-row crossed column is not
 * worth an all-to-all; the reference's parameter server would hold it on one task, python/lib/joint.py:140-143):
 *   wd_small_tables_grad: this rank's sums -- gsum[nsmall][max_rows][max_dim + 2] = per row g_0 .. g_{D-1}, g_wide, hit count of
 *     ITS OWN width D (the slice partials added in a fixed tree; batch == 0: zeros); dx / dlogit NULL: that half is not computed;
 *   the ranks all-reduce(SUM) gsum; wd_small_tables_apply: Adagrad / Ftrl from the sums on the rows some rank's batch holds. */
int wd_small_tables_grad(const wd_slot_t *slots, int32_t S, const int32_t *small_idx, int32_t nsmall, int32_t max_rows,
                         int32_t max_dim, const int32_t *ids, const int32_t *bag_offs, int64_t batch, const float *dx, int64_t ldx,
                         const float *dlogit, float *ws, int64_t ws_floats, float *gsum, wd_stream_t stream);
int wd_small_tables_apply(float *emb, float *emb_accum, float *wide_wzn, const wd_slot_t *slots, int32_t S,
                          const int32_t *small_idx, int32_t nsmall, int32_t max_rows, int32_t max_dim, const float *gsum,
                          float lr_emb, float lr_wide, float l1, float l2, int32_t rec_stride, wd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WD_HIP_H_ */
