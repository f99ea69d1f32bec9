/*
 * include/wd_ingest.h -- C ABI of the host-side TSV ingest (wide_deep_amd/_lib/libwd_ingest.so, csrc/tsv_ingest.c).
 *
 * Replaces the tf.data parser the reference builds per element (python/lib/dataset.py:133-164: decode_csv with
 * field_delim '\t', use_quote_delim=False, na_value '-', record_defaults per field type; tf.string_split(',') for
 * multi-value fields; label = clk == 1).  Host memory only; no GPU code.  A batch is parsed in two passes (count, fill)
 * so that the caller can allocate exact-size outputs; string features are emitted directly in the packed layout
 * wd_fingerprint64 (include/wd_hip.h) consumes.
 */
#ifndef WD_INGEST_H_
#define WD_INGEST_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define WD_TSV_OK 0
#define WD_TSV_FIELDS (-1) /* a line does not have nfields tab-separated fields */
#define WD_TSV_INT (-2)    /* an integer field is neither NA nor a base-10 integer */
#define WD_TSV_FLOAT (-3)  /* a float field is neither NA nor a number (strtod) */

/* Line start offsets of buf[0..len): out[i] = start of line i, out[n] = end; returns n (<= max_lines). */
int64_t wd_tsv_scan(const uint8_t *buf, int64_t len, int64_t *out, int64_t max_lines);

/* Pass 1: per string feature (field index str_cols[f]) the number of tokens and token bytes over the lines
 * [starts[i], ends[i]).  NA ('-' or empty) gives no token; multivalue != 0 splits on ',' and skips empty pieces. */
int wd_tsv_count(const uint8_t *buf, const int64_t *starts, const int64_t *ends, int64_t nlines, int32_t nfields,
                 const int32_t *str_cols, int32_t n_str, int32_t multivalue, int64_t *ntok, int64_t *nbytes,
                 int64_t *err_line);

/* Pass 2: feature f's tokens go to tokens [tok_base[f], ..) / bytes [byte_base[f], ..) of the shared arrays
 * (tok_offs = absolute byte offsets); ex_offs [n_str][nlines + 1] = per-example CSR (relative to tok_base[f]);
 * ints [n_int][nlines] (NA -> 0), flts [n_flt][nlines] (NA -> 0.0; (float)strtod), labels [nlines] = field
 * label_col == 1 (label_col < 0: no label field). */
int wd_tsv_fill(const uint8_t *buf, const int64_t *starts, const int64_t *ends, int64_t nlines, int32_t nfields,
                const int32_t *str_cols, int32_t n_str, int32_t multivalue, const int64_t *tok_base,
                const int64_t *byte_base, uint8_t *tok_bytes, int32_t *tok_offs, int32_t *ex_offs, const int32_t *int_cols,
                int32_t n_int, int32_t *ints, const int32_t *flt_cols, int32_t n_flt, float *flts, int32_t label_col,
                float *labels, int64_t *err_line);

/* categorical_column_with_vocabulary_list (python/lib/build_estimator.py:101-106): out[t - t0] = index of token t in
 * the packed vocabulary, -1 when absent (default_value=-1, dropped by the sparse conversion). */
void wd_vocab_lookup(const uint8_t *tok_bytes, const int32_t *tok_offs, int64_t t0, int64_t t1, const uint8_t *vocab_bytes,
                     const int32_t *vocab_offs, int32_t nvocab, int32_t *out);

/* CRC-32C of data[0..n) continued from `crc` (0 to start; unmasked) -- the per-entry / per-block checksum of TensorFlow's
 * checkpoint container (wide_deep_amd/tf_checkpoint.py: python/train.py:188-191 leaves such files in model_dir). */
uint32_t wd_crc32c(const uint8_t *data, int64_t n, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* WD_INGEST_H_ */
