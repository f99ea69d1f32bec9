/* wide_deep_amd/csrc/tsv_ingest.c -- host-side TSV ingest of the train step's input.
 *
 * Replaces the per-element tf.data parser of the reference (python/lib/dataset.py:133-164: decode_csv with
 * field_delim '\t', na_value '-', record_defaults per field type; tf.string_split(',') for multi-value fields;
 * label = clk == 1) for a whole batch of lines in two passes over the raw bytes, and packs every string feature's
 * tokens straight into the layout the GPU hash kernel reads (wd_fingerprint64: one byte buffer + token offsets),
 * so no Python object is created per field or per token.
 *
 * Plain C, no dependencies; built by wide_deep_amd/csrc/build.sh into _lib/libwd_ingest.so and bound with ctypes
 * (wide_deep_amd/dataset.py).  Semantics are pinned against the pure-Python parser of dataset.py and the oracle's own
 * parser (oracle/columns.py) in tests/test_conf_dataset.py.
 */
#include <errno.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define WD_TSV_OK 0
#define WD_TSV_FIELDS (-1) /* wrong number of fields in a line  */
#define WD_TSV_INT (-2)    /* an integer field does not parse   */
#define WD_TSV_FLOAT (-3)  /* a float field does not parse      */

/* line start offsets of buf[0..len): out[0] = 0, out[i] = byte after the i-th '\n'; returns the number of lines
 * (a last line without '\n' counts; an empty tail does not).  out must hold max_lines + 1 entries. */
int64_t wd_tsv_scan(const uint8_t *buf, int64_t len, int64_t *out, int64_t max_lines) {
  int64_t n = 0, pos = 0;
  while (pos < len && n < max_lines) {
    out[n++] = pos;
    const uint8_t *nl = (const uint8_t *)memchr(buf + pos, '\n', (size_t)(len - pos));
    pos = nl ? (int64_t)(nl - buf) + 1 : len;
  }
  out[n] = pos;
  return n;
}

static inline int is_na(const uint8_t *p, int64_t n) { return n == 0 || (n == 1 && p[0] == '-'); }

/* field boundaries of one line [s, e): fills off[0..nfields] (off[i]..off[i+1]-1 is field i incl. its tab); returns
 * the number of fields found (capped at maxf + 1) */
static int split_fields(const uint8_t *buf, int64_t s, int64_t e, int64_t *off, int maxf) {
  /* one pass over the bytes of the line (fields are a few bytes each: a memchr call per field cost more than the scan) */
  int nf = 0;
  off[0] = s;
  for (int64_t pos = s; pos < e; ++pos) {
    if (buf[pos] == '\t') {
      ++nf;
      if (nf > maxf) return nf;
      off[nf] = pos + 1;
    }
  }
  ++nf;
  if (nf <= maxf) off[nf] = e + 1;
  return nf;
}

/* Decimal fields without a library call where that is exact: [sign] digits [. digits] [e [sign] digits] with at most 19
 * significant digits, mantissa <= 2^53 and |decimal exponent| <= 22 is ONE correctly rounded double operation on two exactly
 * representable doubles (Clinger's fast path) -- the value strtod returns.  Anything else (longer mantissas, inf / nan / hex,
 * blanks, stray characters) returns 0 and the caller takes strtod / strtol as before. */
static const double wd_p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                  1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

static int fast_double(const uint8_t *p, int64_t n, double *out) {
  int64_t i = 0;
  int neg = 0;
  if (n <= 0) return 0;
  if (p[0] == '-' || p[0] == '+') { neg = p[0] == '-'; i = 1; }
  uint64_t m = 0;
  int nd = 0, any = 0, e10 = 0;
  for (; i < n && p[i] >= '0' && p[i] <= '9'; ++i) {
    any = 1;
    if (m || p[i] != '0') { if (++nd > 19) return 0; m = m * 10 + (uint64_t)(p[i] - '0'); }
  }
  if (i < n && p[i] == '.') {
    for (++i; i < n && p[i] >= '0' && p[i] <= '9'; ++i) {
      any = 1;
      if (m || p[i] != '0') { if (++nd > 19) return 0; m = m * 10 + (uint64_t)(p[i] - '0'); }
      --e10;
    }
  }
  if (!any) return 0;
  if (i < n && (p[i] == 'e' || p[i] == 'E')) {
    int eneg = 0, ev = 0, ed = 0;
    ++i;
    if (i < n && (p[i] == '-' || p[i] == '+')) { eneg = p[i] == '-'; ++i; }
    for (; i < n && p[i] >= '0' && p[i] <= '9'; ++i) { if (++ed > 4) return 0; ev = ev * 10 + (p[i] - '0'); }
    if (!ed) return 0;
    e10 += eneg ? -ev : ev;
  }
  if (i != n) return 0;
  if (m == 0) { *out = neg ? -0.0 : 0.0; return 1; }
  if (m > ((uint64_t)1 << 53) || e10 < -22 || e10 > 22) return 0;
  double d = (double)m;
  d = e10 < 0 ? d / wd_p10[-e10] : d * wd_p10[e10];
  *out = neg ? -d : d;
  return 1;
}

static int fast_long(const uint8_t *p, int64_t n, long *out) {
  int64_t i = 0;
  int neg = 0;
  if (n <= 0 || n > 18) return 0;
  if (p[0] == '-' || p[0] == '+') { neg = p[0] == '-'; i = 1; }
  if (i == n) return 0;
  long v = 0;
  for (; i < n; ++i) {
    if (p[i] < '0' || p[i] > '9') return 0;
    v = v * 10 + (p[i] - '0');
  }
  *out = neg ? -v : v;
  return 1;
}

static inline int64_t line_end(const uint8_t *buf, int64_t s, int64_t e) {
  /* strip the trailing "\n" / "\r\n" of [s, e) */
  while (e > s && (buf[e - 1] == '\n' || buf[e - 1] == '\r')) --e;
  return e;
}

/* pass 1: tokens and token bytes per string feature over the given lines */
int wd_tsv_count(const uint8_t *buf, const int64_t *starts, const int64_t *ends, int64_t nlines, int32_t nfields,
                 const int32_t *str_cols, int32_t n_str, int32_t multivalue, int64_t *ntok, int64_t *nbytes,
                 int64_t *err_line) {
  int64_t off[512];
  if (nfields > 510) return WD_TSV_FIELDS;
  for (int f = 0; f < n_str; ++f) ntok[f] = nbytes[f] = 0;
  for (int64_t i = 0; i < nlines; ++i) {
    const int64_t s = starts[i], e = line_end(buf, starts[i], ends[i]);
    if (split_fields(buf, s, e, off, nfields) != nfields) {
      if (err_line) *err_line = i;
      return WD_TSV_FIELDS;
    }
    for (int f = 0; f < n_str; ++f) {
      const uint8_t *p = buf + off[str_cols[f]];
      const int64_t n = off[str_cols[f] + 1] - 1 - off[str_cols[f]];
      if (is_na(p, n)) continue;
      if (!multivalue) {
        ntok[f] += 1;
        nbytes[f] += n;
        continue;
      }
      int64_t a = 0;
      while (a <= n) {
        const uint8_t *c = a < n ? (const uint8_t *)memchr(p + a, ',', (size_t)(n - a)) : NULL;
        const int64_t b = c ? (int64_t)(c - p) : n;
        if (b > a) { /* empty pieces are skipped (tf.string_split default) */
          ntok[f] += 1;
          nbytes[f] += b - a;
        }
        if (!c) break;
        a = b + 1;
      }
    }
  }
  return WD_TSV_OK;
}

/* pass 2: fill.  Token stream of feature f occupies tokens [tok_base[f], tok_base[f] + ntok[f]) of the shared arrays
 * (tok_offs holds ABSOLUTE byte offsets into tok_bytes, tok_offs[total] = total bytes); ex_offs is [n_str][nlines + 1]
 * (token index relative to the feature's base).  ints [n_int][nlines], flts [n_flt][nlines], labels [nlines]
 * (label_col < 0: none). */
int wd_tsv_fill(const uint8_t *buf, const int64_t *starts, const int64_t *ends, int64_t nlines, int32_t nfields,
                const int32_t *str_cols, int32_t n_str, int32_t multivalue, const int64_t *tok_base,
                const int64_t *byte_base, uint8_t *tok_bytes, int32_t *tok_offs, int32_t *ex_offs,
                const int32_t *int_cols, int32_t n_int, int32_t *ints, const int32_t *flt_cols, int32_t n_flt,
                float *flts, int32_t label_col, float *labels, int64_t *err_line) {
  int64_t off[512];
  int64_t tk[256], by[256];
  if (nfields > 510 || n_str > 256) return WD_TSV_FIELDS;
  for (int f = 0; f < n_str; ++f) {
    tk[f] = tok_base[f];
    by[f] = byte_base[f];
    ex_offs[(int64_t)f * (nlines + 1)] = 0;
  }
  for (int64_t i = 0; i < nlines; ++i) {
    const int64_t s = starts[i], e = line_end(buf, starts[i], ends[i]);
    if (split_fields(buf, s, e, off, nfields) != nfields) {
      if (err_line) *err_line = i;
      return WD_TSV_FIELDS;
    }
    for (int f = 0; f < n_str; ++f) {
      const uint8_t *p = buf + off[str_cols[f]];
      const int64_t n = off[str_cols[f] + 1] - 1 - off[str_cols[f]];
      if (!is_na(p, n)) {
        int64_t a = 0;
        while (a <= n) {
          const uint8_t *c = (multivalue && a < n) ? (const uint8_t *)memchr(p + a, ',', (size_t)(n - a)) : NULL;
          const int64_t b = c ? (int64_t)(c - p) : n;
          if (b > a) {
            tok_offs[tk[f]] = (int32_t)by[f];
            memcpy(tok_bytes + by[f], p + a, (size_t)(b - a));
            by[f] += b - a;
            tk[f] += 1;
          }
          if (!c) break;
          a = b + 1;
        }
      }
      ex_offs[(int64_t)f * (nlines + 1) + i + 1] = (int32_t)(tk[f] - tok_base[f]);
    }
    for (int f = 0; f < n_int; ++f) {
      const uint8_t *p = buf + off[int_cols[f]];
      const int64_t n = off[int_cols[f] + 1] - 1 - off[int_cols[f]];
      int32_t v = 0;
      long fl;
      if (!is_na(p, n) && fast_long(p, n, &fl)) {
        v = (int32_t)fl;
      } else if (!is_na(p, n)) {
        char tmp[32];
        if (n >= (int64_t)sizeof(tmp)) { if (err_line) *err_line = i; return WD_TSV_INT; }
        memcpy(tmp, p, (size_t)n);
        tmp[n] = 0;
        char *endp = NULL;
        errno = 0;
        long lv = strtol(tmp, &endp, 10);
        if (errno || endp != tmp + n) { if (err_line) *err_line = i; return WD_TSV_INT; }
        v = (int32_t)lv;
      }
      ints[(int64_t)f * nlines + i] = v;
    }
    for (int f = 0; f < n_flt; ++f) {
      const uint8_t *p = buf + off[flt_cols[f]];
      const int64_t n = off[flt_cols[f] + 1] - 1 - off[flt_cols[f]];
      float v = 0.0f;
      double fd;
      if (!is_na(p, n) && fast_double(p, n, &fd)) {
        v = (float)fd;
      } else if (!is_na(p, n)) {
        char tmp[64];
        if (n >= (int64_t)sizeof(tmp)) { if (err_line) *err_line = i; return WD_TSV_FLOAT; }
        memcpy(tmp, p, (size_t)n);
        tmp[n] = 0;
        char *endp = NULL;
        double d = strtod(tmp, &endp); /* Python float() then np.float32(): the same double -> float rounding */
        if (endp != tmp + n) { if (err_line) *err_line = i; return WD_TSV_FLOAT; }
        v = (float)d;
      }
      flts[(int64_t)f * nlines + i] = v;
    }
    if (label_col >= 0) {
      const uint8_t *p = buf + off[label_col];
      const int64_t n = off[label_col + 1] - 1 - off[label_col];
      float y = 0.0f;
      long fl;
      if (!is_na(p, n) && fast_long(p, n, &fl)) {
        y = fl == 1 ? 1.0f : 0.0f;
      } else if (!is_na(p, n)) {
        char tmp[32];
        if (n >= (int64_t)sizeof(tmp)) { if (err_line) *err_line = i; return WD_TSV_INT; }
        memcpy(tmp, p, (size_t)n);
        tmp[n] = 0;
        char *endp = NULL;
        long lv = strtol(tmp, &endp, 10);
        if (endp != tmp + n) { if (err_line) *err_line = i; return WD_TSV_INT; }
        y = lv == 1 ? 1.0f : 0.0f;
      }
      labels[i] = y;
    }
  }
  return WD_TSV_OK;
}

/* vocabulary_list lookup over a packed token range: out[t - t0] = index of token t in the vocabulary or -1.
 * Vocabularies of the reference are tiny (2-55 entries, conf/feature.yaml): linear scan with length pre-check. */
void wd_vocab_lookup(const uint8_t *tok_bytes, const int32_t *tok_offs, int64_t t0, int64_t t1, const uint8_t *vocab_bytes,
                     const int32_t *vocab_offs, int32_t nvocab, int32_t *out) {
  for (int64_t t = t0; t < t1; ++t) {
    const uint8_t *p = tok_bytes + tok_offs[t];
    const int32_t n = tok_offs[t + 1] - tok_offs[t];
    int32_t hit = -1;
    for (int32_t v = 0; v < nvocab; ++v) {
      const int32_t m = vocab_offs[v + 1] - vocab_offs[v];
      if (m == n && memcmp(p, vocab_bytes + vocab_offs[v], (size_t)n) == 0) {
        hit = v;
        break;
      }
    }
    out[t - t0] = hit;
  }
}


/* CRC-32C (Castagnoli, reflected 0x82F63B78), slice-by-8: the checksum TensorFlow's tensor bundle stores per entry and per table
 * block (tensorflow/core/lib/hash/crc32c).  crc = running value (0 to start); returns the updated CRC (not masked). */
static uint32_t crc_tab[8][256];
static int crc_ready = 0;

static void crc_init(void) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) crc_tab[t][i] = (crc_tab[t - 1][i] >> 8) ^ crc_tab[0][crc_tab[t - 1][i] & 0xff];
  crc_ready = 1;
}

uint32_t wd_crc32c(const uint8_t *data, int64_t n, uint32_t crc) {
  if (!crc_ready) crc_init();
  uint32_t c = crc ^ 0xFFFFFFFFu;
  int64_t i = 0;
  for (; i + 8 <= n; i += 8) {
    const uint32_t lo = ((uint32_t)data[i] | (uint32_t)data[i + 1] << 8 | (uint32_t)data[i + 2] << 16 | (uint32_t)data[i + 3] << 24) ^ c;
    const uint32_t hi = (uint32_t)data[i + 4] | (uint32_t)data[i + 5] << 8 | (uint32_t)data[i + 6] << 16 | (uint32_t)data[i + 7] << 24;
    c = crc_tab[7][lo & 0xff] ^ crc_tab[6][(lo >> 8) & 0xff] ^ crc_tab[5][(lo >> 16) & 0xff] ^ crc_tab[4][lo >> 24] ^
        crc_tab[3][hi & 0xff] ^ crc_tab[2][(hi >> 8) & 0xff] ^ crc_tab[1][(hi >> 16) & 0xff] ^ crc_tab[0][hi >> 24];
  }
  for (; i < n; ++i) c = crc_tab[0][(c ^ data[i]) & 0xff] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
