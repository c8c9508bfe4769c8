/*
 * ORACLE — test infrastructure only (see ora_needle.h).  Literal C restatement of the in-tree parts
 * of needle's analyze/search hot path.  Kept deliberately close to the Rust control flow, including
 * its cost structure (full DP table per pair), because bench.py times it as the CPU baseline.
 */
#include "ora_needle.h"

#include <malloc.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ora_chromaprint.h"

/* ================================================================================================
 * std::time::Duration float conversions (library/core/src/time.rs, try_from_secs! macro):
 * decompose the float, compute secs/nanos with integer arithmetic, round half to even.
 * ================================================================================================ */
typedef unsigned __int128 u128;

static ora_ns from_secs_bits(uint64_t bits, int mant_bits, int exp_bits, int offset) {
  const int min_exp = 1 - (1 << exp_bits) / 2;
  const uint64_t mant_mask = ((uint64_t)1 << mant_bits) - 1;
  const uint64_t exp_mask = ((uint64_t)1 << exp_bits) - 1;
  uint64_t mant = (bits & mant_mask) | (mant_mask + 1);
  int exp = (int)((bits >> mant_bits) & exp_mask) + min_exp;
  uint64_t secs;
  uint32_t nanos;
  if (exp < -31) {
    return 0;
  } else if (exp < 0) {
    u128 t = (u128)mant << (offset + exp);
    int nanos_offset = mant_bits + offset;
    u128 tmp = (u128)1000000000u * t;
    nanos = (uint32_t)(tmp >> nanos_offset);
    u128 rem_mask = ((u128)1 << nanos_offset) - 1;
    u128 rem_msb_mask = (u128)1 << (nanos_offset - 1);
    u128 rem = tmp & rem_mask;
    int is_tie = rem == rem_msb_mask;
    int is_even = (nanos & 1) == 0;
    int rem_msb = (tmp & rem_msb_mask) == 0;
    int add_ns = !(rem_msb || (is_even && is_tie));
    nanos += (uint32_t)add_ns;
    secs = 0;
    if (nanos == 1000000000u) {
      secs = 1;
      nanos = 0;
    }
  } else if (exp < mant_bits) {
    secs = mant >> (mant_bits - exp);
    u128 t = (u128)((mant << exp) & mant_mask);
    int nanos_offset = mant_bits;
    u128 tmp = (u128)1000000000u * t;
    nanos = (uint32_t)(tmp >> nanos_offset);
    u128 rem_mask = ((u128)1 << nanos_offset) - 1;
    u128 rem_msb_mask = (u128)1 << (nanos_offset - 1);
    u128 rem = tmp & rem_mask;
    int is_tie = rem == rem_msb_mask;
    int is_even = (nanos & 1) == 0;
    int rem_msb = (tmp & rem_msb_mask) == 0;
    int add_ns = !(rem_msb || (is_even && is_tie));
    nanos += (uint32_t)add_ns;
    if (nanos == 1000000000u) {
      secs += 1;
      nanos = 0;
    }
  } else if (exp < 64) {
    secs = mant << (exp - mant_bits);
    nanos = 0;
  } else {
    abort(); /* Rust: panics "value is either too big or NaN" */
  }
  return secs * 1000000000ull + nanos;
}

ora_ns ora_duration_from_secs_f32(float s) {
  if (!(s >= 0.0f)) abort(); /* Rust panics on negative / NaN */
  uint32_t b;
  memcpy(&b, &s, 4);
  return from_secs_bits(b, 23, 8, 41);
}

ora_ns ora_duration_from_secs_f64(double s) {
  if (!(s >= 0.0)) abort();
  uint64_t b;
  memcpy(&b, &s, 8);
  return from_secs_bits(b, 52, 11, 44);
}

float ora_duration_as_secs_f32(ora_ns d) {
  uint64_t secs = d / 1000000000ull;
  uint32_t nanos = (uint32_t)(d % 1000000000ull);
  volatile float a = (float)secs;
  volatile float b = (float)nanos / 1000000000.0f;
  return a + b;
}

ora_ns ora_duration_mul_f32(ora_ns d, float rhs) {
  volatile float p = rhs * ora_duration_as_secs_f32(d);
  return ora_duration_from_secs_f32(p);
}

/* ================================================================================================
 * analyzer.rs:288-323
 * ================================================================================================ */
size_t ora_step_and_timestamp(const uint32_t *raw, size_t n_raw, ora_ns hash_duration, int delay_ms,
                              int item_ms, ora_ns seek_to, int has_seek, ora_hash_ts *out,
                              size_t cap, int *err) {
  if (err) *err = 0;
  ora_ns delay = (ora_ns)delay_ms * 1000000ull;       /* Duration::from_millis, chromaprint-rust */
  ora_ns item = (ora_ns)item_ms * 1000000ull;
  /* :293-297  hash_duration.as_millis() as usize / item_duration.as_millis() as usize */
  size_t hd_ms = (size_t)(hash_duration / 1000000ull);
  size_t it_ms = (size_t)(item / 1000000ull);
  if (it_ms == 0) {
    if (err) *err = 1; /* Rust: division by zero panic */
    return 0;
  }
  size_t step_by = hd_ms / it_ms;
  if (step_by == 0) {
    if (err) *err = 1; /* Rust: Iterator::step_by(0) panics */
    return 0;
  }
  size_t k = 0;
  for (size_t i = 0; i < n_raw; i += step_by) {
    /* :309  ts = delay + item_duration.mul_f32(i as f32) */
    ora_ns ts = delay + ora_duration_mul_f32(item, (float)i);
    if (has_seek) ts += seek_to; /* :314-318 */
    if (k < cap) {
      out[k].hash = raw[i];
      out[k].ts = ts;
    }
    k++;
  }
  return k;
}

/* ================================================================================================
 * MD5 (RFC 1321) — util.rs:99-105 formats md5::compute(first 8192 bytes) as lowercase hex
 * ================================================================================================ */
static uint32_t rol(uint32_t x, int c) { return (x << c) | (x >> (32 - c)); }

void ora_md5_hex(const uint8_t *data, size_t n, char out[33]) {
  static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22,
                            5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                            4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                            6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
  uint32_t K[64];
  for (int i = 0; i < 64; i++) K[i] = (uint32_t)floor(fabs(sin((double)(i + 1))) * 4294967296.0);
  uint32_t a0 = 0x67452301u, b0 = 0xefcdab89u, c0 = 0x98badcfeu, d0 = 0x10325476u;
  size_t padded = ((n + 8) / 64 + 1) * 64;
  uint8_t *msg = (uint8_t *)calloc(padded, 1);
  memcpy(msg, data, n);
  msg[n] = 0x80;
  uint64_t bitlen = (uint64_t)n * 8;
  for (int i = 0; i < 8; i++) msg[padded - 8 + i] = (uint8_t)(bitlen >> (8 * i));
  for (size_t off = 0; off < padded; off += 64) {
    uint32_t M[16];
    for (int i = 0; i < 16; i++)
      M[i] = (uint32_t)msg[off + 4 * i] | ((uint32_t)msg[off + 4 * i + 1] << 8) |
             ((uint32_t)msg[off + 4 * i + 2] << 16) | ((uint32_t)msg[off + 4 * i + 3] << 24);
    uint32_t A = a0, B = b0, C = c0, D = d0;
    for (int i = 0; i < 64; i++) {
      uint32_t F;
      int g;
      if (i < 16) {
        F = (B & C) | (~B & D);
        g = i;
      } else if (i < 32) {
        F = (D & B) | (~D & C);
        g = (5 * i + 1) % 16;
      } else if (i < 48) {
        F = B ^ C ^ D;
        g = (3 * i + 5) % 16;
      } else {
        F = C ^ (B | ~D);
        g = (7 * i) % 16;
      }
      F = F + A + K[i] + M[g];
      A = D;
      D = C;
      C = B;
      B = B + rol(F, S[i]);
    }
    a0 += A;
    b0 += B;
    c0 += C;
    d0 += D;
  }
  free(msg);
  uint32_t v[4] = {a0, b0, c0, d0};
  for (int i = 0; i < 16; i++) sprintf(out + 2 * i, "%02x", (unsigned)((v[i / 4] >> (8 * (i % 4))) & 0xff));
  out[32] = 0;
}

int ora_header_md5(const char *path, char out[33]) {
  uint8_t buf[8 * 1024];
  FILE *f = fopen(path, "rb");
  if (!f) return 1;
  size_t got = fread(buf, 1, sizeof(buf), f);
  fclose(f);
  if (got != sizeof(buf)) return 1; /* read_exact: UnexpectedEof */
  ora_md5_hex(buf, sizeof(buf), out);
  return 0;
}

/* ================================================================================================
 * bincode 1.3 (default: little endian, fixint) image of FrameHashes — data.rs:15-26,60-80
 *   u32 variant index of FrameHashesVersion::V1 (= 0, not the 12345 discriminant)
 *   u32 variant index of FrameHashesData::V1 (= 0)
 *   Vec<(u32, Duration)> opening : u64 len, then {u32 hash, u64 secs, u32 nanos} each
 *   Vec<(u32, Duration)> ending
 *   Duration hash_duration : u64 secs, u32 nanos
 *   String md5 : u64 len + bytes
 * ================================================================================================ */
static void put(FILE *f, const void *p, size_t n) { fwrite(p, 1, n, f); }

static void put_vec(FILE *f, const ora_hash_ts *v, size_t n) {
  uint64_t len = n;
  put(f, &len, 8);
  for (size_t i = 0; i < n; i++) {
    uint64_t secs = v[i].ts / 1000000000ull;
    uint32_t nanos = (uint32_t)(v[i].ts % 1000000000ull);
    put(f, &v[i].hash, 4);
    put(f, &secs, 8);
    put(f, &nanos, 4);
  }
}

int ora_frame_hashes_write(const char *path, const ora_frame_hashes *fh) {
  FILE *f = fopen(path, "wb");
  if (!f) return 1;
  uint32_t zero = 0;
  put(f, &zero, 4);
  put(f, &zero, 4);
  put_vec(f, fh->opening, fh->n_opening);
  put_vec(f, fh->ending, fh->n_ending);
  uint64_t secs = fh->hash_duration / 1000000000ull;
  uint32_t nanos = (uint32_t)(fh->hash_duration % 1000000000ull);
  put(f, &secs, 8);
  put(f, &nanos, 4);
  uint64_t len = strlen(fh->md5);
  put(f, &len, 8);
  put(f, fh->md5, len);
  int bad = ferror(f);
  fclose(f);
  return bad ? 1 : 0;
}

typedef struct {
  const uint8_t *p;
  size_t n, off;
  int bad;
} rd_t;

static void get(rd_t *r, void *out, size_t n) {
  if (r->bad || r->off + n > r->n) {
    r->bad = 1;
    memset(out, 0, n);
    return;
  }
  memcpy(out, r->p + r->off, n);
  r->off += n;
}

static int get_vec(rd_t *r, ora_hash_ts **out, size_t *n_out) {
  uint64_t len = 0;
  get(r, &len, 8);
  if (r->bad || len > (r->n - r->off) / 16) {
    r->bad = 1;
    return 1;
  }
  ora_hash_ts *v = (ora_hash_ts *)malloc((len ? len : 1) * sizeof(ora_hash_ts));
  for (uint64_t i = 0; i < len; i++) {
    uint64_t secs;
    uint32_t nanos, hash;
    get(r, &hash, 4);
    get(r, &secs, 8);
    get(r, &nanos, 4);
    /* serde's Duration visitor re-normalises nanos >= 1e9 via Duration::new (carry into secs) */
    v[i].hash = hash;
    v[i].ts = secs * 1000000000ull + nanos;
  }
  *out = v;
  *n_out = (size_t)len;
  return r->bad;
}

int ora_frame_hashes_read(const char *path, ora_frame_hashes *fh) {
  memset(fh, 0, sizeof(*fh));
  FILE *f = fopen(path, "rb");
  if (!f) return 1;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t *buf = (uint8_t *)malloc(sz > 0 ? (size_t)sz : 1);
  size_t got = fread(buf, 1, (size_t)sz, f);
  fclose(f);
  rd_t r = {buf, got, 0, 0};
  uint32_t version = 0, tag = 0;
  get(&r, &version, 4);
  if (!r.bad && version != 0) r.bad = 1; /* unknown enum variant index: bincode error */
  get(&r, &tag, 4);
  if (!r.bad && tag != 0) r.bad = 1;
  if (!r.bad) get_vec(&r, &fh->opening, &fh->n_opening);
  if (!r.bad) get_vec(&r, &fh->ending, &fh->n_ending);
  uint64_t secs = 0, len = 0;
  uint32_t nanos = 0;
  get(&r, &secs, 8);
  get(&r, &nanos, 4);
  fh->hash_duration = secs * 1000000000ull + nanos;
  get(&r, &len, 8);
  if (!r.bad && len > r.n - r.off) r.bad = 1;
  if (!r.bad) {
    /* String of any length is valid bincode; the oracle keeps what fits (md5 is 32 hex chars) */
    size_t keep = len < 32 ? (size_t)len : 32;
    memcpy(fh->md5, r.p + r.off, keep);
    fh->md5[keep] = 0;
  }
  free(buf);
  if (r.bad) {
    ora_frame_hashes_free(fh);
    return 2;
  }
  /* is_version_valid (data.rs:96-101) can only fail if version and tag disagree; with one variant
   * each, both are 0 here. */
  return 0;
}

void ora_frame_hashes_free(ora_frame_hashes *fh) {
  free(fh->opening);
  free(fh->ending);
  fh->opening = fh->ending = NULL;
  fh->n_opening = fh->n_ending = 0;
}

/* ================================================================================================
 * comparator.rs
 * ================================================================================================ */
static int g_threads = 1;

void ora_set_threads(int n) {
  g_threads = n < 1 ? 1 : n;
  /* The per-pair table is 2898 separate 23 KiB rows (comparator.rs:175).  With glibc's default trim
   * threshold every pair returns that memory to the kernel and faults it back in, which serialises many
   * threads on the mm lock; keep freed heap mapped so the CPU baseline measures the algorithm. */
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  mallopt(M_TOP_PAD, 64 << 20);
}
int ora_get_threads(void) { return g_threads; }

/* analyzer.rs:425-455 restated at the PCM boundary: one episode per task (rayon par_iter :440-444),
 * each running the chromaprint restatement + step/timestamp rule on its opening window. */
int ora_analyze_batch(const int16_t *const *pcm, const size_t *num_values, int channels, size_t n_eps,
                      ora_ns hash_duration, ora_frame_hashes *out) {
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
  for (long e = 0; e < (long)n_eps; e++) {
    size_t samples = num_values[e] / (size_t)(channels < 1 ? 1 : channels);
    size_t n_raw = ora_chromaprint_num_items(samples);
    uint32_t *raw = (uint32_t *)malloc((n_raw ? n_raw : 1) * sizeof(uint32_t));
    ora_chromaprint_fingerprint(pcm[e], num_values[e], channels, raw, n_raw, NULL, NULL, NULL);
    ora_hash_ts *hs = (ora_hash_ts *)malloc((n_raw ? n_raw : 1) * sizeof(ora_hash_ts));
    int err = 0;
    size_t kept = ora_step_and_timestamp(raw, n_raw, hash_duration, ora_chromaprint_delay_ms(),
                                         ora_chromaprint_item_duration_ms(), 0, 0, hs, n_raw, &err);
    free(raw);
    memset(&out[e], 0, sizeof(out[e]));
    out[e].opening = hs;
    out[e].n_opening = kept;
    out[e].hash_duration = hash_duration;
    if (err) {
#pragma omp critical
      bad = 1;
    }
  }
  return bad;
}
void ora_comparator_default(ora_comparator *c) {
  c->include_endings = false;
  c->hash_match_threshold = 10;                      /* audio/mod.rs:14 */
  c->min_opening_duration = 20ull * 1000000000ull;   /* audio/mod.rs:29 */
  c->min_ending_duration = 20ull * 1000000000ull;    /* audio/mod.rs:34 */
  c->time_padding = 0;                               /* comparator.rs:91 */
}

/* #[derive(Ord)] on ComparatorHeapEntry: lexicographic in field declaration order (:22-35). */
static int cmp_u64(uint64_t a, uint64_t b) { return a < b ? -1 : a > b ? 1 : 0; }

static int entry_cmp(const ora_entry *a, const ora_entry *b) {
  int c;
  if ((c = cmp_u64(a->score, b->score))) return c;
  if ((c = cmp_u64(a->src_start, b->src_start))) return c;
  if ((c = cmp_u64(a->src_end, b->src_end))) return c;
  if ((c = cmp_u64(a->dst_start, b->dst_start))) return c;
  if ((c = cmp_u64(a->dst_end, b->dst_end))) return c;
  if ((c = cmp_u64(a->src_match_hash, b->src_match_hash))) return c;
  if ((c = cmp_u64(a->dst_match_hash, b->dst_match_hash))) return c;
  if ((c = cmp_u64(a->is_src_opening, b->is_src_opening))) return c;
  if ((c = cmp_u64(a->is_src_ending, b->is_src_ending))) return c;
  if ((c = cmp_u64(a->is_dst_opening, b->is_dst_opening))) return c;
  if ((c = cmp_u64(a->is_dst_ending, b->is_dst_ending))) return c;
  if ((c = cmp_u64(a->src_hash_duration, b->src_hash_duration))) return c;
  return cmp_u64(a->dst_hash_duration, b->dst_hash_duration);
}

typedef struct {
  ora_entry *data;
  size_t len, cap;
} heap_t;

/* std::collections::BinaryHeap::push = Vec::push + sift_up(0, old_len) (max-heap). */
static void heap_push(heap_t *h, const ora_entry *e) {
  if (h->len == h->cap) {
    h->cap = h->cap ? h->cap * 2 : 8;
    h->data = (ora_entry *)realloc(h->data, h->cap * sizeof(ora_entry));
  }
  size_t pos = h->len++;
  ora_entry elt = *e;
  while (pos > 0) {
    size_t parent = (pos - 1) / 2;
    if (entry_cmp(&elt, &h->data[parent]) <= 0) break;
    h->data[pos] = h->data[parent];
    pos = parent;
  }
  h->data[pos] = elt;
}

/* comparator.rs:149-153 */
static uint32_t compute_hash_for_match(const ora_hash_ts *hashes, size_t n, size_t start, size_t end) {
  uint32_t *tmp = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t)); /* the Rust copies ALL hashes */
  for (size_t i = 0; i < n; i++) tmp[i] = hashes[i].hash;
  uint32_t r = ora_simhash32(tmp + start, end + 1 - start);
  free(tmp);
  return r;
}

static ora_ns dur_sub(ora_ns a, ora_ns b) {
  if (a < b) abort(); /* Rust: "overflow when subtracting durations" panic */
  return a - b;
}

ora_entry *ora_longest_common_hash_match(const ora_comparator *c, const ora_hash_ts *src, size_t n,
                                         const ora_hash_ts *dst, size_t m, ora_ns src_hash_duration,
                                         ora_ns dst_hash_duration, bool is_opening, size_t *n_out) {
  *n_out = 0;
  if (n == 0 || m == 0) return NULL; /* :165-167 */
  bool is_ending = !is_opening;
  heap_t heap = {0, 0, 0};

  /* :175  vec![vec![0usize; m+1]; n+1] */
  size_t **table = (size_t **)malloc((n + 1) * sizeof(size_t *));
  for (size_t i = 0; i <= n; i++) table[i] = (size_t *)calloc(m + 1, sizeof(size_t));

  /* :176-187 */
  for (size_t i = 0; i < n; i++) {
    for (size_t j = 0; j < m; j++) {
      uint32_t sh = src[i].hash, dh = dst[j].hash;
      if (i == 0 || j == 0) {
        table[i][j] = 0;
      } else if ((uint32_t)__builtin_popcount(sh ^ dh) <= c->hash_match_threshold) {
        table[i][j] = table[i - 1][j - 1] + 1;
      } else {
        table[i][j] = 0;
      }
    }
  }

  /* :191-247  for i in (1..n).rev() for j in (1..m).rev() */
  for (size_t i = n; i-- > 1;) {
    for (size_t j = m; j-- > 1;) {
      if (table[i][j] == 0 || (i < n - 1 && j < m - 1 && table[i + 1][j + 1] != 0)) continue;
      size_t len = table[i][j];
      size_t src_start_idx = i - len, src_end_idx = i;
      size_t dst_start_idx = j - len, dst_end_idx = j;
      ora_ns src_start = src[src_start_idx].ts, src_end = src[src_end_idx].ts;
      ora_ns dst_start = dst[dst_start_idx].ts, dst_end = dst[dst_end_idx].ts;
      bool is_src_valid = (is_opening && dur_sub(src_end, src_start) >= c->min_opening_duration) ||
                          (is_ending && dur_sub(src_end, src_start) >= c->min_ending_duration);
      bool is_dst_valid = (is_opening && dur_sub(dst_end, dst_start) >= c->min_opening_duration) ||
                          (is_ending && dur_sub(dst_end, dst_start) >= c->min_ending_duration);
      if (!(is_src_valid && is_dst_valid)) continue;
      ora_entry e;
      memset(&e, 0, sizeof(e));
      e.score = len;
      e.src_start = src_start;
      e.src_end = src_end;
      e.dst_start = dst_start;
      e.dst_end = dst_end;
      e.src_match_hash = compute_hash_for_match(src, n, src_start_idx, src_end_idx);
      e.dst_match_hash = compute_hash_for_match(dst, m, dst_start_idx, dst_end_idx);
      e.is_src_opening = is_opening;
      e.is_src_ending = is_ending;
      e.is_dst_opening = is_opening;
      e.is_dst_ending = is_ending;
      e.src_hash_duration = src_hash_duration;
      e.dst_hash_duration = dst_hash_duration;
      e.src_end_idx = (uint32_t)src_end_idx;
      e.dst_end_idx = (uint32_t)dst_end_idx;
      heap_push(&heap, &e);
    }
  }
  for (size_t i = 0; i <= n; i++) free(table[i]);
  free(table);
  *n_out = heap.len; /* :249 heap.into(): the backing Vec, in heap-array order */
  return heap.data;
}

/* The same function WITHOUT the table, for checks at sizes where 2 x 8 bytes x n x m per pair is out of reach (a
 * 2000-episode library): every diagonal d = j - i is walked once with a running match length -- t[i][j] of :176-187 --
 * a run's last cell (the cells :196-200 stop at: t != 0 and, unless on the last row / column, t[i+1][j+1] == 0) is
 * put through the validity test of :212-223 on the spot, and the survivors are pushed in the order the reverse walk
 * of :191-192 meets them (i descending, then j descending).  Same entries, same BinaryHeap array as the literal
 * form (tests/test_oracle.py compares the two entry by entry). */
typedef struct {
  size_t i, j, len;
} run_end_t;

static int run_end_walk_order(const void *pa, const void *pb) {
  const run_end_t *a = (const run_end_t *)pa, *b = (const run_end_t *)pb;
  if (a->i != b->i) return a->i > b->i ? -1 : 1;
  if (a->j != b->j) return a->j > b->j ? -1 : 1;
  return 0;
}

ora_entry *ora_longest_common_hash_match_tablefree(const ora_comparator *c, const ora_hash_ts *src, size_t n,
                                                   const ora_hash_ts *dst, size_t m, ora_ns src_hash_duration,
                                                   ora_ns dst_hash_duration, bool is_opening, size_t *n_out) {
  *n_out = 0;
  if (n == 0 || m == 0) return NULL; /* :165-167 */
  const bool is_ending = !is_opening;
  const ora_ns min_dur = is_opening ? c->min_opening_duration : c->min_ending_duration;
  run_end_t *ends = NULL;
  size_t n_ends = 0, cap = 0;
  if (n >= 2 && m >= 2) {
    const long d_first = -((long)n - 2), d_last = (long)m - 2;
    for (long d = d_first; d <= d_last; d++) { /* cells with both indices >= 1 (:179-180) */
      const long lo = d < 0 ? 1 - d : 1, hi = ((long)n - 1) < ((long)m - 1 - d) ? ((long)n - 1) : ((long)m - 1 - d);
      size_t run = 0;
      for (long a = lo; a <= hi + 1; a++) {
        if (a <= hi && (uint32_t)__builtin_popcount(src[a].hash ^ dst[a + d].hash) <= c->hash_match_threshold) {
          run++;
          continue;
        }
        if (run) { /* the run's last cell is (a - 1, a - 1 + d), t = run */
          const size_t i = (size_t)(a - 1), j = (size_t)(a - 1 + d);
          if (dur_sub(src[i].ts, src[i - run].ts) >= min_dur && dur_sub(dst[j].ts, dst[j - run].ts) >= min_dur) {
            if (n_ends == cap) {
              cap = cap ? 2 * cap : 16;
              ends = (run_end_t *)realloc(ends, cap * sizeof(run_end_t));
            }
            ends[n_ends++] = (run_end_t){i, j, run};
          }
        }
        run = 0;
      }
    }
  }
  qsort(ends, n_ends, sizeof(run_end_t), run_end_walk_order);
  heap_t heap = {0, 0, 0};
  for (size_t k = 0; k < n_ends; k++) {
    const size_t i = ends[k].i, j = ends[k].j, len = ends[k].len;
    ora_entry e;
    memset(&e, 0, sizeof(e));
    e.score = len;
    e.src_start = src[i - len].ts;
    e.src_end = src[i].ts;
    e.dst_start = dst[j - len].ts;
    e.dst_end = dst[j].ts;
    e.src_match_hash = compute_hash_for_match(src, n, i - len, i);
    e.dst_match_hash = compute_hash_for_match(dst, m, j - len, j);
    e.is_src_opening = is_opening;
    e.is_src_ending = is_ending;
    e.is_dst_opening = is_opening;
    e.is_dst_ending = is_ending;
    e.src_hash_duration = src_hash_duration;
    e.dst_hash_duration = dst_hash_duration;
    e.src_end_idx = (uint32_t)i;
    e.dst_end_idx = (uint32_t)j;
    heap_push(&heap, &e);
  }
  free(ends);
  *n_out = heap.len;
  return heap.data;
}

/* OpeningAndEndingInfo, comparator.rs:47-62 */
typedef struct {
  ora_entry *src_openings, *dst_openings, *src_endings, *dst_endings;
  size_t n_src_openings, n_dst_openings, n_src_endings, n_dst_endings;
  int empty;
} info_t;

static void push_entry(ora_entry **v, size_t *n, const ora_entry *e) {
  *v = (ora_entry *)realloc(*v, (*n + 1) * sizeof(ora_entry));
  (*v)[(*n)++] = *e;
}

typedef ora_entry *(*lcs_fn)(const ora_comparator *, const ora_hash_ts *, size_t, const ora_hash_ts *, size_t, ora_ns, ora_ns,
                             bool, size_t *);

/* comparator.rs:252-308; returns 1 on FrameHashDataNoEnding */
static int find_opening_and_ending_with(lcs_fn lcs, const ora_comparator *c, const ora_frame_hashes *s,
                                        const ora_frame_hashes *d, info_t *info) {
  memset(info, 0, sizeof(*info));
  size_t n1 = 0, n2 = 0;
  ora_entry *e1 = lcs(c, s->opening, s->n_opening, d->opening, d->n_opening, s->hash_duration, d->hash_duration, true, &n1);
  ora_entry *e2 = NULL;
  if (c->include_endings) {
    if (s->n_ending == 0 || d->n_ending == 0) {
      free(e1);
      return 1;
    }
    e2 = lcs(c, s->ending, s->n_ending, d->ending, d->n_ending, s->hash_duration, d->hash_duration, false, &n2);
  }
  for (size_t k = 0; k < n1 + n2; k++) {
    const ora_entry *e = k < n1 ? &e1[k] : &e2[k - n1];
    if (e->is_src_opening)
      push_entry(&info->src_openings, &info->n_src_openings, e);
    else if (e->is_src_ending)
      push_entry(&info->src_endings, &info->n_src_endings, e);
    if (e->is_dst_opening)
      push_entry(&info->dst_openings, &info->n_dst_openings, e);
    else if (e->is_dst_ending)
      push_entry(&info->dst_endings, &info->n_dst_endings, e);
  }
  free(e1);
  free(e2);
  info->empty = !(info->n_src_openings || info->n_dst_openings || info->n_src_endings || info->n_dst_endings);
  return 0;
}

static int find_opening_and_ending(const ora_comparator *c, const ora_frame_hashes *s, const ora_frame_hashes *d,
                                   info_t *info) {
  return find_opening_and_ending_with(ora_longest_common_hash_match, c, s, d, info);
}

static void info_free(info_t *i) {
  free(i->src_openings);
  free(i->dst_openings);
  free(i->src_endings);
  free(i->dst_endings);
}

typedef struct {
  ora_ns start, end, hash_duration;
  uint32_t match_hash;
  bool is_opening;
} cand_t;

typedef struct {
  float score;
  size_t idx;
} scored_t;

static int scored_cmp(const void *pa, const void *pb) {
  /* (f32, usize)::partial_cmp, ascending; keys are unique so stability is irrelevant */
  const scored_t *a = (const scored_t *)pa, *b = (const scored_t *)pb;
  if (a->score < b->score) return -1;
  if (a->score > b->score) return 1;
  return a->idx < b->idx ? -1 : a->idx > b->idx ? 1 : 0;
}

typedef struct {
  const info_t *info;
  bool is_source;
} match_t;

/* comparator.rs:405-515; returns 2 on Duration underflow (a Rust panic) */
static int find_best_match(const ora_comparator *c, const match_t *matches, size_t n_matches,
                           ora_search_result *out) {
  memset(out, 0, sizeof(*out));
  if (n_matches == 0) return 0; /* None */
  cand_t *cand = NULL;
  size_t nc = 0;
#define PUSH(E, START, END, HD, MH, OPEN)                          \
  do {                                                             \
    cand = (cand_t *)realloc(cand, (nc + 1) * sizeof(cand_t));     \
    cand[nc].start = (E)->START;                                   \
    cand[nc].end = (E)->END;                                       \
    cand[nc].hash_duration = (E)->HD;                              \
    cand[nc].match_hash = (E)->MH;                                 \
    cand[nc].is_opening = OPEN;                                    \
    nc++;                                                          \
  } while (0)
  for (size_t k = 0; k < n_matches; k++) {
    const info_t *m = matches[k].info;
    if (matches[k].is_source) {
      for (size_t i = 0; i < m->n_src_openings; i++)
        PUSH(&m->src_openings[i], src_start, src_end, src_hash_duration, src_match_hash, true);
      for (size_t i = 0; i < m->n_src_endings; i++)
        PUSH(&m->src_endings[i], src_start, src_end, src_hash_duration, src_match_hash, false);
    } else {
      for (size_t i = 0; i < m->n_dst_openings; i++)
        PUSH(&m->dst_openings[i], dst_start, dst_end, dst_hash_duration, dst_match_hash, true);
      for (size_t i = 0; i < m->n_dst_endings; i++)
        PUSH(&m->dst_endings[i], dst_start, dst_end, dst_hash_duration, dst_match_hash, false);
    }
  }
#undef PUSH
  /* :434-454 distinct_matches: i is a key iff some j (possibly i) is within the biased threshold;
   * the relation is symmetric, so |set(i)| = #{j : dist(i,j) < bound}. */
  uint32_t bound = c->hash_match_threshold + c->hash_match_threshold / 2;
  size_t *count = (size_t *)calloc(nc ? nc : 1, sizeof(size_t));
  for (size_t i = 0; i < nc; i++)
    for (size_t j = 0; j < nc; j++) {
      uint32_t dist = (uint32_t)__builtin_popcount(cand[i].match_hash ^ cand[j].match_hash);
      if (dist >= bound) continue;
      count[i]++;
    }
  out->has_result = true;
  int rc = 0;
  for (int pass = 0; pass < 2; pass++) {
    bool want_opening = pass == 0;
    if (!want_opening && !c->include_endings) break; /* :486 */
    scored_t *best = (scored_t *)malloc((nc ? nc : 1) * sizeof(scored_t));
    size_t nb = 0;
    for (size_t k = 0; k < nc; k++) {
      if (count[k] == 0 || cand[k].is_opening != want_opening) continue;
      volatile float cnt = (float)(int64_t)count[k];
      volatile float duration_secs = ora_duration_as_secs_f32(dur_sub(cand[k].end, cand[k].start));
      volatile float a = cnt * 0.3f;
      volatile float b = duration_secs * 0.7f;
      volatile float s = a + b;
      best[nb].score = -s;
      best[nb].idx = k;
      nb++;
    }
    qsort(best, nb, sizeof(scored_t), scored_cmp);
    if (nb > 0) {
      const cand_t *w = &cand[best[0].idx];
      ora_ns start = w->start + c->time_padding;
      if (w->end < c->time_padding || w->end - c->time_padding < w->hash_duration) {
        rc = 2;
      } else {
        ora_ns end = w->end - c->time_padding - w->hash_duration;
        if (want_opening) {
          out->has_opening = true;
          out->opening_start = start;
          out->opening_end = end;
        } else {
          out->has_ending = true;
          out->ending_start = start;
          out->ending_end = end;
        }
      }
    }
    free(best);
  }
  free(count);
  free(cand);
  return rc;
}

int ora_run_with_frame_hashes(const ora_comparator *c, const ora_frame_hashes *fh, size_t nv,
                              ora_search_result *out) {
  /* :534-545 pair list (i, j), i < j, lexicographic */
  size_t np = nv * (nv ? nv - 1 : 0) / 2;
  size_t *pi = (size_t *)malloc((np ? np : 1) * sizeof(size_t));
  size_t *pj = (size_t *)malloc((np ? np : 1) * sizeof(size_t));
  bool *processed = (bool *)calloc(nv ? nv : 1, sizeof(bool));
  size_t k = 0;
  for (size_t i = 0; i < nv; i++) {
    for (size_t j = 0; j < nv; j++) {
      if (i == j || processed[j]) continue;
      pi[k] = i;
      pj[k] = j;
      k++;
    }
    processed[i] = true;
  }
  free(processed);
  info_t *infos = (info_t *)calloc(np ? np : 1, sizeof(info_t));
  int rc = 0;
  /* :553-563 rayon par_iter over pairs (one pair per task); serial when ora_set_threads(1) */
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
  for (long p = 0; p < (long)np; p++) {
    int r = find_opening_and_ending(c, &fh[pi[p]], &fh[pj[p]], &infos[p]);
    if (r) {
#pragma omp critical
      rc = r;
    }
  }
  /* :583-588 info_map, skipping empties (:562) */
  for (size_t v = 0; v < nv && rc == 0; v++) {
    match_t *matches = (match_t *)malloc((np ? np : 1) * sizeof(match_t));
    size_t nm = 0;
    for (size_t p = 0; p < np; p++) {
      if (infos[p].empty) continue;
      if (pi[p] == v) {
        matches[nm].info = &infos[p];
        matches[nm].is_source = true;
        nm++;
      } else if (pj[p] == v) {
        matches[nm].info = &infos[p];
        matches[nm].is_source = false;
        nm++;
      }
    }
    rc = find_best_match(c, matches, nm, &out[v]);
    free(matches);
  }
  for (size_t p = 0; p < np; p++) info_free(&infos[p]);
  free(infos);
  free(pi);
  free(pj);
  return rc;
}

/* comparator.rs:524-629 for SELECTED videos of a library too large for the full call: every pair that involves a
 * selected video goes through find_opening_and_ending (table-free form of the pair function), is entered in that
 * video's info_map in the global pair order (:534-545,583-588: as source if the video is the pair's first member),
 * and find_best_match gives its result.  out[k] belongs to videos[k].  Same return codes. */
int ora_run_selected_videos(const ora_comparator *c, const ora_frame_hashes *fh, size_t nv, const size_t *videos,
                            size_t n_sel, ora_search_result *out) {
  int rc = 0;
  for (size_t k = 0; k < n_sel && rc == 0; k++) {
    const size_t v = videos[k];
    /* the pairs of v in global order: (u, v) for u < v [v is the destination], then (v, u) for u > v [source] */
    info_t *infos = (info_t *)calloc(nv ? nv : 1, sizeof(info_t));
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (long u = 0; u < (long)nv; u++) {
      if ((size_t)u == v) {
        infos[u].empty = 1;
        continue;
      }
      const ora_frame_hashes *s = (size_t)u < v ? &fh[u] : &fh[v], *d = (size_t)u < v ? &fh[v] : &fh[u];
      int r = find_opening_and_ending_with(ora_longest_common_hash_match_tablefree, c, s, d, &infos[u]);
      if (r) {
#pragma omp critical
        rc = r;
      }
    }
    if (rc == 0) {
      match_t *matches = (match_t *)malloc((nv ? nv : 1) * sizeof(match_t));
      size_t nm = 0;
      for (size_t u = 0; u < nv; u++) {
        if (u == v || infos[u].empty) continue;
        matches[nm].info = &infos[u];
        matches[nm].is_source = u > v; /* pair (v, u): v is the source */
        nm++;
      }
      rc = find_best_match(c, matches, nm, &out[k]);
      free(matches);
    }
    for (size_t u = 0; u < nv; u++)
      if (u != v) info_free(&infos[u]);
    free(infos);
  }
  return rc;
}

/* ================================================================================================
 * skip file body: serde_json of SkipFile{opening: Option<(f32,f32)>, ending, md5} (data.rs:8-13);
 * f32 through ryu's shortest round-trip "pretty" format.
 * ================================================================================================ */
static size_t fmt_f32_ryu(float x, char *out) {
  if (x == 0.0f) return (size_t)sprintf(out, signbit(x) ? "-0.0" : "0.0");
  char *o = out;
  if (x < 0) {
    *o++ = '-';
    x = -x;
  }
  /* shortest digit string that parses back to x */
  char digits[16];
  int ndig = 0, exp10 = 0;
  for (int p = 1; p <= 9; p++) {
    char tmp[40];
    snprintf(tmp, sizeof(tmp), "%.*e", p - 1, (double)x);
    if (strtof(tmp, NULL) == x) {
      ndig = 0;
      for (char *q = tmp; *q && *q != 'e'; q++)
        if (*q >= '0' && *q <= '9') digits[ndig++] = *q;
      exp10 = atoi(strchr(tmp, 'e') + 1);
      break;
    }
  }
  while (ndig > 1 && digits[ndig - 1] == '0') ndig--;
  /* ryu pretty: mantissa digits D (ndig of them), value = D * 10^k, kk = ndig + k */
  int k = exp10 - (ndig - 1), kk = ndig + k;
  if (0 <= k && kk <= 13) {
    for (int i = 0; i < ndig; i++) *o++ = digits[i];
    for (int i = ndig; i < kk; i++) *o++ = '0';
    *o++ = '.';
    *o++ = '0';
  } else if (0 < kk && kk <= 13) {
    for (int i = 0; i < kk; i++) *o++ = digits[i];
    *o++ = '.';
    for (int i = kk; i < ndig; i++) *o++ = digits[i];
  } else if (-6 < kk && kk <= 0) {
    *o++ = '0';
    *o++ = '.';
    for (int i = 0; i < -kk; i++) *o++ = '0';
    for (int i = 0; i < ndig; i++) *o++ = digits[i];
  } else if (ndig == 1) {
    *o++ = digits[0];
    o += sprintf(o, "e%d", kk - 1);
  } else {
    *o++ = digits[0];
    *o++ = '.';
    for (int i = 1; i < ndig; i++) *o++ = digits[i];
    o += sprintf(o, "e%d", kk - 1);
  }
  *o = 0;
  return (size_t)(o - out);
}

size_t ora_skip_file_json(const ora_search_result *r, const char *md5, char *buf, size_t cap) {
  if (!r->has_opening && !r->has_ending) return 0; /* comparator.rs:336-338 */
  char tmp[256], a[48], b[48];
  char *o = tmp;
  o += sprintf(o, "{\"opening\":");
  if (r->has_opening) {
    fmt_f32_ryu(ora_duration_as_secs_f32(r->opening_start), a);
    fmt_f32_ryu(ora_duration_as_secs_f32(r->opening_end), b);
    o += sprintf(o, "[%s,%s]", a, b);
  } else {
    o += sprintf(o, "null");
  }
  o += sprintf(o, ",\"ending\":");
  if (r->has_ending) {
    fmt_f32_ryu(ora_duration_as_secs_f32(r->ending_start), a);
    fmt_f32_ryu(ora_duration_as_secs_f32(r->ending_end), b);
    o += sprintf(o, "[%s,%s]", a, b);
  } else {
    o += sprintf(o, "null");
  }
  o += sprintf(o, ",\"md5\":\"%s\"}", md5);
  size_t n = (size_t)(o - tmp);
  if (n + 1 > cap) return 0;
  memcpy(buf, tmp, n + 1);
  return n;
}

void ora_format_time(ora_ns t, char out[32]) {
  uint64_t secs = t / 1000000000ull;
  snprintf(out, 32, "%02llu:%02llus", (unsigned long long)(secs / 60), (unsigned long long)(secs % 60));
}

/* ---- optimised CPU variant of the scan (NOT the reference's cost structure; see ora_needle.h) ------------------ */
size_t ora_diagonal_runs_all_pairs(const uint32_t *const *hashes, const size_t *lens, size_t n_videos,
                                   uint32_t threshold, uint32_t min_len, uint32_t *runs, size_t cap) {
  const long np = n_videos < 2 ? 0 : (long)(n_videos * (n_videos - 1) / 2);
  size_t total = 0;
  if (min_len < 1) min_len = 1;
  enum { kChunks = 8 }; /* tasks = pairs x slices of the diagonals: enough of them for hundreds of threads */
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
  for (long task = 0; task < np * kChunks; task++) {
    const long p = task / kChunks, chunk = task % kChunks;
    size_t i = 0, rest = (size_t)p; /* lexicographic pair index -> (i, j), comparator.rs:534-545 */
    while (rest >= n_videos - 1 - i) {
      rest -= n_videos - 1 - i;
      i++;
    }
    const size_t j = i + 1 + rest;
    const uint32_t *s = hashes[i], *t = hashes[j];
    const long n = (long)lens[i], m = (long)lens[j];
    if (n < 2 || m < 2) continue;
    const long d_first = -(n - 2), d_count = (m - 2) - d_first + 1;
    const long d_lo = d_first + d_count * chunk / kChunks, d_hi = d_first + d_count * (chunk + 1) / kChunks;
    for (long d = d_lo; d < d_hi; d++) { /* diagonal dst - src = d; cells with both indices >= 1 */
      const long lo = d < 0 ? 1 - d : 1, hi = (n - 1) < (m - 1 - d) ? (n - 1) : (m - 1 - d);
      uint32_t run = 0;
      for (long a = lo; a <= hi + 1; a++) {
        if (a <= hi && (uint32_t)__builtin_popcount(s[a] ^ t[a + d]) <= threshold) {
          run++;
          continue;
        }
        if (run >= min_len) { /* the run ended on row a - 1 */
          size_t slot;
#pragma omp atomic capture
          slot = total++;
          if (runs && slot < cap) {
            runs[4 * slot + 0] = (uint32_t)p;
            runs[4 * slot + 1] = (uint32_t)(a - 1);
            runs[4 * slot + 2] = (uint32_t)(a - 1 + d);
            runs[4 * slot + 3] = run;
          }
        }
        run = 0;
      }
    }
  }
  return total;
}
