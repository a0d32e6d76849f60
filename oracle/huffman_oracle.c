/*
 * TEST INFRASTRUCTURE ONLY -- see huffman_oracle.h.
 *
 * Plain-C restatement of the reference's scalar K-stream Huffman path.  Each
 * function names the reference lines it follows.  K is a run-time value here
 * (a template parameter in the reference); nothing else differs.
 */
#include "huffman_oracle.h"

#include <stdlib.h>
#include <pthread.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------- */
/* small helpers                                                             */

static uint32_t rd_u32(const uint8_t* p) { /* codec/huffman.cpp:250-255 */
  uint32_t x;
  memcpy(&x, p, 4);
  return x;
}
static void wr_u32(uint8_t* p, uint32_t x) { /* codec/huffman.cpp:242-248 */
  memcpy(p, &x, 4);
}

uint64_t ho_fnv1a64_seeded(const uint8_t* p, size_t n, uint64_t basis) {
  uint64_t h = basis;
  for (size_t i = 0; i < n; ++i) {
    h ^= p[i];
    h *= 0x100000001b3ull;
  }
  return h;
}

uint64_t ho_fnv1a64(const uint8_t* p, size_t n) {
  return ho_fnv1a64_seeded(p, n, 0xcbf29ce484222325ull);
}

void ho_slice_sizes(size_t n, int K, size_t* sizes) { /* codec/huffman.cpp:98-108 */
  for (int i = 0; i < K; ++i) sizes[i] = n / (size_t)K;
  for (size_t i = 0; i < n % (size_t)K; ++i) ++sizes[i];
}

void ho_histogram(const uint8_t* p, size_t n, uint32_t hist[256]) {
  /* codec/histogram.cpp:184-191 */
  memset(hist, 0, 256 * sizeof(uint32_t));
  for (size_t i = 0; i < n; ++i) ++hist[p[i]];
}

/* ------------------------------------------------------------------------- */
/* libstdc++ std::sort, restated for uint8_t symbols ordered by count desc.   */

typedef struct {
  const uint32_t* count;
} cmp_ctx;

/* comparator of codec/huffman.cpp:354 */
static int cmp_less(const cmp_ctx* c, uint8_t a, uint8_t b) {
  return c->count[a] > c->count[b];
}

static void swap_u8(uint8_t* a, uint8_t* b) {
  uint8_t t = *a;
  *a = *b;
  *b = t;
}

static void move_median_to_first(uint8_t* result, uint8_t* a, uint8_t* b, uint8_t* c,
                                 const cmp_ctx* cx) {
  if (cmp_less(cx, *a, *b)) {
    if (cmp_less(cx, *b, *c))
      swap_u8(result, b);
    else if (cmp_less(cx, *a, *c))
      swap_u8(result, c);
    else
      swap_u8(result, a);
  } else if (cmp_less(cx, *a, *c))
    swap_u8(result, a);
  else if (cmp_less(cx, *b, *c))
    swap_u8(result, c);
  else
    swap_u8(result, b);
}

static uint8_t* unguarded_partition(uint8_t* first, uint8_t* last, uint8_t* pivot,
                                    const cmp_ctx* cx) {
  for (;;) {
    while (cmp_less(cx, *first, *pivot)) ++first;
    --last;
    while (cmp_less(cx, *pivot, *last)) --last;
    if (!(first < last)) return first;
    swap_u8(first, last);
    ++first;
  }
}

static void push_heap_(uint8_t* first, long hole, long top, uint8_t value,
                       const cmp_ctx* cx) {
  long parent = (hole - 1) / 2;
  while (hole > top && cmp_less(cx, first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

static void adjust_heap(uint8_t* first, long hole, long len, uint8_t value,
                        const cmp_ctx* cx) {
  const long top = hole;
  long child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (cmp_less(cx, first[child], first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  push_heap_(first, hole, top, value, cx);
}

static void heap_sort_range(uint8_t* first, uint8_t* last, const cmp_ctx* cx) {
  /* __partial_sort(first, last, last): make_heap, empty select loop, sort_heap */
  long len = last - first;
  if (len >= 2) {
    long parent = (len - 2) / 2;
    for (;;) {
      uint8_t v = first[parent];
      adjust_heap(first, parent, len, v, cx);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {
    --last;
    uint8_t v = *last;
    *last = *first;
    adjust_heap(first, 0, last - first, v, cx);
  }
}

static void introsort_loop(uint8_t* first, uint8_t* last, long depth_limit,
                           const cmp_ctx* cx) {
  while (last - first > 16) {
    if (depth_limit == 0) {
      heap_sort_range(first, last, cx);
      return;
    }
    --depth_limit;
    uint8_t* mid = first + (last - first) / 2;
    move_median_to_first(first, first + 1, mid, last - 1, cx);
    uint8_t* cut = unguarded_partition(first + 1, last, first, cx);
    introsort_loop(cut, last, depth_limit, cx);
    last = cut;
  }
}

static void unguarded_linear_insert(uint8_t* last, const cmp_ctx* cx) {
  uint8_t val = *last;
  uint8_t* next = last - 1;
  while (cmp_less(cx, val, *next)) {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}

static void insertion_sort(uint8_t* first, uint8_t* last, const cmp_ctx* cx) {
  if (first == last) return;
  for (uint8_t* i = first + 1; i != last; ++i) {
    if (cmp_less(cx, *i, *first)) {
      uint8_t val = *i;
      memmove(first + 1, first, (size_t)(i - first));
      *first = val;
    } else {
      unguarded_linear_insert(i, cx);
    }
  }
}

void ho_sort_syms_desc(uint8_t* syms, int n, const uint32_t count[256]) {
  cmp_ctx cx = {count};
  if (n <= 0) return;
  long lg = 0;
  for (long v = n; v > 1; v >>= 1) ++lg; /* std::__lg */
  introsort_loop(syms, syms + n, lg * 2, &cx);
  if (n > 16) {
    insertion_sort(syms, syms + 16, &cx);
    for (uint8_t* i = syms + 16; i != syms + n; ++i) unguarded_linear_insert(i, &cx);
  } else {
    insertion_sort(syms, syms + n, &cx);
  }
}

/* ------------------------------------------------------------------------- */
/* canonical, length-limited code                                            */

static void limit_code_lengths(uint16_t* len_count) { /* codec/huffman.cpp:297-327 */
  for (int i = HO_MAX_CODE_LEN + 1; i <= HO_MAX_OPTIMAL_CODE_LEN; ++i) {
    len_count[HO_MAX_CODE_LEN] += len_count[i];
    len_count[i] = 0;
  }
  uint32_t kraft_total = 0;
  for (int i = 0; i <= HO_MAX_CODE_LEN; ++i)
    kraft_total += (uint32_t)len_count[i] << (HO_MAX_CODE_LEN - i);
  const uint32_t one = 1u << HO_MAX_CODE_LEN;
  while (kraft_total > one) {
    --len_count[HO_MAX_CODE_LEN];
    for (int j = HO_MAX_CODE_LEN - 1; j >= 0; --j) {
      if (len_count[j] > 0) {
        --len_count[j];
        len_count[j + 1] += 2;
        break;
      }
    }
    --kraft_total;
  }
}

static void collect_code_len(int32_t children[256][2], int node, int len,
                             uint16_t* len_count) { /* codec/huffman.cpp:329-337 */
  if (node < 0) {
    ++len_count[len];
  } else {
    collect_code_len(children, children[node][0], len + 1, len_count);
    collect_code_len(children, children[node][1], len + 1, len_count);
  }
}

void ho_make_coding(const uint32_t hist[256], ho_coding* c) {
  /* codec/huffman.cpp:339-437 */
  memset(c, 0, sizeof(*c));
  for (int s = 0; s < 256; ++s)
    if (hist[s] != 0) c->sorted_syms[c->num_syms++] = (uint8_t)s;
  if (c->num_syms == 0) return;

  ho_sort_syms_desc(c->sorted_syms, c->num_syms, hist);

  /* two-queue Huffman tree, leaves consumed from the low-count end; on a tie
   * the leaf is taken first (:375). */
  int next_sym = c->num_syms - 1;
  uint32_t tree_count[256];
  int32_t children[256][2];
  int next_tree = 0, tree_size = 0;
  memset(tree_count, 0, sizeof(tree_count));

  uint32_t pc[2];
  int32_t pn[2];
  while ((tree_size - next_tree) + (next_sym + 1) > 1) {
    for (int t = 0; t < 2; ++t) {
      int pop_sym = 0;
      if (next_sym >= 0) {
        if (next_tree == tree_size)
          pop_sym = 1;
        else
          pop_sym = hist[c->sorted_syms[next_sym]] <= tree_count[next_tree];
      }
      if (pop_sym) {
        pc[t] = hist[c->sorted_syms[next_sym]];
        pn[t] = -1;
        --next_sym;
      } else {
        pc[t] = tree_count[next_tree];
        pn[t] = next_tree;
        ++next_tree;
      }
    }
    children[tree_size][0] = pn[0];
    children[tree_size][1] = pn[1];
    tree_count[tree_size] = pc[0] + pc[1];
    ++tree_size;
  }
  /* the root: the one remaining element */
  int root;
  if (next_sym >= 0 && (next_tree == tree_size ||
                        hist[c->sorted_syms[next_sym]] <= tree_count[next_tree]))
    root = -1;
  else
    root = next_tree;
  collect_code_len(children, root, 0, c->len_count);

  limit_code_lengths(c->len_count);

  for (int i = 0; i <= HO_MAX_CODE_LEN; ++i)
    if (c->len_count[i] != 0) c->len_mask |= 1u << i;

  /* ForallCodes, codec/huffman.cpp:260-284 */
  uint32_t cur = 0, inc = 1u << HO_MAX_CODE_LEN;
  int i = 0;
  for (int len = 0; len <= HO_MAX_CODE_LEN; ++len) {
    for (int j = 0; j < c->len_count[len]; ++j) {
      c->codes[c->sorted_syms[i]].bits = (uint16_t)cur;
      c->codes[c->sorted_syms[i]].len = (uint16_t)len;
      ++i;
      cur += inc;
    }
    inc >>= 1;
  }
}

/* ------------------------------------------------------------------------- */
/* encoder                                                                   */

typedef struct { /* codec/huffman.cpp:439-500 (CodeWriter) */
  uint8_t* output;
  uint64_t buf;
  uint64_t buf_size;
} code_writer;

static void cw_init(code_writer* w, uint8_t* end) {
  w->output = end - 8;
  w->buf = 0;
  w->buf_size = 0;
}
static void cw_write_fast(code_writer* w, ho_bitcode code) {
  w->buf |= (uint64_t)code.bits << (64 - w->buf_size - HO_MAX_CODE_LEN);
  w->buf_size += code.len;
}
static void cw_flush(code_writer* w) {
  uint32_t num_bytes = (uint32_t)(w->buf_size >> 3);
  memcpy(w->output, &w->buf, 8);
  w->output -= num_bytes;
  w->buf <<= (w->buf_size & ~7ull);
  w->buf_size &= 7;
}
static void cw_finish(code_writer* w) {
  uint8_t* out_byte = w->output + 7;
  const int bytes_left = (int)((w->buf_size + 7) / 8);
  for (int j = 0; j < bytes_left; ++j) {
    *out_byte-- = (uint8_t)(w->buf >> 56);
    w->buf <<= 8;
  }
}

static int popcount32(uint32_t x) {
  int n = 0;
  while (x) {
    x &= x - 1;
    ++n;
  }
  return n;
}

size_t ho_compress_bound(size_t n, int K) {
  /* header <= 8 + 13 + 256 + 4(K-1); every code <= 12 bits; per stream one
   * round-up byte and 8 slop bytes. */
  return 8 + 13 + 256 + 4 * (size_t)(K > 0 ? K - 1 : 0) + (n * 12 + 7) / 8 +
         9 * (size_t)(K > 0 ? K : 0);
}

size_t ho_compress_with_coding(int K, const ho_coding* coding, const uint8_t* raw,
                               size_t n, uint8_t* out, size_t cap) {
  /* codec/huffman.cpp:738-846 */
  if (K < 1 || K > HO_MAX_STREAMS) return 0;
  size_t sizes[HO_MAX_STREAMS];
  ho_slice_sizes(n, K, sizes);
  const uint8_t* part_input[HO_MAX_STREAMS];
  part_input[0] = raw;
  for (int i = 1; i < K; ++i) part_input[i] = part_input[i - 1] + sizes[i - 1];

  int32_t end_offset[HO_MAX_STREAMS];
  for (int part = 0; part < K; ++part) { /* :770-786 */
    uint32_t hist[256];
    ho_histogram(part_input[part], sizes[part], hist);
    int64_t num_bits = 0;
    for (int s = 0; s < 256; ++s) num_bits += (int64_t)hist[s] * coding->codes[s].len;
    end_offset[part] = (int32_t)((num_bits + 7) / 8 + HO_SLOP);
  }
  for (int i = 1; i < K; ++i) end_offset[i] += end_offset[i - 1];

  const size_t header_size = 4 + 4 + (size_t)popcount32(coding->len_mask) +
                             (size_t)coding->num_syms + (size_t)(K - 1) * 4; /* :794 */
  const size_t compressed_size = header_size + (size_t)end_offset[K - 1];
  if (compressed_size > cap) return 0;
  memset(out, 0, compressed_size); /* std::string::resize zero-fills, :814 */

  uint8_t* p = out;
  wr_u32(p, (uint32_t)n);
  p += 4;
  wr_u32(p, coding->len_mask);
  p += 4;
  for (int len = 0; len < 32; ++len) { /* :801-806 (256 is written as 0x00) */
    int count = coding->len_count[len];
    if (count != 0) *p++ = (uint8_t)count;
  }
  memcpy(p, coding->sorted_syms, (size_t)coding->num_syms);
  p += coding->num_syms;
  for (int k = 0; k < K - 1; ++k) {
    wr_u32(p, (uint32_t)end_offset[k]);
    p += 4;
  }

  for (int k = 0; k < K; ++k) { /* :825-843 */
    uint8_t* part_end_out = out + header_size + end_offset[k];
    code_writer w;
    cw_init(&w, part_end_out);
    const uint8_t* rp = part_input[k];
    const uint8_t* re = part_input[k] + sizes[k];
    while (rp + 3 < re) {
      for (int j = 0; j < 4; ++j) cw_write_fast(&w, coding->codes[*rp++]);
      cw_flush(&w);
    }
    while (rp != re) {
      cw_write_fast(&w, coding->codes[*rp++]);
      cw_flush(&w);
    }
    cw_finish(&w);
  }
  return compressed_size;
}

size_t ho_compress(int K, const uint8_t* raw, size_t n, uint8_t* out, size_t cap) {
  /* :755-768: per-part histograms summed == histogram of the whole input */
  uint32_t total[256];
  ho_coding coding;
  ho_histogram(raw, n, total);
  ho_make_coding(total, &coding);
  return ho_compress_with_coding(K, &coding, raw, n, out, cap);
}

/* ------------------------------------------------------------------------- */
/* decoder                                                                   */

typedef struct { /* codec/huffman.cpp:502-586 (CodeReader) */
  const uint8_t* input;
  const uint8_t* begin;
  uint64_t buf_bits;
  uint64_t bits_used;
} code_reader;

static void cr_fill(code_reader* r) { /* FillBuffer :536-556 */
  r->input -= r->bits_used >> 3;
  r->bits_used &= 7;
  if (r->input < r->begin) {
    long num_bytes_available = 8 - (long)(r->begin - r->input);
    r->buf_bits = 0;
    if (num_bytes_available > 0) {
      memcpy(&r->buf_bits, r->begin, 8);
      r->buf_bits <<= 8 * (r->begin - r->input);
    }
  } else {
    memcpy(&r->buf_bits, r->input, 8);
  }
}
static int cr_fill_fast(code_reader* r) { /* FillBufferFast :559-572 */
  r->input -= r->bits_used >> 3;
  r->bits_used &= 7;
  if (r->input < r->begin) return 0;
  memcpy(&r->buf_bits, r->input, 8);
  return 1;
}
static void cr_init(code_reader* r, const uint8_t* begin, const uint8_t* end) {
  r->begin = begin;
  r->input = end - 8;
  r->buf_bits = 0;
  r->bits_used = 0;
  cr_fill(r);
}
static uint16_t cr_code(const code_reader* r) { /* :522-527 */
  return (uint16_t)((r->buf_bits << r->bits_used) >> (64 - HO_MAX_CODE_LEN));
}

typedef struct { /* DecodedSym :588-592 */
  uint8_t code_len, sym;
} dsym1;
typedef struct { /* DecodedSym2x :634-640 */
  uint8_t num_bits, sym0, sym1, num_syms;
} dsym2;

typedef struct {
  dsym1 t1[(1 << HO_MAX_CODE_LEN) + 4];
  dsym2 t2[(1 << HO_MAX_CODE_LEN) + 4];
} decoder2x;

/* Expands ForallCodes (:260-284) into arrays; returns the number of codes. */
static int list_codes(const uint16_t* len_count, const uint8_t* syms, int num_syms,
                      uint8_t* out_sym, uint16_t* out_bits, uint8_t* out_len) {
  uint32_t cur = 0, inc = 1u << HO_MAX_CODE_LEN;
  int i = 0;
  for (int len = 0; len <= HO_MAX_CODE_LEN; ++len) {
    for (int j = 0; j < len_count[len] && i < num_syms; ++j) {
      out_sym[i] = syms[i];
      out_bits[i] = (uint16_t)cur;
      out_len[i] = (uint8_t)len;
      ++i;
      cur += inc;
    }
    inc >>= 1;
  }
  return i;
}

static void decoder2x_build(decoder2x* d, const uint16_t* len_count, const uint8_t* syms,
                            int num_syms) {
  uint8_t cs[256], cl[256];
  uint16_t cb[256];
  memset(d, 0, sizeof(*d));
  int nc = list_codes(len_count, syms, num_syms, cs, cb, cl);
  /* Decoder1x ctor :596-609 */
  for (int a = 0; a < nc; ++a) {
    uint32_t inc = 1u << (HO_MAX_CODE_LEN - cl[a]);
    for (uint32_t e = cb[a]; e < cb[a] + inc; ++e) {
      d->t1[e].code_len = cl[a];
      d->t1[e].sym = cs[a];
    }
  }
  /* Decoder2x ctor :644-681 */
  for (int a = 0; a < nc; ++a) {
    uint32_t last_code = cb[a];
    for (int b = 0; b < nc; ++b) {
      if (cl[a] + cl[b] > HO_MAX_CODE_LEN) break;
      dsym2 m = {(uint8_t)(cl[a] + cl[b]), cs[a], cs[b], 2};
      const uint32_t code = cb[a] | ((uint32_t)cb[b] >> cl[a]);
      const uint32_t inc = 1u << (HO_MAX_CODE_LEN - cl[a] - cl[b]);
      for (uint32_t e = code; e < code + inc; ++e) d->t2[e] = m;
      last_code = code + inc;
    }
    dsym2 m1 = {cl[a], cs[a], 0, 1};
    uint32_t inc1 = 1u << (HO_MAX_CODE_LEN - cl[a]);
    for (uint32_t e = last_code; e < cb[a] + inc1; ++e) d->t2[e] = m1;
  }
}

static int dec2(const decoder2x* d, uint16_t code, uint8_t** out) { /* :683-689 */
  dsym2 s = d->t2[code];
  (*out)[0] = s.sym0;
  (*out)[1] = s.sym1;
  *out += s.num_syms;
  return s.num_bits;
}
static int dec1(const decoder2x* d, uint16_t code, uint8_t** out) { /* :613-620 */
  dsym1 s = d->t1[code];
  **out = s.sym;
  ++*out;
  return s.code_len;
}

static void decode_single_stream(code_reader* r, const decoder2x* d, uint8_t* out,
                                 uint8_t* out_end) { /* :848-879 */
  int good = cr_fill_fast(r);
  const int output_slop = 4 * 2 - 1;
  while ((out + output_slop < out_end) & good) {
    for (int j = 0; j < 4; ++j) {
      int bits = dec2(d, cr_code(r), &out);
      r->bits_used += (uint64_t)bits;
    }
    good = cr_fill_fast(r);
  }
  cr_fill(r);
  while (out <= out_end - 2) {
    int bits = dec2(d, cr_code(r), &out);
    r->bits_used += (uint64_t)bits;
    cr_fill(r);
  }
  while (out < out_end) {
    int bits = dec1(d, cr_code(r), &out);
    r->bits_used += (uint64_t)bits;
    cr_fill(r);
  }
}

uint32_t ho_raw_size(const uint8_t* comp) { return rd_u32(comp); }

long ho_decompress(int K, const uint8_t* comp, size_t n, uint8_t* out, size_t cap) {
  /* ParseCompressedHeader :714-736, DecompressMultiImpl :892-955 */
  if (K < 1 || K > HO_MAX_STREAMS || n < 8) return -1;
  const uint8_t* p = comp;
  const uint8_t* comp_end = comp + n;
  uint32_t raw_size = rd_u32(p);
  p += 4;
  uint32_t len_mask = rd_u32(p);
  p += 4;
  uint16_t len_count[HO_MAX_CODE_LEN + 1];
  memset(len_count, 0, sizeof(len_count));
  int num_syms = 0;
  const int one_size = popcount32(len_mask) == 1;
  for (int i = 0; i <= HO_MAX_CODE_LEN; ++i) {
    if (len_mask & (1u << i)) {
      if (p >= comp_end) return -1;
      len_count[i] = *p;
      if (one_size && *p == 0) len_count[i] = 256;
      ++p;
      num_syms += len_count[i];
    }
  }
  if (num_syms > 256 || (size_t)(comp_end - p) < (size_t)num_syms + 4 * (size_t)(K - 1))
    return -1;
  const uint8_t* syms = p;
  p += num_syms;
  if (raw_size > cap) return -1;

  decoder2x* d = (decoder2x*)malloc(sizeof(decoder2x));
  if (!d) return -1;
  decoder2x_build(d, len_count, syms, num_syms);

  int32_t end_offset[HO_MAX_STREAMS];
  for (int k = 0; k < K - 1; ++k) {
    end_offset[k] = (int32_t)rd_u32(p);
    p += 4;
  }
  end_offset[K - 1] = (int32_t)(comp_end - p);

  size_t sizes[HO_MAX_STREAMS];
  ho_slice_sizes(raw_size, K, sizes);
  code_reader reader[HO_MAX_STREAMS];
  for (int k = 0; k < K; ++k) cr_init(&reader[k], p, p + end_offset[k]); /* :915 */

  uint8_t* part_output[HO_MAX_STREAMS];
  uint8_t* part_end[HO_MAX_STREAMS];
  part_output[0] = out;
  for (int i = 1; i < K; ++i) part_output[i] = part_output[i - 1] + sizes[i - 1];
  for (int k = 0; k < K; ++k) part_end[k] = part_output[k] + sizes[k];

  const int output_slop = 4 * 2 - 1;
  for (;;) { /* :931-948 */
    int can_continue = 1;
    for (int k = 0; k < K; ++k) {
      if (!cr_fill_fast(&reader[k]) || part_output[k] + output_slop >= part_end[k])
        can_continue = 0;
    }
    if (!can_continue) break;
    for (int j = 0; j < 4; ++j) {
      for (int k = 0; k < K; ++k) {
        int code_len = dec2(d, cr_code(&reader[k]), &part_output[k]);
        reader[k].bits_used += (uint64_t)code_len;
      }
    }
  }
  for (int k = 0; k < K; ++k)
    decode_single_stream(&reader[k], d, part_output[k], part_end[k]);
  free(d);
  return (long)raw_size;
}

/* ------------------------------------------------------------------------- */
/* synthetic source and timing helper                                        */

static uint64_t splitmix64_at(uint64_t seed, uint64_t index) {
  uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void ho_generate(const uint32_t cdf[256], uint64_t seed, uint64_t first_index,
                 uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    uint32_t u = (uint32_t)(splitmix64_at(seed, first_index + i) >> 32);
    /* number of j in [0,255) with cdf[j] <= u; cdf is non-decreasing */
    int lo = 0, hi = 255;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u)
        lo = mid + 1;
      else
        hi = mid;
    }
    out[i] = (uint8_t)lo;
  }
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* One worker of ho_time_roundtrip_mt: frames tid, tid + threads, ... */
typedef struct {
  int K, tid, threads, phase, rc;
  const uint8_t* raw;
  size_t total, frame_bytes, nframes, bound;
  uint8_t *comp, *back;
  size_t* csz;
} ho_worker;

static void* ho_worker_run(void* arg) {
  ho_worker* w = (ho_worker*)arg;
  for (size_t f = (size_t)w->tid; f < w->nframes; f += (size_t)w->threads) {
    size_t off = f * w->frame_bytes;
    size_t len = w->total - off < w->frame_bytes ? w->total - off : w->frame_bytes;
    if (w->phase == 0) {
      w->csz[f] = ho_compress(w->K, w->raw + off, len, w->comp + f * w->bound, w->bound);
    } else {
      long got = ho_decompress(w->K, w->comp + f * w->bound, w->csz[f], w->back + off, len);
      if (got != (long)len) w->rc = 1;
    }
  }
  return NULL;
}

/* As ho_time_roundtrip, with the frames of each phase spread over `threads`
 * POSIX threads (independent frames, the way a multi-core user of the
 * reference would run it); times are wall-clock per phase. */
int ho_time_roundtrip_mt(int K, const uint8_t* raw, size_t total, size_t frame_bytes, int reps,
                         int threads, double* t_c, double* t_d, size_t* comp_total) {
  if (frame_bytes == 0 || K < 1 || threads < 1 || threads > 256) return 2;
  size_t nframes = (total + frame_bytes - 1) / frame_bytes;
  size_t bound = ho_compress_bound(frame_bytes, K);
  uint8_t* comp = (uint8_t*)malloc(bound * (nframes ? nframes : 1));
  size_t* csz = (size_t*)calloc(nframes ? nframes : 1, sizeof(size_t));
  uint8_t* back = (uint8_t*)malloc(total ? total : 1);
  ho_worker* ws = (ho_worker*)calloc((size_t)threads, sizeof(ho_worker));
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  if (!comp || !csz || !back || !ws || !th) {
    free(comp); free(csz); free(back); free(ws); free(th);
    return 3;
  }
  double tc = 0, td = 0;
  int rc = 0;
  for (int r = 0; r < reps; ++r) {
    for (int phase = 0; phase < 2; ++phase) {
      double t0 = now_s();
      for (int t = 0; t < threads; ++t) {
        ho_worker w = {K, t, threads, phase, 0, raw, total, frame_bytes, nframes, bound, comp, back, csz};
        ws[t] = w;
        if (pthread_create(&th[t], NULL, ho_worker_run, &ws[t]) != 0) { ho_worker_run(&ws[t]); th[t] = 0; }
      }
      for (int t = 0; t < threads; ++t) {
        if (th[t]) pthread_join(th[t], NULL);
        if (ws[t].rc) rc = 1;
      }
      double t1 = now_s();
      if (phase == 0) tc += t1 - t0; else td += t1 - t0;
    }
  }
  size_t ctot = 0;
  for (size_t f = 0; f < nframes; ++f) ctot += csz[f];
  if (memcmp(back, raw, total) != 0) rc = 1;
  *t_c = tc;
  *t_d = td;
  *comp_total = ctot;
  free(comp); free(csz); free(back); free(ws); free(th);
  return rc;
}

int ho_time_roundtrip(int K, const uint8_t* raw, size_t total, size_t frame_bytes,
                      int reps, double* t_c, double* t_d, size_t* comp_total) {
  if (frame_bytes == 0 || K < 1) return 2;
  size_t nframes = (total + frame_bytes - 1) / frame_bytes;
  size_t bound = ho_compress_bound(frame_bytes, K);
  uint8_t* comp = (uint8_t*)malloc(bound * (nframes ? nframes : 1));
  size_t* csz = (size_t*)malloc(sizeof(size_t) * (nframes ? nframes : 1));
  uint8_t* back = (uint8_t*)malloc(total ? total : 1);
  if (!comp || !csz || !back) {
    free(comp);
    free(csz);
    free(back);
    return 3;
  }
  double tc = 0, td = 0;
  size_t ctot = 0;
  int rc = 0;
  for (int r = 0; r < reps; ++r) {
    double t0 = now_s();
    ctot = 0;
    for (size_t f = 0; f < nframes; ++f) {
      size_t off = f * frame_bytes;
      size_t len = total - off < frame_bytes ? total - off : frame_bytes;
      csz[f] = ho_compress(K, raw + off, len, comp + f * bound, bound);
      ctot += csz[f];
    }
    double t1 = now_s();
    for (size_t f = 0; f < nframes; ++f) {
      size_t off = f * frame_bytes;
      size_t len = total - off < frame_bytes ? total - off : frame_bytes;
      long got = ho_decompress(K, comp + f * bound, csz[f], back + off, len);
      if (got != (long)len) rc = 1;
    }
    double t2 = now_s();
    tc += t1 - t0;
    td += t2 - t1;
  }
  if (memcmp(back, raw, total) != 0) rc = 1;
  *t_c = tc;
  *t_d = td;
  *comp_total = ctot;
  free(comp);
  free(csz);
  free(back);
  return rc;
}
