/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the K-stream Huffman codec.
 *
 * A plain-C restatement of the scalar path of the reference codec
 * (codec/huffman.cpp, codec/histogram.cpp).  It exists so that tests and the
 * benchmark's cpu_baseline leg can check (and time) the GPU library against an
 * independent implementation.  Nothing in the product library links, loads or
 * calls this code.
 *
 * Parity pin: checked against known-answer sizes and FNV-1a-64 hashes of the
 * compiled reference's CompressMulti<K> output for nine inputs at
 * K in {1,4,32,64} (tests/golden/pins.json, tests/test_oracle_pins.py).
 */
#ifndef HUFFMAN_ORACLE_H_
#define HUFFMAN_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HO_MAX_CODE_LEN 12         /* codec/huffman.cpp:38 */
#define HO_MAX_OPTIMAL_CODE_LEN 32 /* codec/huffman.cpp:42 */
#define HO_SLOP 8                  /* codec/huffman.cpp:770 */
#define HO_MAX_STREAMS 256

/* codec/huffman.cpp:214-224 (BitCode): code left-aligned in 12 bits. */
typedef struct {
  uint16_t bits;
  uint16_t len;
} ho_bitcode;

/* codec/huffman.cpp:286-292 (CanonicalCoding). */
typedef struct {
  ho_bitcode codes[256];
  uint8_t sorted_syms[256];
  int32_t num_syms;
  uint16_t len_count[HO_MAX_OPTIMAL_CODE_LEN + 1];
  uint32_t len_mask;
} ho_coding;

/* codec/huffman.cpp:98-108 (SliceSizes<K>). sizes has K entries. */
void ho_slice_sizes(size_t n, int K, size_t* sizes);

/* codec/histogram.cpp:184-191 (MakeHistogramSimple); every variant of
 * MakeHistogram returns the same counts (codec/histogram_test.cpp:28-42). */
void ho_histogram(const uint8_t* p, size_t n, uint32_t hist[256]);

/* std::sort(syms, syms+n, [&](a,b){return count[a] > count[b];}) as libstdc++
 * executes it (introsort: median-of-3 to first, unguarded Hoare partition,
 * threshold 16, heapsort at depth 2*floor(log2 n), final insertion sort).
 * Call site: codec/huffman.cpp:353-354.  The order of equal-count symbols is
 * part of the on-wire format, so the algorithm is restated step for step. */
void ho_sort_syms_desc(uint8_t* syms, int n, const uint32_t count[256]);

/* codec/huffman.cpp:339-437 (MakeCanonicalCoding). */
void ho_make_coding(const uint32_t hist[256], ho_coding* c);

/* Upper bound on ho_compress output for n raw bytes and K streams. */
size_t ho_compress_bound(size_t n, int K);

/* codec/huffman.cpp:738-846 (CompressMulti<K>) with K a run-time value.
 * Returns the compressed size, or 0 if cap is too small / K out of range. */
size_t ho_compress(int K, const uint8_t* raw, size_t n, uint8_t* out, size_t cap);

/* Same as ho_compress but codes every frame with a caller-supplied coding
 * (shared-table mode; the reference always derives the table from the input
 * itself, so this has no reference twin -- it is checked by ho_decompress). */
size_t ho_compress_with_coding(int K, const ho_coding* coding, const uint8_t* raw,
                               size_t n, uint8_t* out, size_t cap);

/* codec/huffman.cpp:892-960 (DecompressMultiImpl<K, Decoder2x>).
 * Returns the raw size, or -1 on a header that cannot be decoded / small cap.
 * Like the reference it trusts the bit stream itself. */
long ho_decompress(int K, const uint8_t* comp, size_t n, uint8_t* out, size_t cap);

/* Raw size stored in header word 0 (codec/huffman.cpp:717). */
uint32_t ho_raw_size(const uint8_t* comp);

/* FNV-1a 64-bit.  The known-answer pins were recorded with the offset basis
 * 1469598103934665603 (not the standard 0xcbf29ce484222325), hence the seeded
 * form. */
uint64_t ho_fnv1a64(const uint8_t* p, size_t n);
uint64_t ho_fnv1a64_seeded(const uint8_t* p, size_t n, uint64_t basis);

/* Integer-only synthetic source used by the benchmark and the tests:
 * byte i = number of j in [0,255) with cdf[j] <= u, where
 * u = high 32 bits of splitmix64(seed + (i+1)*0x9E3779B97F4A7C15).
 * cdf[j] = floor(2^32 * P(X <= j)), non-decreasing, 255 used entries. */
void ho_generate(const uint32_t cdf[256], uint64_t seed, uint64_t first_index,
                 uint8_t* out, size_t n);

/* Whole-call timing helper for the cpu_baseline leg: runs ho_compress and
 * ho_decompress over `frames` frames of `frame_bytes` each, `reps` times,
 * single thread; returns seconds for compress in *t_c and decompress in *t_d.
 * Returns 0 on success, nonzero if a round trip mismatched. */
int ho_time_roundtrip_mt(int K, const uint8_t* raw, size_t total, size_t frame_bytes, int reps,
                         int threads, double* t_c, double* t_d, size_t* comp_total);
int ho_time_roundtrip(int K, const uint8_t* raw, size_t total, size_t frame_bytes,
                      int reps, double* t_c, double* t_d, size_t* comp_total);

#ifdef __cplusplus
}
#endif
#endif /* HUFFMAN_ORACLE_H_ */
