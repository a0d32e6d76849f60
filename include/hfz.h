/*
 * hfz -- K-stream static Huffman codec on AMD MI355X (gfx950), C ABI.
 *
 * Drop-in replacement for the scalar codec path of ahartik/huffman-avx512:
 * the frames this library writes are byte-identical to what the reference's
 * CompressMulti<K> (codec/huffman.cpp:738-846) writes for the same input and
 * K, and it decodes every frame the reference writes (DecompressMulti<K>,
 * codec/huffman.cpp:892-960).  K is not stored in a frame; both sides must
 * agree on it, as in the reference.
 *
 * All entry points are plain C: pointers, sizes, integer return codes.
 * Return value 0 (HFZ_OK) is success; negative values are errors (the
 * reference asserts/aborts instead, codec/huffman.cpp:715, README.md:142-146).
 *
 * The encode/decode/histogram work runs on the GPU.  There is no CPU
 * fallback: without a usable HIP device every compute entry point returns
 * HFZ_E_NO_DEVICE.
 */
#ifndef HFZ_H_
#define HFZ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  HFZ_OK = 0,
  HFZ_E_ARG = -1,        /* bad argument (null pointer, K out of range, ...) */
  HFZ_E_CAPACITY = -2,   /* destination too small */
  HFZ_E_CORRUPT = -3,    /* frame header or stream offsets are inconsistent */
  HFZ_E_NO_DEVICE = -4,  /* no HIP device / HIP runtime error at start-up */
  HFZ_E_HIP = -5,        /* a HIP call failed (hfz_last_hip_error has it) */
  HFZ_E_TOO_LARGE = -6,  /* frame larger than HFZ_MAX_FRAME_BYTES */
  HFZ_E_WORKSPACE = -7,  /* workspace smaller than hfz_workspace_bytes() */
  HFZ_E_CODE_MISMATCH = -8 /* shared-code decode: the frame's code differs from its group's */
};

/* Stream counts accepted by the GPU kernels: one wavefront lane per stream.
 * The reference instantiates K in {1,2,4,8,16,32} (scalar) and
 * {8,...,48} (AVX-512), codec/huffman.cpp:1977-2004. */
#define HFZ_MAX_STREAMS 64
/* One frame: raw_size is a u32 and stream offsets are int in the reference
 * (codec/huffman.cpp:772, 799); the per-stream bit count is formed in 32-bit
 * arithmetic there (:778), which is exact below 2^28 bytes. */
#define HFZ_MAX_FRAME_BYTES ((size_t)268435455)

const char* hfz_version(void);
const char* hfz_strerror(int code);
/* Text of the last HIP error seen by this thread ("" if none). */
const char* hfz_last_hip_error(void);
/* Number of HIP devices usable by the library (0 if none). */
int hfz_device_count(void);

/* ---- host-buffer, single-frame API ------------------------------------- */

/* Largest possible frame for n raw bytes at K streams:
 * 8 + 13 + 256 + 4(K-1) + ceil(12n/8) + 9K. */
size_t hfz_compress_bound(size_t n, int K);

/* Replaces huffman::CompressMulti<K>(raw) (codec/huffman.h:9-10,
 * codec/huffman.cpp:738-846).  src/dst are host pointers; the frame is built
 * on the current HIP device and copied back. */
int hfz_compress(int K, const void* src, size_t n, void* dst, size_t cap, size_t* out_n);

/* Replaces huffman::DecompressMulti<K>(compressed) (codec/huffman.h:11-12,
 * codec/huffman.cpp:892-960).  The raw size comes from header word 0. */
int hfz_decompress(int K, const void* src, size_t n, void* dst, size_t cap, size_t* out_n);

/* Header word 0 (codec/huffman.cpp:717, 799). */
int hfz_decompressed_size(const void* src, size_t n, size_t* raw_size);

/* Replaces huffman::MakeHistogram(str) (codec/histogram.h:12,
 * codec/histogram.cpp:193-201): 256 byte counts, computed on the GPU. */
int hfz_histogram(const void* src, size_t n, uint32_t hist[256]);

/* ---- canonical code table (host side, 256 symbols) --------------------- */

/* Everything CompressMulti derives from the total histogram
 * (MakeCanonicalCoding, codec/huffman.cpp:339-437): per-symbol code and
 * length, canonical symbol order, per-length counts. */
typedef struct hfz_code_table {
  uint16_t bits[256];      /* code, left-aligned in 12 bits (BitCode.bits) */
  uint8_t len[256];        /* code length 0..12; 0 also for absent symbols */
  uint8_t sorted_syms[256];
  uint16_t len_count[13];
  uint16_t num_syms;
  uint32_t len_mask;
} hfz_code_table;

/* Pure host function (no GPU needed).  Bit-exact with MakeCanonicalCoding,
 * including the order of equal-count symbols (std::sort call of
 * codec/huffman.cpp:353-354). */
int hfz_build_code_table(const uint32_t hist[256], hfz_code_table* out);

/* What the encoder kernels consume per frame: enc[c] = (code << 4) | len with
 * the code left-aligned in 12 bits (the packing of the reference's permute
 * encoder, codec/huffman.cpp:1095-1111) and the header bytes that follow
 * raw_size in a frame (len_mask, per-length counts, canonical symbol order;
 * codec/huffman.cpp:800-808). */
typedef struct hfz_frame_code {
  uint16_t enc[256];
  uint8_t hdr[276];
  uint32_t hdr_len;
} hfz_frame_code;

/* Host: hfz_code_table -> hfz_frame_code. */
int hfz_frame_code_from_table(const hfz_code_table* table, hfz_frame_code* out);

/* Device twin of hfz_build_code_table + hfz_frame_code_from_table, one record
 * per 256-bin histogram: d_codes[i] from d_hist[i*256 ...].  This is the
 * kernel the batch encoder runs between its histogram and encode passes. */
int hfz_build_codes_dev(const uint32_t* d_hist, size_t n, hfz_frame_code* d_codes, void* stream);

/* ---- device-resident batch API (the measured path) --------------------- */
/*
 * A batch is one contiguous raw buffer of `total` bytes cut into frames of
 * `frame_bytes` (the last one may be shorter; total == 0 gives one empty
 * frame).  Frame i is exactly CompressMulti<K>(raw[i*frame_bytes, ...)).
 * All d_* pointers are device pointers on the current device; `stream` is a
 * hipStream_t (NULL = default stream).  Calls are asynchronous on `stream`
 * unless stated otherwise.
 *
 * Alignment: d_ws on 256 bytes, d_comp_off on 8, histograms passed to
 * hfz_build_codes_dev on 16, every other u32/i32 array on 4 (hipMalloc and
 * torch allocations satisfy all of these); a misaligned pointer is refused
 * with HFZ_E_ARG.  d_raw and d_comp may have any alignment.
 *
 * Workspace: hfz_workspace_bytes(K, n_frames) is the worst case for any frame
 * size, about 35 KiB per frame whatever K is (histogram, code record, stream
 * offsets, and 32 KiB for the per-stream counters of the byte values that occur;
 * large slices and the shared-code encoder with a given code do not touch the
 * counters, but the size is checked all the same).  For small frames it exceeds
 * the raw data; allocate it once and reuse it.  A batch with too few frames to
 * fill the chip one wavefront per frame (n_frames * K below about 65 000) gets
 * room on top for counting every stream in up to eight pieces (33 KiB per piece
 * and frame: the several-threads-per-stream encoder then knows where each
 * thread's bits begin without a pass of its own over the input); a caller that
 * passes only the 35 KiB per frame gets the same bytes from the slower path.
 */
size_t hfz_batch_frames(size_t total, size_t frame_bytes);
size_t hfz_workspace_bytes(int K, size_t n_frames);

/*
 * comp_stride == 0: frames are packed back to back in d_comp in frame order.
 * comp_stride  > 0: frame i starts at i*comp_stride (must be >=
 *                   hfz_compress_bound(frame_bytes, K)).
 * d_comp_off[i] (u64) and d_comp_size[i] (u32) receive each frame's start
 * offset in d_comp and its size.  d_status[i] (may be NULL) receives HFZ_OK
 * or HFZ_E_CAPACITY per frame.
 */
int hfz_compress_batch_dev(int K, const void* d_raw, size_t total, size_t frame_bytes,
                           void* d_comp, size_t comp_cap, size_t comp_stride,
                           uint64_t* d_comp_off, uint32_t* d_comp_size, int32_t* d_status,
                           void* d_ws, size_t ws_bytes, void* stream);

/*
 * Decodes n_frames frames.  Frame i is read from d_comp + d_comp_off[i]
 * (d_comp_size[i] bytes; all frames must lie inside [d_comp, d_comp+comp_bytes))
 * and written to d_raw + i*frame_bytes; its header raw size must be
 * <= min(frame_bytes, raw_cap - i*frame_bytes).  d_status[i] (may be NULL)
 * receives HFZ_OK, HFZ_E_CORRUPT or HFZ_E_CAPACITY.
 */
int hfz_decompress_batch_dev(int K, const void* d_comp, size_t comp_bytes,
                             const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                             size_t n_frames, void* d_raw, size_t raw_cap, size_t frame_bytes,
                             int32_t* d_status, void* stream);

/*
 * One code for the whole batch ("shared table").  The reference derives every
 * frame's code from the frame itself (CompressMulti<K>, codec/huffman.cpp:755-767),
 * so this mode has no twin there; its frames are ordinary frames all the same:
 * each carries the code in its header and DecompressMulti<K> (or
 * hfz_decompress*) decodes it.  Typical use: hfz_histogram_dev over the batch
 * (sum the 256 counters across GPUs on the host), hfz_build_code_table +
 * hfz_frame_code_from_table, copy the record to the device, pass it as d_code.
 * Every byte value that occurs in the batch must have a code (a nonzero count in
 * the histogram the code was built from); bytes without one are coded as garbage.
 * Other arguments as hfz_compress_batch_dev.
 */
int hfz_compress_batch_shared_dev(int K, const void* d_raw, size_t total, size_t frame_bytes,
                                  const hfz_frame_code* d_code, void* d_comp, size_t comp_cap,
                                  size_t comp_stride, uint64_t* d_comp_off, uint32_t* d_comp_size,
                                  int32_t* d_status, void* d_ws, size_t ws_bytes, void* stream);

/*
 * The same in two steps, so that the input is read twice instead of three
 * times (histogram, sizes, encode): hfz_count_batch_dev makes one pass that
 * leaves per-stream byte counts in the workspace and the batch's 256 byte
 * counts (64-bit) in d_hist_total; after the code has been built from those
 * (summed over GPUs where the job has several), hfz_compress_batch_counted_dev
 * turns the counts into stream sizes and encodes.  Both calls must see the same
 * K, d_raw, total, frame_bytes and workspace, with no other call on that
 * workspace in between.  No counterpart in the reference.
 */
int hfz_count_batch_dev(int K, const void* d_raw, size_t total, size_t frame_bytes, void* d_ws,
                        size_t ws_bytes, uint64_t* d_hist_total, void* stream);
int hfz_compress_batch_counted_dev(int K, const void* d_raw, size_t total, size_t frame_bytes,
                                   const hfz_frame_code* d_code, void* d_comp, size_t comp_cap,
                                   size_t comp_stride, uint64_t* d_comp_off, uint32_t* d_comp_size,
                                   int32_t* d_status, void* d_ws, size_t ws_bytes, void* stream);

/*
 * Decoder for batches whose frames share one code.  With frames of up to
 * 128 KiB, groups of three consecutive frames build one lookup table (from the
 * first frame of the group) instead of three; a frame whose code section
 * differs from its group's first frame then gets HFZ_E_CODE_MISMATCH in d_status
 * and is left undecoded.  Larger frames take the same path as
 * hfz_decompress_batch_dev (a table per frame costs nothing measurable there),
 * which decodes any batch.  Arguments as hfz_decompress_batch_dev.
 */
int hfz_decompress_batch_shared_dev(int K, const void* d_comp, size_t comp_bytes,
                                    const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                                    size_t n_frames, void* d_raw, size_t raw_cap,
                                    size_t frame_bytes, int32_t* d_status, void* stream);

/*
 * hfz_compress_batch_dev without a workspace argument and without any allocation.
 * No counterpart in the reference; it exists for callers that cannot allocate:
 * code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  The workspace is a device array inside the library, sized for at
 * most HFZ_STATIC_MAX_FRAMES frames per call (about 35 KiB per frame, 9 MiB per
 * GPU, held by every GPU the library runs on until the process ends); a batch of
 * more frames is refused with HFZ_E_TOO_LARGE.  The call makes no hipMalloc,
 * hipFree or other allocating call as long as kernel timing (hfz_timing_enable)
 * is off.  Output is identical to hfz_compress_batch_dev's.  All calls on one
 * GPU share that GPU's array: calls that overlap in time on one GPU (different
 * streams or threads) race; serialise them.  hfz_static_workspace_init resolves
 * the array's address on the current device (the entry point does the same on
 * its first call on a device); call it once per device before a graph capture.
 */
#define HFZ_STATIC_MAX_FRAMES 256
int hfz_static_workspace_init(void);
size_t hfz_static_max_frames(void);
int hfz_compress_batch_static_dev(int K, const void* d_raw, size_t total, size_t frame_bytes,
                                  void* d_comp, size_t comp_cap, size_t comp_stride,
                                  uint64_t* d_comp_off, uint32_t* d_comp_size, int32_t* d_status,
                                  void* stream);

/* 256 byte counts (u32, wraps at 2^32) of a device buffer of any size into
 * d_hist[256] (buffers beyond 1 GiB are counted in pieces on the same stream). */
int hfz_histogram_dev(const void* d_src, size_t n, uint32_t* d_hist, void* stream);

/* Per-frame histograms: d_hist[i*256 + c] for the batch's frames. */
int hfz_histogram_batch_dev(const void* d_raw, size_t total, size_t frame_bytes,
                            uint32_t* d_hist, void* stream);

/*
 * Optional per-kernel timing of the batch entry points (used by the benchmark
 * to price each kernel against its roofline).  hfz_timing_enable(n), n >= 1,
 * makes every n-th of the following batch calls of each kind on this thread
 * (the first included) launch each of its kernels with a pair of HIP events that
 * receive the kernel's own begin and end timestamps (hipExtLaunchKernelGGL), so
 * a kernel's time is its execution alone, without the dispatch latency that an
 * event recorded on the stream before the launch would add (at most 512
 * recorded calls of each kind between reads); 0 turns it off.  A timed launch
 * stops one kernel's tail from overlapping the next kernel's start, about 5 %
 * of a 256 MiB batch, so a benchmark samples (n = 4) rather than timing every call.
 * hfz_timing_read waits for the events, returns for each kernel the mean
 * milliseconds over the recorded calls that launched it since the previous read
 * (0 for a kernel none of them launched), and starts a new window.  hfz_timing_enable(0) also destroys the events.  Has no
 * counterpart in the reference.
 */
typedef struct hfz_batch_timing {
  float hist_ms;    /* input pass: per-stream byte counts (or the plain histogram, K1) */
  float table_ms;   /* canonical code construction (+ stream sizes from the counts) */
  float sizes_ms;   /* separate code-length pass (K2); 0 when the counts gave the sizes */
  float place_ms;   /* frame placement (prefix sum) */
  float encode_ms;  /* encode (K3) */
  float decode_ms;  /* table build + decode (K4+K5) */
  int32_t compress_calls;   /* calls averaged into the first five fields */
  int32_t decompress_calls; /* calls averaged into decode_ms */
  float count_ms;           /* input pass of hfz_count_batch_dev (the two-step shared-code encoder) */
  int32_t count_calls;      /* calls averaged into count_ms */
} hfz_batch_timing;
int hfz_timing_enable(int every);
int hfz_timing_read(hfz_batch_timing* out);

/*
 * The decoder has two builds of its lookup table (the reference has one,
 * Decoder2x, codec/huffman.cpp:634-704): 12 index bits (16 KiB of LDS per
 * frame, every code resolved by the index) and 11 index bits (8 KiB, more frames resident per
 * CU).  bits = 0 (default) picks by the number of frames in the call; 11 or 12
 * forces one, for tests and tuning.  HFZ_DECODER_TABLE_FREE selects a third
 * build that uses no lookup table at all: code lengths by comparing the window
 * with the first code of each length, symbols by indexing the canonical symbol
 * list, one symbol per step (the method of the reference's permute decoder,
 * codec/huffman.cpp:1697-1756, 1892-1940); it is never chosen automatically (it
 * is 3-4 times slower than the table builds on this machine) and exists for
 * comparison.  Decoded bytes are identical.  Process-wide.
 */
#define HFZ_DECODER_TABLE_FREE (-1)
int hfz_set_decoder_table_bits(int bits);

/*
 * With K <= 32 one frame fills at most half a wavefront, so the encoder and the
 * decoder can give each half of a wave a frame of its own.  n = 0 (default)
 * does so when the batch has enough frames to keep the chip busy either way,
 * 1 never, 2 always (K <= 32 only).  Output is identical.  Process-wide; has no
 * counterpart in the reference.
 */
int hfz_set_frames_per_wave(int n);

/*
 * The encoder writes its streams in aligned blocks of 32 or 64 bytes; bytes = 0
 * (default) takes 64 once the batch has more frames than SIMDs, 32 below.
 * 32 / 64 force one, for tests and tuning.  Output is identical.  Process-wide;
 * no counterpart in the reference.
 */
int hfz_set_encoder_block(int bytes);

/*
 * Threads per stream in the encoder.  The reference encodes stream after
 * stream, one symbol at a time (codec/huffman.cpp:825-843); one lane per
 * stream leaves most of the chip idle when a batch has few frames (a single
 * frame is one wavefront).  With n > 1 every stream is cut into n parts (a
 * power of two, at most 512 / K and 64) that are encoded by n threads at once
 * into the same bytes.  n = 0 (default) picks by batch size: as many parts as
 * it takes to occupy the chip, none for large batches; 1 = never; other values
 * are rounded down to what K allows.  Output is identical.  Process-wide; no
 * counterpart in the reference.
 */
int hfz_set_encoder_parts(int n);

/*
 * Threads per stream in the decoder.  The reference hides the latency of a
 * stream's chain of lookups behind the K streams of a frame
 * (codec/huffman.cpp:931-948); one lane per stream gives a frame one wavefront
 * whatever its size, so a batch of few frames (a single frame above all) leaves
 * most of the chip idle.  With n > 1 every stream is decoded by n threads (a
 * power of two, at most 512 / K and 64): the stream's bits are cut into n pieces,
 * every thread finds a code boundary near the start of its piece by decoding
 * from a little before it (a Huffman code falls into step by itself), neighbours
 * check that each piece starts where the one before it ended and repeat from
 * there if not, a prefix sum over the symbol counts places the pieces, and then
 * each is decoded like a stream of its own.  n = 0 (default) picks by batch size:
 * as many as it takes to occupy the chip, none for large batches; 1 = never;
 * other values are rounded down to what K allows.  Decoded bytes are identical.
 * Process-wide; no counterpart in the reference.
 */
int hfz_set_decoder_parts(int n);

/*
 * How the one-frame-per-wave decoder with the 11-bit table writes its output:
 * mode = 1, every lane stores the blocks of its own stream, two 32-byte blocks
 * at a time (best for long streams); mode = 2, neighbouring lanes store each
 * other's 32-byte blocks together, 16 bytes each (best for streams of up to
 * 2 KiB, e.g. 128 KiB frames of 64 streams); 0 (default) picks by frame size.  Decoded bytes are identical.  Process-wide; no counterpart in the
 * reference.
 */
int hfz_set_decoder_store(int mode);

/*
 * Bytes per block in which the 11-bit-table decoder fetches a stream: 64 (a
 * 128-byte input ring per lane; default, also for bytes = 0) or 32 (a 64-byte
 * ring: 4 KiB less LDS per wave, so more frames resident per CU, and twice as
 * many staging steps; measured slower, kept for comparison).  Decoded bytes are
 * identical.  Process-wide; no counterpart in the reference.
 */
int hfz_set_decoder_input_block(int bytes);

/*
 * Deterministic integer-only synthetic source, filled on the device:
 * byte i = #{ j in [0,255) : cdf[j] <= u_i },  u_i = high 32 bits of
 * splitmix64(seed + (first_index + i + 1) * 0x9E3779B97F4A7C15).
 * d_cdf points to 256 u32 on the device.
 */
int hfz_generate_dev(const uint32_t* d_cdf, uint64_t seed, uint64_t first_index, void* d_out,
                     size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HFZ_H_ */
