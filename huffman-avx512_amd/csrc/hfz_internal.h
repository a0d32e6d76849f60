// Shared between the host side (hfz_host.cpp) and the kernels (hfz_kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

#include "../../include/hfz.h"

namespace hfz {

constexpr int kMaxCodeLen = 12;  // codec/huffman.cpp:38
constexpr int kSlop = 8;         // codec/huffman.cpp:770

// Everything a frame needs besides its raw bytes, one record per frame in the
// workspace.  `enc[c]` = (code << 4) | len with the code left-aligned in 12
// bits (the packing of the reference's permute encoder,
// codec/huffman.cpp:1095-1111).  `hdr` holds header bytes 4.. of the frame
// (len_mask, per-length counts, canonical symbol order; codec/huffman.cpp:800-808).
struct FrameCode {
  uint16_t enc[256];
  uint8_t hdr[276];
  uint32_t hdr_len;
};
static_assert(sizeof(FrameCode) == 512 + 276 + 4, "FrameCode layout");

// Workspace carve-up for n frames (all offsets 256-byte aligned).
struct WorkspaceLayout {
  size_t hist;        // u32 [n][256]
  size_t code;        // FrameCode [n]
  size_t stream_end;  // u32 [n][64]  cumulative stream end offsets (incl. slop)
  size_t csize;       // u32 [n]      frame sizes
  size_t counts;      // u32 [n][256][32]  per-stream byte counts (count_kernel)
  size_t total;
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

// A batch of few frames is counted in 2^log2 pieces per stream (count_kernel's
// split): the units' histograms and counters and the first bit of every piece lie
// behind the ordinary workspace of the batch.
constexpr uint32_t kMaxSplitLog2 = 3;
struct SplitLayout {
  size_t hist_u;      // u32 [n << log2][256]
  size_t counts_u;    // u32 [n << log2][256][32]
  size_t part_first;  // u32 [n][64][8]
  size_t total;
};

inline WorkspaceLayout workspace_layout(size_t n) {
  WorkspaceLayout w;
  size_t off = 0;
  w.hist = off;
  off = align256(off + n * 256 * sizeof(uint32_t));
  w.code = off;
  off = align256(off + n * sizeof(FrameCode));
  w.stream_end = off;
  off = align256(off + n * 64 * sizeof(uint32_t));
  w.csize = off;
  off = align256(off + n * sizeof(uint32_t));
  w.counts = off;
  off = align256(off + n * 256 * 32 * sizeof(uint32_t));
  w.total = off;
  return w;
}

inline SplitLayout split_layout(size_t n, uint32_t log2) {
  SplitLayout w;
  size_t off = workspace_layout(n).total;
  const size_t units = n << log2;
  w.hist_u = off;
  off = align256(off + units * 256 * sizeof(uint32_t));
  w.counts_u = off;
  off = align256(off + units * 256 * 32 * sizeof(uint32_t));
  w.part_first = off;
  off = align256(off + n * 64 * 8 * sizeof(uint32_t));
  w.total = off;
  return w;
}

// Process-wide tuning knobs; any thread may set them while others launch, so they
// are atomics (relaxed: a launch sees the old or the new value, both are valid).
// 0 = choose by frame count, 11 / 12 = forced (hfz_set_decoder_table_bits).
extern std::atomic<int> g_decoder_table_bits;
// 0 = choose by frame count, 1 / 2 = forced (hfz_set_frames_per_wave).
extern std::atomic<int> g_frames_per_wave;
// 0 = choose by frame count, 32 / 64 = forced (hfz_set_encoder_block).
extern std::atomic<int> g_encoder_block;
// 0 = choose by batch size, n = up to n threads per stream (hfz_set_encoder_parts).
extern std::atomic<int> g_encoder_parts;
// 0 = choose by batch size, 1 / 2 = forced (hfz_set_decoder_store).
extern std::atomic<int> g_decoder_store;
// 0 = choose by batch size, n = up to n threads per stream (hfz_set_decoder_parts).
extern std::atomic<int> g_decoder_parts;
// 0 / 64 = 64-byte input blocks, 32 = 32-byte ones (hfz_set_decoder_input_block).
extern std::atomic<int> g_decoder_input_block;

// Host-side canonical code construction (hfz_host.cpp).
void build_frame_code(const uint32_t hist[256], FrameCode* out);

// The batch encoder behind hfz_compress_batch_dev and its variants (hfz_host.cpp):
// argument checks, workspace carve-up, kernel sequence.  d_shared == nullptr:
// every frame gets its own code.  Calls no allocating routine while kernel
// timing (hfz_timing_enable) is off.
int compress_batch(int K, const void* d_raw, size_t total, size_t frame_bytes, void* d_comp,
                   size_t comp_cap, size_t comp_stride, uint64_t* d_comp_off,
                   uint32_t* d_comp_size, int32_t* d_status, void* d_ws, size_t ws_bytes,
                   const FrameCode* d_shared, void* stream);
// Records the text of a failed HIP call for hfz_last_hip_error and maps it to an HFZ_E_* code.
int hip_failure(int hip_error);

// A pair of events that receive the begin and the end timestamp of ONE kernel's
// execution (either may be null).  Recording an event on the stream before and
// after a launch would time the stream's arrival at the markers instead, which
// adds the dispatch latency (about 5 us) to every kernel.
struct KernelSpan {
  hipEvent_t begin = nullptr, end = nullptr;
};

// Kernel launchers (hfz_kernels.hip).  All return a hipError_t as int.  `span`,
// where a launcher has one, times the launcher's kernel (nullptr: untimed).
int launch_histogram(const uint8_t* d_raw, uint64_t total, uint64_t frame_bytes,
                     uint64_t n_frames, uint32_t* d_hist, void* stream, bool accumulate = false,
                     const KernelSpan* span = nullptr);
int launch_build_codes(const uint32_t* d_hist, uint64_t n_frames, FrameCode* d_code, void* stream);
// d_hist != nullptr: also builds d_code[f] from d_hist[f] (separate-pass encoder);
// d_hist == nullptr: d_code is given, per frame or (shared_code) one for all frames.
int launch_stream_sizes(int K, const uint8_t* d_raw, uint64_t total, uint64_t frame_bytes,
                        uint64_t n_frames, const uint32_t* d_hist, FrameCode* d_code,
                        bool shared_code, uint32_t* d_stream_end, uint32_t* d_csize, void* stream,
                        const KernelSpan* span = nullptr);
// One pass over the input that yields code (build) and stream sizes; usable when
// count_kernel_fits(): with more than 32 streams its per-stream counters are 16 bits wide.
inline bool count_kernel_fits(int K, uint64_t frame_bytes) {
  return K <= 32 || (frame_bytes + (uint64_t)K - 1) / (uint64_t)K <= 65535;
}
// d_counts_out != nullptr: writes the per-stream counters and d_hist_out for
// launch_code_sizes; == nullptr: d_code is given and the sizes are final.
// split_log2 > 0: every stream is counted in 2^split_log2 pieces, one workgroup and one set of
// counters and histogram per piece (d_counts_out / d_hist_out hold n_frames << split_log2 units).
int launch_count(int K, const uint8_t* d_raw, uint64_t total, uint64_t frame_bytes,
                 uint64_t n_frames, const FrameCode* d_code, bool shared_code,
                 uint32_t* d_counts_out, uint32_t* d_hist_out, uint32_t* d_stream_end,
                 uint32_t* d_csize, void* stream, const KernelSpan* span = nullptr,
                 uint32_t split_log2 = 0);
// code and stream sizes of a batch counted in pieces; also the first bit of every piece
int launch_code_sizes_split(int K, const uint32_t* d_hist_u, const uint32_t* d_counts_u,
                            uint64_t n_frames, uint32_t split_log2, FrameCode* d_code,
                            uint32_t* d_part_first, uint32_t* d_stream_end, uint32_t* d_csize,
                            void* stream, const KernelSpan* span = nullptr);
// threads per stream (log2) the encoder will use on a batch; pieces hfz_workspace_bytes provides for
uint32_t encoder_parts_plan(int K, uint64_t n_frames, uint64_t frame_bytes);
uint32_t max_split_log2(int K, uint64_t n_frames);
// d_given == nullptr: builds d_code[f] from d_hist[f]; otherwise no code is built and the
// sizes are those under d_given (per frame, or one for all with shared_code).
int launch_code_sizes(int K, const uint32_t* d_hist, const uint32_t* d_counts, uint64_t n_frames,
                      FrameCode* d_code, const FrameCode* d_given, bool shared_code,
                      uint32_t* d_stream_end, uint32_t* d_csize, void* stream,
                      const KernelSpan* span = nullptr);
int launch_hist_total(const uint32_t* d_hist, uint64_t n_frames, uint64_t* d_total, void* stream,
                      const KernelSpan* span = nullptr);
int launch_frame_offsets(const uint32_t* d_csize, uint64_t n_frames, uint64_t comp_stride,
                         uint64_t comp_cap, uint64_t* d_comp_off, uint32_t* d_comp_size,
                         int32_t* d_status, void* stream, const KernelSpan* span = nullptr);
int launch_encode(int K, const uint8_t* d_raw, uint64_t total, uint64_t frame_bytes,
                  uint64_t n_frames, const FrameCode* d_code, bool shared_code,
                  const uint32_t* d_stream_end, const uint64_t* d_comp_off,
                  const uint32_t* d_comp_size, uint8_t* d_comp, uint64_t comp_cap, void* stream,
                  const KernelSpan* span = nullptr, int plan_log2_parts = -1,
                  const uint32_t* d_part_first = nullptr);
int launch_decode(int K, const uint8_t* d_comp, uint64_t comp_bytes, const uint64_t* d_comp_off,
                  const uint32_t* d_comp_size, uint64_t n_frames, bool shared_code, uint8_t* d_raw,
                  uint64_t raw_cap, uint64_t frame_bytes, int32_t* d_status, void* stream,
                  const KernelSpan* span = nullptr);
int launch_generate(const uint32_t* d_cdf, uint64_t seed, uint64_t first_index, uint8_t* d_out,
                    uint64_t n, void* stream);

}  // namespace hfz
