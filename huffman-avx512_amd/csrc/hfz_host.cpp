// Host side of libhfz: canonical code construction, frame header bytes, the
// C ABI entry points and HIP plumbing.  No compute happens here apart from the
// 256-symbol table construction, which the reference also does once per frame
// on the CPU (codec/huffman.cpp:339-437).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "hfz_internal.h"

namespace hfz {
namespace {

thread_local std::string g_last_hip_error;

int hip_fail(hipError_t e) {
  g_last_hip_error = hipGetErrorString(e);
  return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? HFZ_E_NO_DEVICE : HFZ_E_HIP;
}
#define HFZ_HIP(call)                            \
  do {                                           \
    hipError_t e__ = (hipError_t)(call);         \
    if (e__ != hipSuccess) return hip_fail(e__); \
  } while (0)

// Per-thread event sets for hfz_timing_*: while timing is on, every `every`-th
// batch call of a kind (the first included) takes a fresh set, and each of its
// kernels is launched with a pair of events that receive the kernel's own begin
// and end timestamps (KernelSpan); hfz_timing_read averages over all sets taken
// since the last read.  Timed launches keep one kernel's tail from overlapping
// the next one's start, which costs a few percent of a short batch: sampling
// keeps the measurement and most of the overlap.
constexpr int kEncSpans = 5;  // hist, table, sizes, place, encode
constexpr int kEncEvents = 2 * kEncSpans, kDecEvents = 2, kCntEvents = 2, kMaxTimedCalls = 512;
struct Timing {
  int every = 0;                          // 0 = off
  std::vector<hipEvent_t> enc, dec, cnt;  // kEncEvents / kDecEvents / kCntEvents per timed call
  std::vector<uint8_t> enc_used;          // per timed compress call: bit i = span i was launched
  size_t enc_calls = 0, dec_calls = 0, cnt_calls = 0;  // timed calls
  size_t enc_seen = 0, dec_seen = 0, cnt_seen = 0;     // all calls while on
  hipEvent_t* cur = nullptr;
  KernelSpan span_buf;
  void reset_window() {
    enc_calls = dec_calls = cnt_calls = 0;
    enc_seen = dec_seen = cnt_seen = 0;
    cur = nullptr;
  }
  // Called when timing is switched off: the events go back to the runtime (a
  // destructor at thread or process exit could run after HIP has shut down).
  void release() {
    for (auto* v : {&enc, &dec, &cnt}) {
      for (hipEvent_t e : *v) (void)hipEventDestroy(e);
      v->clear();
    }
    enc_used.clear();
  }
  // n_frames == 0: the call launches nothing, so there is nothing to time
  hipError_t take(std::vector<hipEvent_t>& v, size_t& calls, size_t& seen, int per_call,
                  size_t n_frames) {
    cur = nullptr;
    if (every <= 0 || n_frames == 0) return hipSuccess;
    if (seen++ % (size_t)every != 0 || calls >= (size_t)kMaxTimedCalls) return hipSuccess;
    while (v.size() < (calls + 1) * per_call) {
      hipEvent_t e;
      hipError_t r = hipEventCreate(&e);
      if (r != hipSuccess) return r;
      v.push_back(e);
    }
    cur = &v[calls * per_call];
    ++calls;
    if (&v == &enc) {
      enc_used.resize(calls);
      enc_used[calls - 1] = 0;
    }
    return hipSuccess;
  }
  // The span for kernel i of the call being timed (nullptr: the call is not timed).
  // The pointer is good until the next call of span().
  const KernelSpan* span(int i) {
    if (!cur) return nullptr;
    span_buf.begin = cur[2 * i];
    span_buf.end = cur[2 * i + 1];
    return &span_buf;
  }
  const KernelSpan* enc_span(int i) {
    if (!cur) return nullptr;
    enc_used[enc_calls - 1] |= (uint8_t)(1u << i);
    return span(i);
  }
};
thread_local Timing g_timing;

// The kernels read histograms and counters with 16-byte vector loads and carve
// the workspace on 256-byte boundaries.
bool misaligned(const void* p, uintptr_t a) { return ((uintptr_t)p & (a - 1)) != 0; }

bool have_device() {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

// LimitCodeLengths, codec/huffman.cpp:297-327 (MiniZ method).  Lengths above
// 12 are folded into 12, then codes are lengthened until the Kraft sum is 1.
void limit_code_lengths(uint32_t* len_count, int max_seen) {
  for (int i = kMaxCodeLen + 1; i <= max_seen; ++i) {
    len_count[kMaxCodeLen] += len_count[i];
    len_count[i] = 0;
  }
  uint32_t kraft = 0;
  for (int i = 0; i <= kMaxCodeLen; ++i) kraft += len_count[i] << (kMaxCodeLen - i);
  const uint32_t one = 1u << kMaxCodeLen;
  while (kraft > one) {
    --len_count[kMaxCodeLen];
    for (int j = kMaxCodeLen - 1; j >= 0; --j) {
      if (len_count[j] > 0) {
        --len_count[j];
        len_count[j + 1] += 2;
        break;
      }
    }
    --kraft;
  }
}

}  // namespace

int hip_failure(int hip_error) { return hip_fail((hipError_t)hip_error); }

// MakeCanonicalCoding, codec/huffman.cpp:339-437.  The symbol order comes from
// the same std::sort call on the same initial sequence with the same
// comparator as the reference (:342-354), so equal counts fall the same way.
static void make_code_table(const uint32_t hist[256], hfz_code_table* t) {
  std::memset(t, 0, sizeof(*t));
  int n = 0;
  for (int c = 0; c < 256; ++c)
    if (hist[c] != 0) t->sorted_syms[n++] = (uint8_t)c;
  t->num_syms = (uint16_t)n;
  if (n == 0) return;
  std::sort(t->sorted_syms, t->sorted_syms + n,
            [&](uint8_t a, uint8_t b) { return hist[a] > hist[b]; });

  // Two-queue Huffman construction (:364-416).  Queue 1: leaves, read from the
  // low-count end of sorted_syms; queue 2: internal nodes in creation order.
  // A leaf wins ties (:375).  Only depths are needed, so each node records its
  // parent and depths are resolved from the root down afterwards.
  uint32_t tree_count[256];
  int leaf_parent[256], node_parent[256];
  int next_sym = n - 1, next_node = 0, tree_size = 0;
  auto pop = [&](int parent, uint32_t* count) {
    bool leaf = false;
    if (next_sym >= 0)
      leaf = next_node == tree_size || hist[t->sorted_syms[next_sym]] <= tree_count[next_node];
    if (leaf) {
      *count = hist[t->sorted_syms[next_sym]];
      leaf_parent[next_sym--] = parent;
    } else {
      *count = tree_count[next_node];
      node_parent[next_node++] = parent;
    }
  };
  while ((tree_size - next_node) + (next_sym + 1) > 1) {
    uint32_t a, b;
    pop(tree_size, &a);
    pop(tree_size, &b);
    tree_count[tree_size++] = a + b;
  }
  uint32_t len_count[256] = {};
  int max_seen = 0;
  if (tree_size == 0) {
    len_count[0] = 1;  // a single symbol gets a zero-length code
  } else {
    int depth[256];
    depth[tree_size - 1] = 0;  // the root is the last node created
    for (int i = tree_size - 2; i >= 0; --i) depth[i] = depth[node_parent[i]] + 1;
    for (int i = 0; i < n; ++i) {
      int d = depth[leaf_parent[i]] + 1;
      ++len_count[d];
      max_seen = std::max(max_seen, d);
    }
  }
  limit_code_lengths(len_count, max_seen);

  uint32_t cur = 0, inc = 1u << kMaxCodeLen;  // ForallCodes, :260-284
  int i = 0;
  for (int len = 0; len <= kMaxCodeLen; ++len) {
    t->len_count[len] = (uint16_t)len_count[len];
    if (len_count[len]) t->len_mask |= 1u << len;
    for (uint32_t j = 0; j < len_count[len]; ++j) {
      const uint8_t s = t->sorted_syms[i++];
      t->bits[s] = (uint16_t)cur;
      t->len[s] = (uint8_t)len;
      cur += inc;
    }
    inc >>= 1;
  }
}

static void frame_code_from_table(const hfz_code_table& t, FrameCode* out) {
  for (int c = 0; c < 256; ++c) out->enc[c] = (uint16_t)((t.bits[c] << 4) | t.len[c]);
  // header bytes after raw_size: len_mask, counts (256 -> 0x00), symbols (:800-808)
  uint8_t* p = out->hdr;
  std::memcpy(p, &t.len_mask, 4);
  p += 4;
  for (int len = 0; len <= kMaxCodeLen; ++len)
    if (t.len_count[len]) *p++ = (uint8_t)t.len_count[len];
  std::memcpy(p, t.sorted_syms, t.num_syms);
  p += t.num_syms;
  out->hdr_len = (uint32_t)(p - out->hdr);
  std::memset(p, 0, sizeof(out->hdr) - out->hdr_len);
}

void build_frame_code(const uint32_t hist[256], FrameCode* out) {
  hfz_code_table t;
  make_code_table(hist, &t);
  frame_code_from_table(t, out);
}

}  // namespace hfz

using namespace hfz;

extern "C" {

const char* hfz_version(void) { return "hfz 0.1 (gfx950)"; }

const char* hfz_strerror(int code) {
  switch (code) {
    case HFZ_OK: return "ok";
    case HFZ_E_ARG: return "bad argument";
    case HFZ_E_CAPACITY: return "destination too small";
    case HFZ_E_CORRUPT: return "corrupt frame";
    case HFZ_E_NO_DEVICE: return "no HIP device";
    case HFZ_E_HIP: return "HIP call failed";
    case HFZ_E_TOO_LARGE: return "frame too large";
    case HFZ_E_WORKSPACE: return "workspace too small";
    case HFZ_E_CODE_MISMATCH: return "frame does not carry the shared code";
    default: return "unknown error";
  }
}

const char* hfz_last_hip_error(void) { return g_last_hip_error.c_str(); }

int hfz_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

size_t hfz_compress_bound(size_t n, int K) {
  if (K < 1) K = 1;
  return 8 + 13 + 256 + 4 * (size_t)(K - 1) + (n * 12 + 7) / 8 + 9 * (size_t)K;
}

int hfz_build_code_table(const uint32_t hist[256], hfz_code_table* out) {
  if (!hist || !out) return HFZ_E_ARG;
  make_code_table(hist, out);
  return HFZ_OK;
}

size_t hfz_batch_frames(size_t total, size_t frame_bytes) {
  if (frame_bytes == 0) return 0;
  if (total == 0) return 1;
  return (total + frame_bytes - 1) / frame_bytes;
}

size_t hfz_workspace_bytes(int K, size_t n_frames) {
  // a batch of few frames is counted in pieces (more workgroups, and the encoder's
  // parts learn their first bits from it): room for the most pieces it can take
  const uint32_t split = (K >= 1 && K <= HFZ_MAX_STREAMS) ? max_split_log2(K, n_frames) : 0;
  return split ? split_layout(n_frames, split).total : workspace_layout(n_frames).total;
}

int hfz_histogram_batch_dev(const void* d_raw, size_t total, size_t frame_bytes, uint32_t* d_hist,
                            void* stream) {
  if (!d_hist || frame_bytes == 0 || (total && !d_raw) || misaligned(d_hist, 4)) return HFZ_E_ARG;
  if (frame_bytes > HFZ_MAX_FRAME_BYTES) return HFZ_E_TOO_LARGE;
  const size_t nf = hfz_batch_frames(total, frame_bytes);
  HFZ_HIP(launch_histogram((const uint8_t*)d_raw, total, frame_bytes, nf, d_hist, stream));
  return HFZ_OK;
}

int hfz_histogram_dev(const void* d_src, size_t n, uint32_t* d_hist, void* stream) {
  if (!d_hist || (n && !d_src)) return HFZ_E_ARG;
  if (((uintptr_t)d_hist & 3) != 0) return HFZ_E_ARG;
  // Counts wrap at 2^32 like the u32 bins of ByteHistogram (codec/histogram.h:10).
  // The kernel's frame length is 32 bits wide, so a large buffer is counted in
  // pieces of at most 1 GiB, each added to the same 256 counters.
  constexpr size_t kPiece = size_t(1) << 30;
  if (n <= kPiece) {
    HFZ_HIP(launch_histogram((const uint8_t*)d_src, n, n ? n : 1, 1, d_hist, stream));
    return HFZ_OK;
  }
  HFZ_HIP(hipMemsetAsync(d_hist, 0, 256 * sizeof(uint32_t), (hipStream_t)stream));
  for (size_t lo = 0; lo < n; lo += kPiece) {
    const size_t len = n - lo < kPiece ? n - lo : kPiece;
    HFZ_HIP(launch_histogram((const uint8_t*)d_src + lo, len, len, 1, d_hist, stream, true));
  }
  return HFZ_OK;
}

}  // extern "C" (reopened below)

namespace hfz {

// d_shared == nullptr: every frame gets its own code (the reference's
// CompressMulti<K>); otherwise all frames are coded with *d_shared.
int compress_batch(int K, const void* d_raw, size_t total, size_t frame_bytes, void* d_comp,
                   size_t comp_cap, size_t comp_stride, uint64_t* d_comp_off,
                   uint32_t* d_comp_size, int32_t* d_status, void* d_ws, size_t ws_bytes,
                   const FrameCode* d_shared, void* stream) {
  if (K < 1 || K > HFZ_MAX_STREAMS || frame_bytes == 0 || !d_comp || !d_comp_off ||
      !d_comp_size || !d_ws || (total && !d_raw))
    return HFZ_E_ARG;
  if (frame_bytes > HFZ_MAX_FRAME_BYTES) return HFZ_E_TOO_LARGE;
  if (comp_stride && comp_stride < hfz_compress_bound(frame_bytes, K)) return HFZ_E_ARG;
  if (misaligned(d_ws, 256) || misaligned(d_comp_off, 8) || misaligned(d_comp_size, 4) ||
      misaligned(d_status, 4) || misaligned(d_shared, 4))
    return HFZ_E_ARG;
  const size_t nf = hfz_batch_frames(total, frame_bytes);
  const WorkspaceLayout w = workspace_layout(nf);
  if (ws_bytes < w.total) return HFZ_E_WORKSPACE;
  uint8_t* ws = (uint8_t*)d_ws;
  uint32_t* d_hist = (uint32_t*)(ws + w.hist);
  FrameCode* d_code = (FrameCode*)(ws + w.code);
  uint32_t* d_send = (uint32_t*)(ws + w.stream_end);
  uint32_t* d_csize = (uint32_t*)(ws + w.csize);
  const uint8_t* raw = (const uint8_t*)d_raw;
  const bool shared = d_shared != nullptr;
  const FrameCode* enc_code = shared ? d_shared : d_code;

  HFZ_HIP(g_timing.take(g_timing.enc, g_timing.enc_calls, g_timing.enc_seen, kEncEvents, nf));
  // how many threads the encoder will put on a stream: asked once, here, so that
  // counting and encoding agree whatever happens to the knobs in between
  const uint32_t enc_parts = encoder_parts_plan(K, nf, frame_bytes);
  const uint32_t* d_part_first = nullptr;
  if (count_kernel_fits(K, frame_bytes)) {
    // one pass over the input for the per-stream counts (hist_ms), then code and
    // stream sizes from those (table_ms); sizes_ms reads 0.  With a shared code
    // the counting kernel turns the counts into sizes itself.
    const bool split = !shared && enc_parts >= 1 && enc_parts <= kMaxSplitLog2 &&
                       ws_bytes >= split_layout(nf, enc_parts).total;
    if (split) {
      // few frames: every stream is counted in the encoder's parts, one workgroup per
      // part, and the parts' first bits come out of the size computation
      const SplitLayout sl = split_layout(nf, enc_parts);
      uint32_t* d_pf = (uint32_t*)(ws + sl.part_first);
      HFZ_HIP(launch_count(K, raw, total, frame_bytes, nf, nullptr, false,
                           (uint32_t*)(ws + sl.counts_u), (uint32_t*)(ws + sl.hist_u), d_send,
                           d_csize, stream, g_timing.enc_span(0), enc_parts));
      HFZ_HIP(launch_code_sizes_split(K, (const uint32_t*)(ws + sl.hist_u),
                                      (const uint32_t*)(ws + sl.counts_u), nf, enc_parts, d_code,
                                      d_pf, d_send, d_csize, stream, g_timing.enc_span(1)));
      d_part_first = d_pf;
    } else {
      uint32_t* d_counts = (uint32_t*)(ws + w.counts);
      HFZ_HIP(launch_count(K, raw, total, frame_bytes, nf, shared ? d_shared : nullptr, shared,
                           shared ? nullptr : d_counts, shared ? nullptr : d_hist, d_send, d_csize,
                           stream, g_timing.enc_span(0)));
      if (!shared)
        HFZ_HIP(launch_code_sizes(K, d_hist, d_counts, nf, d_code, nullptr, false, d_send, d_csize,
                                  stream, g_timing.enc_span(1)));
    }
  } else {
    // the code is built inside the sizes kernel (table_ms reads 0)
    if (!shared)
      HFZ_HIP(launch_histogram(raw, total, frame_bytes, nf, d_hist, stream, false,
                               g_timing.enc_span(0)));
    HFZ_HIP(launch_stream_sizes(K, raw, total, frame_bytes, nf, shared ? nullptr : d_hist,
                                shared ? const_cast<FrameCode*>(d_shared) : d_code, shared, d_send,
                                d_csize, stream, g_timing.enc_span(2)));
  }
  HFZ_HIP(launch_frame_offsets(d_csize, nf, comp_stride, comp_cap, d_comp_off, d_comp_size,
                               d_status, stream, g_timing.enc_span(3)));
  HFZ_HIP(launch_encode(K, raw, total, frame_bytes, nf, enc_code, shared, d_send, d_comp_off,
                        d_comp_size, (uint8_t*)d_comp, comp_cap, stream, g_timing.enc_span(4),
                        (int)enc_parts, d_part_first));
  g_timing.cur = nullptr;
  return HFZ_OK;
}

}  // namespace hfz

namespace {

int decompress_batch(int K, const void* d_comp, size_t comp_bytes, const uint64_t* d_comp_off,
                     const uint32_t* d_comp_size, size_t n_frames, void* d_raw, size_t raw_cap,
                     size_t frame_bytes, int32_t* d_status, bool shared, void* stream) {
  if (K < 1 || K > HFZ_MAX_STREAMS || !d_comp || !d_comp_off || !d_comp_size || frame_bytes == 0 ||
      (raw_cap && !d_raw))
    return HFZ_E_ARG;
  if (misaligned(d_comp_off, 8) || misaligned(d_comp_size, 4) || misaligned(d_status, 4))
    return HFZ_E_ARG;
  if (frame_bytes > HFZ_MAX_FRAME_BYTES) return HFZ_E_TOO_LARGE;
  HFZ_HIP(g_timing.take(g_timing.dec, g_timing.dec_calls, g_timing.dec_seen, kDecEvents, n_frames));
  HFZ_HIP(launch_decode(K, (const uint8_t*)d_comp, comp_bytes, d_comp_off, d_comp_size, n_frames,
                        shared, (uint8_t*)d_raw, raw_cap, frame_bytes, d_status, stream,
                        g_timing.span(0)));
  g_timing.cur = nullptr;
  return HFZ_OK;
}

}  // namespace

extern "C" {

int hfz_compress_batch_dev(int K, const void* d_raw, size_t total, size_t frame_bytes,
                           void* d_comp, size_t comp_cap, size_t comp_stride,
                           uint64_t* d_comp_off, uint32_t* d_comp_size, int32_t* d_status,
                           void* d_ws, size_t ws_bytes, void* stream) {
  return compress_batch(K, d_raw, total, frame_bytes, d_comp, comp_cap, comp_stride, d_comp_off,
                        d_comp_size, d_status, d_ws, ws_bytes, nullptr, stream);
}

int hfz_count_batch_dev(int K, const void* d_raw, size_t total, size_t frame_bytes, void* d_ws,
                        size_t ws_bytes, uint64_t* d_hist_total, void* stream) {
  if (K < 1 || K > HFZ_MAX_STREAMS || frame_bytes == 0 || !d_ws || !d_hist_total || (total && !d_raw))
    return HFZ_E_ARG;
  if (frame_bytes > HFZ_MAX_FRAME_BYTES) return HFZ_E_TOO_LARGE;
  const size_t nf = hfz_batch_frames(total, frame_bytes);
  const WorkspaceLayout w = workspace_layout(nf);
  if (ws_bytes < w.total) return HFZ_E_WORKSPACE;
  if (misaligned(d_ws, 256) || misaligned(d_hist_total, 8)) return HFZ_E_ARG;
  uint8_t* ws = (uint8_t*)d_ws;
  uint32_t* d_hist = (uint32_t*)(ws + w.hist);
  const uint8_t* raw = (const uint8_t*)d_raw;
  HFZ_HIP(g_timing.take(g_timing.cnt, g_timing.cnt_calls, g_timing.cnt_seen, kCntEvents, nf));
  // count_ms is the input pass alone; the 256-bin reduction below is not in it
  if (count_kernel_fits(K, frame_bytes)) {
    HFZ_HIP(launch_count(K, raw, total, frame_bytes, nf, nullptr, false, (uint32_t*)(ws + w.counts),
                         d_hist, (uint32_t*)(ws + w.stream_end), (uint32_t*)(ws + w.csize), stream,
                         g_timing.span(0)));
  } else {
    HFZ_HIP(launch_histogram(raw, total, frame_bytes, nf, d_hist, stream, false, g_timing.span(0)));
  }
  g_timing.cur = nullptr;
  HFZ_HIP(launch_hist_total(d_hist, nf, d_hist_total, stream));
  return HFZ_OK;
}

int hfz_compress_batch_counted_dev(int K, const void* d_raw, size_t total, size_t frame_bytes,
                                   const hfz_frame_code* d_code, void* d_comp, size_t comp_cap,
                                   size_t comp_stride, uint64_t* d_comp_off, uint32_t* d_comp_size,
                                   int32_t* d_status, void* d_ws, size_t ws_bytes, void* stream) {
  if (K < 1 || K > HFZ_MAX_STREAMS || frame_bytes == 0 || !d_comp || !d_comp_off || !d_comp_size ||
      !d_ws || !d_code || (total && !d_raw))
    return HFZ_E_ARG;
  if (frame_bytes > HFZ_MAX_FRAME_BYTES) return HFZ_E_TOO_LARGE;
  if (comp_stride && comp_stride < hfz_compress_bound(frame_bytes, K)) return HFZ_E_ARG;
  if (misaligned(d_ws, 256) || misaligned(d_comp_off, 8) || misaligned(d_comp_size, 4) ||
      misaligned(d_status, 4) || misaligned(d_code, 4))
    return HFZ_E_ARG;
  const size_t nf = hfz_batch_frames(total, frame_bytes);
  const WorkspaceLayout w = workspace_layout(nf);
  if (ws_bytes < w.total) return HFZ_E_WORKSPACE;
  uint8_t* ws = (uint8_t*)d_ws;
  uint32_t* d_send = (uint32_t*)(ws + w.stream_end);
  uint32_t* d_csize = (uint32_t*)(ws + w.csize);
  const FrameCode* code = (const FrameCode*)d_code;
  const uint8_t* raw = (const uint8_t*)d_raw;
  HFZ_HIP(g_timing.take(g_timing.enc, g_timing.enc_calls, g_timing.enc_seen, kEncEvents, nf));
  // the input pass was hfz_count_batch_dev (hist_ms reads 0 here)
  if (count_kernel_fits(K, frame_bytes)) {
    HFZ_HIP(launch_code_sizes(K, (const uint32_t*)(ws + w.hist), (const uint32_t*)(ws + w.counts), nf, nullptr, code, true,
                              d_send, d_csize, stream, g_timing.enc_span(1)));
  } else {
    HFZ_HIP(launch_stream_sizes(K, raw, total, frame_bytes, nf, nullptr, const_cast<FrameCode*>(code),
                                true, d_send, d_csize, stream, g_timing.enc_span(2)));
  }
  HFZ_HIP(launch_frame_offsets(d_csize, nf, comp_stride, comp_cap, d_comp_off, d_comp_size,
                               d_status, stream, g_timing.enc_span(3)));
  HFZ_HIP(launch_encode(K, raw, total, frame_bytes, nf, code, true, d_send, d_comp_off,
                        d_comp_size, (uint8_t*)d_comp, comp_cap, stream, g_timing.enc_span(4)));
  g_timing.cur = nullptr;
  return HFZ_OK;
}

int hfz_compress_batch_shared_dev(int K, const void* d_raw, size_t total, size_t frame_bytes,
                                  const hfz_frame_code* d_code, void* d_comp, size_t comp_cap,
                                  size_t comp_stride, uint64_t* d_comp_off, uint32_t* d_comp_size,
                                  int32_t* d_status, void* d_ws, size_t ws_bytes, void* stream) {
  if (!d_code) return HFZ_E_ARG;
  return compress_batch(K, d_raw, total, frame_bytes, d_comp, comp_cap, comp_stride, d_comp_off,
                        d_comp_size, d_status, d_ws, ws_bytes, (const FrameCode*)d_code, stream);
}

int hfz_decompress_batch_dev(int K, const void* d_comp, size_t comp_bytes,
                             const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                             size_t n_frames, void* d_raw, size_t raw_cap, size_t frame_bytes,
                             int32_t* d_status, void* stream) {
  return decompress_batch(K, d_comp, comp_bytes, d_comp_off, d_comp_size, n_frames, d_raw, raw_cap,
                          frame_bytes, d_status, false, stream);
}

int hfz_decompress_batch_shared_dev(int K, const void* d_comp, size_t comp_bytes,
                                    const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                                    size_t n_frames, void* d_raw, size_t raw_cap,
                                    size_t frame_bytes, int32_t* d_status, void* stream) {
  return decompress_batch(K, d_comp, comp_bytes, d_comp_off, d_comp_size, n_frames, d_raw, raw_cap,
                          frame_bytes, d_status, true, stream);
}

int hfz_build_codes_dev(const uint32_t* d_hist, size_t n, hfz_frame_code* d_codes, void* stream) {
  static_assert(sizeof(hfz_frame_code) == sizeof(FrameCode), "public and internal layout differ");
  if (!d_hist || !d_codes || misaligned(d_hist, 16) || misaligned(d_codes, 4)) return HFZ_E_ARG;
  HFZ_HIP(launch_build_codes(d_hist, n, (FrameCode*)d_codes, stream));
  return HFZ_OK;
}

int hfz_frame_code_from_table(const hfz_code_table* t, hfz_frame_code* out) {
  if (!t || !out) return HFZ_E_ARG;
  frame_code_from_table(*t, (FrameCode*)out);
  return HFZ_OK;
}

int hfz_set_decoder_table_bits(int bits) {
  if (bits != 0 && bits != 11 && bits != 12 && bits != HFZ_DECODER_TABLE_FREE) return HFZ_E_ARG;
  g_decoder_table_bits = bits;
  return HFZ_OK;
}

int hfz_set_frames_per_wave(int n) {
  if (n < 0 || n > 2) return HFZ_E_ARG;
  g_frames_per_wave = n;
  return HFZ_OK;
}

int hfz_set_encoder_block(int bytes) {
  if (bytes != 0 && bytes != 32 && bytes != 64) return HFZ_E_ARG;
  g_encoder_block = bytes;
  return HFZ_OK;
}

int hfz_set_encoder_parts(int n) {
  if (n < 0 || n > 64) return HFZ_E_ARG;
  g_encoder_parts = n;
  return HFZ_OK;
}

int hfz_set_decoder_parts(int n) {
  if (n < 0 || n > 64) return HFZ_E_ARG;
  g_decoder_parts = n;
  return HFZ_OK;
}

int hfz_set_decoder_input_block(int bytes) {
  if (bytes != 0 && bytes != 32 && bytes != 64) return HFZ_E_ARG;
  g_decoder_input_block = bytes;
  return HFZ_OK;
}

int hfz_set_decoder_store(int mode) {
  if (mode < 0 || mode > 2) return HFZ_E_ARG;
  g_decoder_store = mode;
  return HFZ_OK;
}

int hfz_timing_enable(int every) {
  if (every < 0) return HFZ_E_ARG;
  g_timing.every = every;
  g_timing.reset_window();
  if (every == 0) g_timing.release();
  return HFZ_OK;
}

int hfz_timing_read(hfz_batch_timing* out) {
  if (!out) return HFZ_E_ARG;
  std::memset(out, 0, sizeof(*out));
  float* dst[5] = {&out->hist_ms, &out->table_ms, &out->sizes_ms, &out->place_ms,
                   &out->encode_ms};
  size_t launched[5] = {0, 0, 0, 0, 0};  // timed calls that ran kernel i
  for (size_t c = 0; c < g_timing.enc_calls; ++c) {
    hipEvent_t* ev = &g_timing.enc[c * kEncEvents];
    for (int i = 0; i < kEncSpans; ++i) {
      if (!((g_timing.enc_used[c] >> i) & 1)) continue;  // that kernel was not part of the call
      HFZ_HIP(hipEventSynchronize(ev[2 * i + 1]));
      float ms = 0;
      HFZ_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
      *dst[i] += ms;
      ++launched[i];
    }
  }
  for (size_t c = 0; c < g_timing.dec_calls; ++c) {
    hipEvent_t* ev = &g_timing.dec[c * kDecEvents];
    HFZ_HIP(hipEventSynchronize(ev[1]));
    float ms = 0;
    HFZ_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
    out->decode_ms += ms;
  }
  for (size_t c = 0; c < g_timing.cnt_calls; ++c) {
    hipEvent_t* ev = &g_timing.cnt[c * kCntEvents];
    HFZ_HIP(hipEventSynchronize(ev[1]));
    float ms = 0;
    HFZ_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
    out->count_ms += ms;
  }
  // a kernel's mean is over the calls that launched it (a window may mix the
  // counting path, the separate-pass path and the shared-code variants)
  for (int i = 0; i < 5; ++i)
    if (launched[i]) *dst[i] /= (float)launched[i];
  if (g_timing.dec_calls) out->decode_ms /= (float)g_timing.dec_calls;
  if (g_timing.cnt_calls) out->count_ms /= (float)g_timing.cnt_calls;
  out->compress_calls = (int32_t)g_timing.enc_calls;
  out->decompress_calls = (int32_t)g_timing.dec_calls;
  out->count_calls = (int32_t)g_timing.cnt_calls;
  g_timing.reset_window();
  return HFZ_OK;
}

int hfz_generate_dev(const uint32_t* d_cdf, uint64_t seed, uint64_t first_index, void* d_out,
                     size_t n, void* stream) {
  if (!d_cdf || (n && !d_out)) return HFZ_E_ARG;
  HFZ_HIP(launch_generate(d_cdf, seed, first_index, (uint8_t*)d_out, n, stream));
  return HFZ_OK;
}

// ---- host-buffer single-frame entry points --------------------------------

namespace {
// Buffers of the host-buffer entry points (hfz_compress / hfz_decompress / hfz_histogram): one
// device allocation, one pinned host allocation and one stream, kept between
// calls so that a call costs two copies, the kernels and ONE synchronisation
// instead of four allocations and three blocking copies.  A call takes an
// arena of the current device from a pool (or makes one) and puts it back, so
// concurrent callers never share one; an arena that grew beyond kArenaKeep is
// freed instead of pooled.  Pooled arenas live until the process ends.
constexpr size_t kArenaKeep = size_t(256) << 20;
struct Arena {
  int device = -1;
  hipStream_t stream = nullptr;
  uint8_t* dev = nullptr;
  size_t dev_bytes = 0;
  uint8_t* pin = nullptr;
  size_t pin_bytes = 0;
  hipError_t reserve(size_t dev_need, size_t pin_need) {
    if (!stream) {
      hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
      if (e != hipSuccess) return e;
    }
    if (dev_need > dev_bytes) {
      if (dev) (void)hipFree(dev);
      dev = nullptr;
      dev_bytes = 0;
      const size_t want = dev_need + dev_need / 4;
      hipError_t e = hipMalloc((void**)&dev, want);
      if (e != hipSuccess) return e;
      dev_bytes = want;
    }
    if (pin_need > pin_bytes) {
      if (pin) (void)hipHostFree(pin);
      pin = nullptr;
      pin_bytes = 0;
      const size_t want = pin_need + pin_need / 4;
      hipError_t e = hipHostMalloc((void**)&pin, want, hipHostMallocDefault);
      if (e != hipSuccess) return e;
      pin_bytes = want;
    }
    return hipSuccess;
  }
  void release() {
    if (dev) (void)hipFree(dev);
    if (pin) (void)hipHostFree(pin);
    if (stream) (void)hipStreamDestroy(stream);
    dev = pin = nullptr;
    stream = nullptr;
    dev_bytes = pin_bytes = 0;
  }
};
std::mutex g_arena_mutex;
std::vector<Arena>* arena_pool() {
  static std::vector<Arena>* pool = new std::vector<Arena>();  // never destroyed: HIP may be gone at exit
  return pool;
}
// RAII handle: takes an arena of the current device, returns it on scope exit
struct ArenaLease {
  Arena a;
  bool ok = false;
  ArenaLease() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    ok = true;
    std::lock_guard<std::mutex> g(g_arena_mutex);
    auto& pool = *arena_pool();
    for (size_t i = 0; i < pool.size(); ++i) {
      if (pool[i].device == dev) {
        a = pool[i];
        pool[i] = pool.back();
        pool.pop_back();
        return;
      }
    }
    a.device = dev;
  }
  ~ArenaLease() {
    if (!ok) return;
    if (a.dev_bytes + a.pin_bytes > kArenaKeep) {
      a.release();
      return;
    }
    std::lock_guard<std::mutex> g(g_arena_mutex);
    arena_pool()->push_back(a);
  }
};
size_t up256(size_t n) { return (n + 255) & ~size_t(255); }
}  // namespace

int hfz_histogram(const void* src, size_t n, uint32_t hist[256]) {
  if (!hist || (n && !src)) return HFZ_E_ARG;
  if (!have_device()) return HFZ_E_NO_DEVICE;
  // device: 256 counters | input, from the pooled arena (no allocation once it is warm)
  constexpr size_t kHistBytes = 256 * sizeof(uint32_t);
  constexpr size_t kStaged = size_t(4) << 20;  // larger inputs are copied from the caller's memory directly
  const bool staged = n <= kStaged;
  ArenaLease lease;
  if (!lease.ok) return HFZ_E_HIP;
  Arena& ar = lease.a;
  HFZ_HIP(ar.reserve(kHistBytes + n, std::max(staged ? n : size_t(0), kHistBytes)));
  hipStream_t st = ar.stream;
  if (n) {
    if (staged) std::memcpy(ar.pin, src, n);
    HFZ_HIP(hipMemcpyAsync(ar.dev + kHistBytes, staged ? (const void*)ar.pin : src, n, hipMemcpyHostToDevice, st));
  }
  int rc = hfz_histogram_dev(ar.dev + kHistBytes, n, (uint32_t*)ar.dev, st);
  if (rc != HFZ_OK) {
    (void)hipStreamSynchronize(st);
    return rc;
  }
  HFZ_HIP(hipMemcpyAsync(ar.pin, ar.dev, kHistBytes, hipMemcpyDeviceToHost, st));
  HFZ_HIP(hipStreamSynchronize(st));
  std::memcpy(hist, ar.pin, kHistBytes);
  return HFZ_OK;
}

int hfz_compress(int K, const void* src, size_t n, void* dst, size_t cap, size_t* out_n) {
  if (K < 1 || K > HFZ_MAX_STREAMS || !dst || !out_n || (n && !src)) return HFZ_E_ARG;
  if (n > HFZ_MAX_FRAME_BYTES) return HFZ_E_TOO_LARGE;
  if (!have_device()) return HFZ_E_NO_DEVICE;
  const size_t bound = hfz_compress_bound(n, K);
  const size_t ws_bytes = hfz_workspace_bytes(K, 1);
  // device: raw | meta (32 bytes) + frame, contiguous so that one copy brings both back | workspace
  const size_t o_meta = up256(n), o_comp = o_meta + 32, o_ws = up256(o_comp + bound);
  // Frames of up to kOneCopy bytes come back in a single copy of meta + bound
  // (the size is not known before the kernels have run); larger ones fetch the
  // 32 bytes first and then exactly the frame.
  constexpr size_t kOneCopy = size_t(1) << 20;
  const bool one_copy = bound <= kOneCopy;
  constexpr size_t kStaged = size_t(4) << 20;  // larger inputs are copied from the caller's memory directly
  const bool staged = n <= kStaged;
  ArenaLease lease;
  if (!lease.ok) return HFZ_E_HIP;
  Arena& ar = lease.a;
  HFZ_HIP(ar.reserve(o_ws + ws_bytes, std::max(staged ? n : size_t(0), 32 + (one_copy ? bound : size_t(0)))));
  hipStream_t st = ar.stream;
  if (n) {
    if (staged) std::memcpy(ar.pin, src, n);
    HFZ_HIP(hipMemcpyAsync(ar.dev, staged ? (const void*)ar.pin : src, n, hipMemcpyHostToDevice, st));
  }
  uint64_t* d_off = (uint64_t*)(ar.dev + o_meta);
  uint32_t* d_size = (uint32_t*)(ar.dev + o_meta + 8);
  int32_t* d_status = (int32_t*)(ar.dev + o_meta + 16);
  int rc = hfz_compress_batch_dev(K, ar.dev, n, n ? n : 1, ar.dev + o_comp, bound, 0, d_off, d_size,
                                  d_status, ar.dev + o_ws, ws_bytes, st);
  if (rc != HFZ_OK) {
    (void)hipStreamSynchronize(st);
    return rc;
  }
  // the input has been consumed by the time the results are copied (same stream), so the
  // pinned buffer is reused for them
  HFZ_HIP(hipMemcpyAsync(ar.pin, ar.dev + o_meta, 32 + (one_copy ? bound : 0), hipMemcpyDeviceToHost, st));
  HFZ_HIP(hipStreamSynchronize(st));
  uint32_t size;
  int32_t status;
  std::memcpy(&size, ar.pin + 8, 4);
  std::memcpy(&status, ar.pin + 16, 4);
  if (status != HFZ_OK) return status;
  *out_n = size;
  if (size > cap) return HFZ_E_CAPACITY;
  if (one_copy) {
    std::memcpy(dst, ar.pin + 32, size);
  } else {
    HFZ_HIP(hipMemcpyAsync(dst, ar.dev + o_comp, size, hipMemcpyDeviceToHost, st));
    HFZ_HIP(hipStreamSynchronize(st));
  }
  return HFZ_OK;
}

int hfz_decompressed_size(const void* src, size_t n, size_t* raw_size) {
  if (!src || !raw_size) return HFZ_E_ARG;
  if (n < 8) return HFZ_E_CORRUPT;
  uint32_t v;
  std::memcpy(&v, src, 4);
  *raw_size = v;
  return HFZ_OK;
}

int hfz_decompress(int K, const void* src, size_t n, void* dst, size_t cap, size_t* out_n) {
  if (K < 1 || K > HFZ_MAX_STREAMS || !src || !out_n || (cap && !dst)) return HFZ_E_ARG;
  size_t raw_size = 0;
  int rc = hfz_decompressed_size(src, n, &raw_size);
  if (rc != HFZ_OK) return rc;
  *out_n = raw_size;
  if (raw_size > cap) return HFZ_E_CAPACITY;
  if (raw_size > HFZ_MAX_FRAME_BYTES) return HFZ_E_TOO_LARGE;
  if (n > 0xffffffffull) return HFZ_E_TOO_LARGE;
  if (!have_device()) return HFZ_E_NO_DEVICE;
  // device: meta (32 bytes) + frame, sent in one copy | meta back + raw, fetched in one copy
  const size_t o_comp = 32, o_back = up256(o_comp + n), o_raw = o_back + 32;
  // large frames go to and from the caller's memory directly, small ones through the pinned buffer
  constexpr size_t kStaged = size_t(4) << 20;
  const bool staged = n <= kStaged && raw_size <= kStaged;
  ArenaLease lease;
  if (!lease.ok) return HFZ_E_HIP;
  Arena& ar = lease.a;
  HFZ_HIP(ar.reserve(o_raw + raw_size, 32 + (staged ? std::max(n, raw_size) : size_t(0))));
  hipStream_t st = ar.stream;
  std::memset(ar.pin, 0, 32);
  const uint32_t size32 = (uint32_t)n;
  std::memcpy(ar.pin + 8, &size32, 4);
  if (staged) {
    std::memcpy(ar.pin + 32, src, n);
    HFZ_HIP(hipMemcpyAsync(ar.dev, ar.pin, 32 + n, hipMemcpyHostToDevice, st));
  } else {
    HFZ_HIP(hipMemcpyAsync(ar.dev, ar.pin, 32, hipMemcpyHostToDevice, st));
    HFZ_HIP(hipMemcpyAsync(ar.dev + o_comp, src, n, hipMemcpyHostToDevice, st));
  }
  int32_t* d_status = (int32_t*)(ar.dev + o_back + 16);
  rc = hfz_decompress_batch_dev(K, ar.dev + o_comp, n, (const uint64_t*)ar.dev, (const uint32_t*)(ar.dev + 8), 1,
                                ar.dev + o_raw, raw_size, raw_size ? raw_size : 1, d_status, st);
  if (rc != HFZ_OK) {
    (void)hipStreamSynchronize(st);
    return rc;
  }
  int32_t status;
  if (staged) {
    HFZ_HIP(hipMemcpyAsync(ar.pin, ar.dev + o_back, 32 + raw_size, hipMemcpyDeviceToHost, st));
    HFZ_HIP(hipStreamSynchronize(st));
    std::memcpy(&status, ar.pin + 16, 4);
    if (status != HFZ_OK) return status;
    if (raw_size) std::memcpy(dst, ar.pin + 32, raw_size);
  } else {
    // the frame's status first: dst stays untouched when the frame is refused, as
    // on the staged path
    HFZ_HIP(hipMemcpyAsync(ar.pin, ar.dev + o_back, 32, hipMemcpyDeviceToHost, st));
    HFZ_HIP(hipStreamSynchronize(st));
    std::memcpy(&status, ar.pin + 16, 4);
    if (status != HFZ_OK) return status;
    if (raw_size) {
      HFZ_HIP(hipMemcpyAsync(dst, ar.dev + o_raw, raw_size, hipMemcpyDeviceToHost, st));
      HFZ_HIP(hipStreamSynchronize(st));
    }
  }
  return HFZ_OK;
}

}  // extern "C"
