// hfz: file -> file tool over the batch C ABI (include/hfz.h).
//
// The reference's codec/compress_tool.cpp:1-15 is an empty stub and its format
// has no framing beyond one buffer, so this defines the smallest container
// that makes a file out of frames:
//
//   "HFZC" | u32 version = 1 | u32 K | u32 frame_bytes | u64 raw_bytes | u64 n_frames
//   | u32 frame_size[n_frames] | frames back to back
//
// Every frame is exactly CompressMulti<K>(chunk) of the reference
// (codec/huffman.cpp:738-846): frame i covers raw bytes [i*frame_bytes, ...).
// All integers little-endian.  The file is processed in batches of at most
// `--batch-mib` raw MiB so that device memory stays bounded.
//
//   hfz c [-k K] [-f frame_KiB] [-b batch_MiB] <in> <out>     compress
//   hfz d [-b batch_MiB] <in> <out>                            decompress
//   hfz t <in>                                                 decode and discard (integrity)
#include <hip/hip_runtime.h>

#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hfz.h"

namespace {

constexpr char kMagic[4] = {'H', 'F', 'Z', 'C'};

[[noreturn]] void die(const char* what, const char* detail = "") {
  std::fprintf(stderr, "hfz: %s%s%s\n", what, *detail ? ": " : "", detail);
  std::exit(1);
}
void hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess) die(what, hipGetErrorString(e));
}
void hfz_ok(int rc, const char* what) {
  if (rc != HFZ_OK) die(what, hfz_strerror(rc));
}
void put32(std::vector<uint8_t>& v, uint32_t x) {
  for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i)));
}
void put64(std::vector<uint8_t>& v, uint64_t x) {
  for (int i = 0; i < 8; ++i) v.push_back((uint8_t)(x >> (8 * i)));
}
uint32_t get32(const uint8_t* p) {
  return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}
uint64_t get64(const uint8_t* p) { return (uint64_t)get32(p) | (uint64_t)get32(p + 4) << 32; }

struct Dev {
  void* p = nullptr;
  void alloc(size_t n, const char* what) {
    release();
    hip_ok(hipMalloc(&p, n ? n : 1), what);
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  ~Dev() { release(); }
};

uint64_t file_size(FILE* f) {
  if (std::fseek(f, 0, SEEK_END) != 0) die("cannot seek input");
  const long long n = std::ftell(f);
  if (n < 0 || std::fseek(f, 0, SEEK_SET) != 0) die("cannot size input");
  return (uint64_t)n;
}

int compress_file(const char* in_path, const char* out_path, int K, uint32_t frame_bytes,
                  uint64_t batch_bytes) {
  FILE* in = std::fopen(in_path, "rb");
  if (!in) die("cannot open", in_path);
  FILE* out = std::fopen(out_path, "wb");
  if (!out) die("cannot create", out_path);
  const uint64_t raw_bytes = file_size(in);
  const uint64_t n_frames = hfz_batch_frames(raw_bytes, frame_bytes);
  batch_bytes = (batch_bytes / frame_bytes) * frame_bytes;
  if (batch_bytes == 0) batch_bytes = frame_bytes;

  // header with a zeroed size table, filled in at the end
  std::vector<uint8_t> head;
  head.insert(head.end(), kMagic, kMagic + 4);
  put32(head, 1);
  put32(head, (uint32_t)K);
  put32(head, frame_bytes);
  put64(head, raw_bytes);
  put64(head, n_frames);
  const size_t table_at = head.size();
  head.resize(head.size() + 4 * n_frames, 0);
  if (std::fwrite(head.data(), 1, head.size(), out) != head.size()) die("write failed");

  const uint64_t max_frames = batch_bytes / frame_bytes;
  const size_t bound = hfz_compress_bound(frame_bytes, K);
  const size_t ws_bytes = hfz_workspace_bytes(K, max_frames);
  Dev d_raw, d_comp, d_off, d_size, d_status, d_ws;
  d_raw.alloc(batch_bytes, "hipMalloc raw");
  d_comp.alloc(bound * max_frames, "hipMalloc comp");
  d_off.alloc(8 * max_frames, "hipMalloc offsets");
  d_size.alloc(4 * max_frames, "hipMalloc sizes");
  d_status.alloc(4 * max_frames, "hipMalloc status");
  d_ws.alloc(ws_bytes, "hipMalloc workspace");
  std::vector<uint8_t> host(batch_bytes), comp;
  std::vector<uint64_t> offs(max_frames);
  std::vector<uint32_t> sizes(max_frames);
  std::vector<int32_t> status(max_frames);

  uint64_t done = 0, frame0 = 0, comp_total = 0;
  do {  // an empty file is one empty frame, as hfz_batch_frames counts it
    const uint64_t n = raw_bytes - done < batch_bytes ? raw_bytes - done : batch_bytes;
    if (n && std::fread(host.data(), 1, n, in) != n) die("read failed", in_path);
    const uint64_t nf = hfz_batch_frames(n, frame_bytes);
    hip_ok(hipMemcpy(d_raw.p, host.data(), n, hipMemcpyHostToDevice), "copy in");
    hfz_ok(hfz_compress_batch_dev(K, d_raw.p, n, frame_bytes, d_comp.p, bound * max_frames, 0,
                                  (uint64_t*)d_off.p, (uint32_t*)d_size.p, (int32_t*)d_status.p,
                                  d_ws.p, ws_bytes, nullptr),
           "hfz_compress_batch_dev");
    hip_ok(hipMemcpy(offs.data(), d_off.p, 8 * nf, hipMemcpyDeviceToHost), "copy offsets");
    hip_ok(hipMemcpy(sizes.data(), d_size.p, 4 * nf, hipMemcpyDeviceToHost), "copy sizes");
    hip_ok(hipMemcpy(status.data(), d_status.p, 4 * nf, hipMemcpyDeviceToHost), "copy status");
    uint64_t bytes = 0;
    for (uint64_t i = 0; i < nf; ++i) {
      if (status[i] != HFZ_OK) die("frame failed", hfz_strerror(status[i]));
      if (offs[i] != bytes) die("frames are not packed");
      bytes += sizes[i];
      for (int b = 0; b < 4; ++b) head[table_at + 4 * (frame0 + i) + b] = (uint8_t)(sizes[i] >> (8 * b));
    }
    comp.resize(bytes);
    hip_ok(hipMemcpy(comp.data(), d_comp.p, bytes, hipMemcpyDeviceToHost), "copy out");
    if (std::fwrite(comp.data(), 1, bytes, out) != bytes) die("write failed", out_path);
    done += n;
    frame0 += nf;
    comp_total += bytes;
  } while (done < raw_bytes);
  if (std::fseek(out, 0, SEEK_SET) != 0 || std::fwrite(head.data(), 1, head.size(), out) != head.size())
    die("write failed", out_path);
  std::fclose(out);
  std::fclose(in);
  std::fprintf(stderr, "hfz: %" PRIu64 " -> %" PRIu64 " bytes, %" PRIu64 " frames, K=%d\n", raw_bytes,
               (uint64_t)head.size() + comp_total, n_frames, K);
  return 0;
}

int decompress_file(const char* in_path, const char* out_path, uint64_t batch_bytes) {
  FILE* in = std::fopen(in_path, "rb");
  if (!in) die("cannot open", in_path);
  const uint64_t file_bytes = file_size(in);
  uint8_t fixed[32];
  if (file_bytes < sizeof(fixed) || std::fread(fixed, 1, sizeof(fixed), in) != sizeof(fixed))
    die("not an hfz container (too short)");
  if (std::memcmp(fixed, kMagic, 4) != 0 || get32(fixed + 4) != 1) die("not an hfz container");
  const int K = (int)get32(fixed + 8);
  const uint32_t frame_bytes = get32(fixed + 12);
  const uint64_t raw_bytes = get64(fixed + 16);
  const uint64_t n_frames = get64(fixed + 24);
  if (K < 1 || K > HFZ_MAX_STREAMS || frame_bytes == 0 || frame_bytes > HFZ_MAX_FRAME_BYTES ||
      n_frames != hfz_batch_frames(raw_bytes, frame_bytes) || n_frames > (file_bytes - 32) / 4)
    die("corrupt container header");
  std::vector<uint32_t> all_sizes(n_frames);
  {
    std::vector<uint8_t> t(4 * n_frames);
    if (std::fread(t.data(), 1, t.size(), in) != t.size()) die("corrupt container (size table)");
    uint64_t sum = 32 + 4 * n_frames;
    for (uint64_t i = 0; i < n_frames; ++i) {
      all_sizes[i] = get32(&t[4 * i]);
      sum += all_sizes[i];
    }
    if (sum != file_bytes) die("corrupt container (sizes do not add up to the file size)");
  }
  FILE* out = nullptr;
  if (out_path) {
    out = std::fopen(out_path, "wb");
    if (!out) die("cannot create", out_path);
  }
  batch_bytes = (batch_bytes / frame_bytes) * frame_bytes;
  if (batch_bytes == 0) batch_bytes = frame_bytes;
  const uint64_t max_frames = batch_bytes / frame_bytes;
  const size_t bound = hfz_compress_bound(frame_bytes, K);
  Dev d_raw, d_comp, d_off, d_size, d_status;
  d_raw.alloc(batch_bytes, "hipMalloc raw");
  d_comp.alloc(bound * max_frames, "hipMalloc comp");
  d_off.alloc(8 * max_frames, "hipMalloc offsets");
  d_size.alloc(4 * max_frames, "hipMalloc sizes");
  d_status.alloc(4 * max_frames, "hipMalloc status");
  std::vector<uint8_t> comp, host(batch_bytes);
  std::vector<uint64_t> offs(max_frames);
  std::vector<int32_t> status(max_frames);

  for (uint64_t f0 = 0; f0 < n_frames; f0 += max_frames) {
    const uint64_t nf = n_frames - f0 < max_frames ? n_frames - f0 : max_frames;
    uint64_t bytes = 0;
    for (uint64_t i = 0; i < nf; ++i) {
      if (all_sizes[f0 + i] > bound) die("corrupt container (frame larger than the bound)");
      offs[i] = bytes;
      bytes += all_sizes[f0 + i];
    }
    comp.resize(bytes);
    if (std::fread(comp.data(), 1, bytes, in) != bytes) die("read failed", in_path);
    const uint64_t raw0 = f0 * frame_bytes;
    const uint64_t n = raw_bytes - raw0 < nf * (uint64_t)frame_bytes ? raw_bytes - raw0
                                                                     : nf * (uint64_t)frame_bytes;
    // A frame's header (word 0 = raw size) must claim exactly the bytes the
    // container gives it: frame_bytes, or what is left for the last frame.  The
    // device only refuses a frame that claims MORE than its room; one that
    // claims less would decode "ok" and leave stale bytes in the output.
    for (uint64_t i = 0; i < nf; ++i) {
      const uint64_t at = (f0 + i) * (uint64_t)frame_bytes;
      const uint64_t expect = raw_bytes - at < frame_bytes ? raw_bytes - at : frame_bytes;
      if (all_sizes[f0 + i] < 8 || get32(&comp[offs[i]]) != expect)
        die("corrupt frame (raw size in the frame header differs from the container's)");
    }
    hip_ok(hipMemcpy(d_comp.p, comp.data(), bytes, hipMemcpyHostToDevice), "copy in");
    hip_ok(hipMemcpy(d_off.p, offs.data(), 8 * nf, hipMemcpyHostToDevice), "copy offsets");
    hip_ok(hipMemcpy(d_size.p, &all_sizes[f0], 4 * nf, hipMemcpyHostToDevice), "copy sizes");
    hfz_ok(hfz_decompress_batch_dev(K, d_comp.p, bytes, (const uint64_t*)d_off.p,
                                    (const uint32_t*)d_size.p, nf, d_raw.p, n, frame_bytes,
                                    (int32_t*)d_status.p, nullptr),
           "hfz_decompress_batch_dev");
    hip_ok(hipMemcpy(status.data(), d_status.p, 4 * nf, hipMemcpyDeviceToHost), "copy status");
    for (uint64_t i = 0; i < nf; ++i)
      if (status[i] != HFZ_OK) die("corrupt frame", hfz_strerror(status[i]));
    if (out) {
      hip_ok(hipMemcpy(host.data(), d_raw.p, n, hipMemcpyDeviceToHost), "copy out");
      if (n && std::fwrite(host.data(), 1, n, out) != n) die("write failed", out_path);
    }
  }
  if (out) std::fclose(out);
  std::fclose(in);
  std::fprintf(stderr, "hfz: %" PRIu64 " frames, %" PRIu64 " raw bytes ok\n", n_frames, raw_bytes);
  return 0;
}

void usage() {
  std::fprintf(stderr,
               "usage: hfz c [-k K] [-f frame_KiB] [-b batch_MiB] <in> <out>\n"
               "       hfz d [-b batch_MiB] <in> <out>\n"
               "       hfz t <in>\n");
  std::exit(2);
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) usage();
  const std::string mode = argv[1];
  int K = 32;
  uint64_t frame_kib = 256, batch_mib = 256;
  std::vector<const char*> files;
  for (int i = 2; i < argc; ++i) {
    const std::string a = argv[i];
    if ((a == "-k" || a == "-f" || a == "-b") && i + 1 < argc) {
      const long long v = std::atoll(argv[++i]);
      if (v <= 0) usage();
      if (a == "-k") K = (int)v;
      if (a == "-f") frame_kib = (uint64_t)v;
      if (a == "-b") batch_mib = (uint64_t)v;
    } else {
      files.push_back(argv[i]);
    }
  }
  if (hfz_device_count() <= 0) die("no GPU", hfz_last_hip_error());
  if (mode == "c") {
    if (files.size() != 2 || K > HFZ_MAX_STREAMS || (frame_kib << 10) > HFZ_MAX_FRAME_BYTES) usage();
    return compress_file(files[0], files[1], K, (uint32_t)(frame_kib << 10), batch_mib << 20);
  }
  if (mode == "d") {
    if (files.size() != 2) usage();
    return decompress_file(files[0], files[1], batch_mib << 20);
  }
  if (mode == "t") {
    if (files.size() != 1) usage();
    return decompress_file(files[0], nullptr, batch_mib << 20);
  }
  usage();
}
