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
// `-b` raw MiB so that device memory stays bounded.
//
// Several GPUs (`-g N`) and several workers per GPU (`-j W`): frames are
// independent, so the file's batches (each a contiguous range of frames) are
// dealt to N * W workers in turn, batch b to worker b mod (N * W), and worker w
// runs on GPU w mod N; every worker is a host thread with a HIP stream and
// buffers of its own, reads its batches from the input by offset and hands its
// results over in batch order, so the output is the same file whatever N and W
// are.  No data moves between GPUs; a second worker on a GPU overlaps one
// batch's file I/O and copies with the other's kernels.
//
//   hfz c [-k K] [-f frame_KiB] [-b batch_MiB] [-g gpus] [-j workers] <in> <out>   compress
//   hfz d [-b batch_MiB] [-g gpus] [-j workers] <in> <out>                         decompress
//   hfz t [-b batch_MiB] [-g gpus] [-j workers] <in>                               decode and discard (integrity)
//   hfz p <n_frames> <frames_per_batch> <gpus> <workers_per_gpu>                   print the batch plan (no GPU needed)
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cinttypes>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hfz.h"

namespace {

constexpr char kMagic[4] = {'H', 'F', 'Z', 'C'};

// Called from any thread; the process ends at once (output files are
// incomplete after a failure whatever is flushed).
[[noreturn]] void die(const char* what, const char* detail = "") {
  std::fprintf(stderr, "hfz: %s%s%s\n", what, *detail ? ": " : "", detail);
  std::fflush(stderr);
  std::_Exit(1);
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

void read_at(int fd, void* dst, uint64_t n, uint64_t at, const char* path) {
  uint8_t* p = (uint8_t*)dst;
  while (n) {
    const ssize_t got = pread(fd, p, n, (off_t)at);
    if (got <= 0) die("read failed", path);
    p += got;
    at += (uint64_t)got;
    n -= (uint64_t)got;
  }
}
void write_at(int fd, const void* src, uint64_t n, uint64_t at, const char* path) {
  const uint8_t* p = (const uint8_t*)src;
  while (n) {
    const ssize_t put = pwrite(fd, p, n, (off_t)at);
    if (put <= 0) die("write failed", path);
    p += put;
    at += (uint64_t)put;
    n -= (uint64_t)put;
  }
}
uint64_t file_size(int fd, const char* path) {
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < 0) die("cannot size input", path);
  return (uint64_t)st.st_size;
}

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
struct Pinned {
  uint8_t* p = nullptr;
  void alloc(size_t n, const char* what) { hip_ok(hipHostMalloc((void**)&p, n ? n : 1, hipHostMallocDefault), what); }
  ~Pinned() {
    if (p) (void)hipHostFree(p);
  }
};

// The batch plan: batch b covers frames [b * per_batch, min(n_frames, (b + 1) *
// per_batch)) and goes to worker b mod n_workers, which runs on GPU
// (worker mod n_gpus).  Never more workers than batches.  An empty input is one
// (empty) frame, hence one batch.
struct Plan {
  uint64_t n_frames, per_batch, n_batches;
  int n_gpus, n_workers;
  Plan(uint64_t frames, uint64_t frames_per_batch, int gpus, int workers_per_gpu)
      : n_frames(frames), per_batch(frames_per_batch ? frames_per_batch : 1),
        n_batches((frames + per_batch - 1) / per_batch), n_gpus(gpus < 1 ? 1 : gpus) {
    n_workers = n_gpus * (workers_per_gpu < 1 ? 1 : workers_per_gpu);
    if ((uint64_t)n_workers > n_batches && n_batches > 0) n_workers = (int)n_batches;
    if (n_gpus > n_workers) n_gpus = n_workers;
  }
  uint64_t first(uint64_t b) const { return b * per_batch; }
  uint64_t count(uint64_t b) const {
    const uint64_t lo = first(b);
    return n_frames - lo < per_batch ? n_frames - lo : per_batch;
  }
  int worker(uint64_t b) const { return (int)(b % (uint64_t)n_workers); }
  int gpu_of(int w) const { return w % n_gpus; }
};

// Results leave in batch order: a worker waits until the batch before its own has been handed over.
struct Turn {
  std::mutex m;
  std::condition_variable cv;
  uint64_t next = 0;
  template <class F>
  void in_order(uint64_t b, F&& f) {
    std::unique_lock<std::mutex> lock(m);
    cv.wait(lock, [&] { return next == b; });
    f();
    ++next;
    cv.notify_all();
  }
};

int compress_file(const char* in_path, const char* out_path, int K, uint32_t frame_bytes,
                  uint64_t batch_bytes, int gpus, int per_gpu) {
  const int in = open(in_path, O_RDONLY);
  if (in < 0) die("cannot open", in_path);
  const int out = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (out < 0) die("cannot create", out_path);
  const uint64_t raw_bytes = file_size(in, in_path);
  const uint64_t n_frames = hfz_batch_frames(raw_bytes, frame_bytes);
  batch_bytes = (batch_bytes / frame_bytes) * frame_bytes;
  if (batch_bytes == 0) batch_bytes = frame_bytes;
  const uint64_t max_frames = batch_bytes / frame_bytes;
  const Plan plan(n_frames, max_frames, gpus, per_gpu);

  // header with a zeroed size table, filled in as the batches arrive
  std::vector<uint8_t> head;
  head.insert(head.end(), kMagic, kMagic + 4);
  put32(head, 1);
  put32(head, (uint32_t)K);
  put32(head, frame_bytes);
  put64(head, raw_bytes);
  put64(head, n_frames);
  const size_t table_at = head.size();
  head.resize(head.size() + 4 * n_frames, 0);
  uint64_t out_pos = head.size();  // guarded by turn.m
  Turn turn;

  auto worker = [&](int w) {
    hip_ok(hipSetDevice(plan.gpu_of(w)), "hipSetDevice");
    hipStream_t st;
    hip_ok(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    const size_t bound = hfz_compress_bound(frame_bytes, K);
    const size_t ws_bytes = hfz_workspace_bytes(K, max_frames);
    Dev d_raw, d_comp, d_off, d_size, d_status, d_ws;
    d_raw.alloc(batch_bytes, "hipMalloc raw");
    d_comp.alloc(bound * max_frames, "hipMalloc comp");
    d_off.alloc(8 * max_frames, "hipMalloc offsets");
    d_size.alloc(4 * max_frames, "hipMalloc sizes");
    d_status.alloc(4 * max_frames, "hipMalloc status");
    d_ws.alloc(ws_bytes, "hipMalloc workspace");
    Pinned host;
    host.alloc(batch_bytes, "hipHostMalloc");
    std::vector<uint8_t> comp;
    std::vector<uint64_t> offs(max_frames);
    std::vector<uint32_t> sizes(max_frames);
    std::vector<int32_t> status(max_frames);
    for (uint64_t b = (uint64_t)w; b < plan.n_batches; b += (uint64_t)plan.n_workers) {
      const uint64_t at = plan.first(b) * frame_bytes;
      const uint64_t n = raw_bytes - at < batch_bytes ? raw_bytes - at : batch_bytes;
      const uint64_t nf = plan.count(b);
      if (n) read_at(in, host.p, n, at, in_path);
      hip_ok(hipMemcpyAsync(d_raw.p, host.p, n, hipMemcpyHostToDevice, st), "copy in");
      hfz_ok(hfz_compress_batch_dev(K, d_raw.p, n, frame_bytes, d_comp.p, bound * max_frames, 0,
                                    (uint64_t*)d_off.p, (uint32_t*)d_size.p, (int32_t*)d_status.p,
                                    d_ws.p, ws_bytes, st),
             "hfz_compress_batch_dev");
      hip_ok(hipMemcpyAsync(offs.data(), d_off.p, 8 * nf, hipMemcpyDeviceToHost, st), "copy offsets");
      hip_ok(hipMemcpyAsync(sizes.data(), d_size.p, 4 * nf, hipMemcpyDeviceToHost, st), "copy sizes");
      hip_ok(hipMemcpyAsync(status.data(), d_status.p, 4 * nf, hipMemcpyDeviceToHost, st), "copy status");
      hip_ok(hipStreamSynchronize(st), "synchronize");
      uint64_t bytes = 0;
      for (uint64_t i = 0; i < nf; ++i) {
        if (status[i] != HFZ_OK) die("frame failed", hfz_strerror(status[i]));
        if (offs[i] != bytes) die("frames are not packed");
        bytes += sizes[i];
      }
      comp.resize(bytes);
      hip_ok(hipMemcpyAsync(comp.data(), d_comp.p, bytes, hipMemcpyDeviceToHost, st), "copy out");
      hip_ok(hipStreamSynchronize(st), "synchronize");
      turn.in_order(b, [&] {
        for (uint64_t i = 0; i < nf; ++i)
          for (int k = 0; k < 4; ++k)
            head[table_at + 4 * (plan.first(b) + i) + k] = (uint8_t)(sizes[i] >> (8 * k));
        write_at(out, comp.data(), bytes, out_pos, out_path);
        out_pos += bytes;
      });
    }
    (void)hipStreamDestroy(st);
  };
  std::vector<std::thread> threads;
  for (int w = 1; w < plan.n_workers; ++w) threads.emplace_back(worker, w);
  worker(0);
  for (auto& t : threads) t.join();
  write_at(out, head.data(), head.size(), 0, out_path);
  close(out);
  close(in);
  std::fprintf(stderr, "hfz: %" PRIu64 " -> %" PRIu64 " bytes, %" PRIu64 " frames, K=%d, %d GPU%s\n",
               raw_bytes, out_pos, n_frames, K, plan.n_gpus, plan.n_gpus == 1 ? "" : "s");
  return 0;
}

int decompress_file(const char* in_path, const char* out_path, uint64_t batch_bytes, int gpus,
                    int per_gpu) {
  const int in = open(in_path, O_RDONLY);
  if (in < 0) die("cannot open", in_path);
  const uint64_t file_bytes = file_size(in, in_path);
  uint8_t fixed[32];
  if (file_bytes < sizeof(fixed)) die("not an hfz container (too short)");
  read_at(in, fixed, sizeof(fixed), 0, in_path);
  if (std::memcmp(fixed, kMagic, 4) != 0 || get32(fixed + 4) != 1) die("not an hfz container");
  const int K = (int)get32(fixed + 8);
  const uint32_t frame_bytes = get32(fixed + 12);
  const uint64_t raw_bytes = get64(fixed + 16);
  const uint64_t n_frames = get64(fixed + 24);
  if (K < 1 || K > HFZ_MAX_STREAMS || frame_bytes == 0 || frame_bytes > HFZ_MAX_FRAME_BYTES ||
      n_frames != hfz_batch_frames(raw_bytes, frame_bytes) || n_frames > (file_bytes - 32) / 4)
    die("corrupt container header");
  std::vector<uint32_t> all_sizes(n_frames);
  std::vector<uint64_t> frame_at(n_frames + 1);  // file offset of every frame
  {
    std::vector<uint8_t> t(4 * n_frames);
    read_at(in, t.data(), t.size(), 32, in_path);
    uint64_t sum = 32 + 4 * n_frames;
    for (uint64_t i = 0; i < n_frames; ++i) {
      all_sizes[i] = get32(&t[4 * i]);
      frame_at[i] = sum;
      sum += all_sizes[i];
    }
    frame_at[n_frames] = sum;
    if (sum != file_bytes) die("corrupt container (sizes do not add up to the file size)");
  }
  int out = -1;
  if (out_path) {
    out = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (out < 0) die("cannot create", out_path);
  }
  batch_bytes = (batch_bytes / frame_bytes) * frame_bytes;
  if (batch_bytes == 0) batch_bytes = frame_bytes;
  const uint64_t max_frames = batch_bytes / frame_bytes;
  const Plan plan(n_frames, max_frames, gpus, per_gpu);
  const size_t bound = hfz_compress_bound(frame_bytes, K);

  // every batch knows where its bytes come from and where they go: no hand-over in order
  auto worker = [&](int w) {
    hip_ok(hipSetDevice(plan.gpu_of(w)), "hipSetDevice");
    hipStream_t st;
    hip_ok(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    Dev d_raw, d_comp, d_off, d_size, d_status;
    d_raw.alloc(batch_bytes, "hipMalloc raw");
    d_comp.alloc(bound * max_frames, "hipMalloc comp");
    d_off.alloc(8 * max_frames, "hipMalloc offsets");
    d_size.alloc(4 * max_frames, "hipMalloc sizes");
    d_status.alloc(4 * max_frames, "hipMalloc status");
    Pinned host;
    host.alloc(batch_bytes, "hipHostMalloc");
    std::vector<uint8_t> comp;
    std::vector<uint64_t> offs(max_frames);
    std::vector<int32_t> status(max_frames);
    for (uint64_t b = (uint64_t)w; b < plan.n_batches; b += (uint64_t)plan.n_workers) {
      const uint64_t f0 = plan.first(b), nf = plan.count(b);
      uint64_t bytes = 0;
      for (uint64_t i = 0; i < nf; ++i) {
        if (all_sizes[f0 + i] > bound) die("corrupt container (frame larger than the bound)");
        offs[i] = bytes;
        bytes += all_sizes[f0 + i];
      }
      comp.resize(bytes);
      read_at(in, comp.data(), bytes, frame_at[f0], in_path);
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
      hip_ok(hipMemcpyAsync(d_comp.p, comp.data(), bytes, hipMemcpyHostToDevice, st), "copy in");
      hip_ok(hipMemcpyAsync(d_off.p, offs.data(), 8 * nf, hipMemcpyHostToDevice, st), "copy offsets");
      hip_ok(hipMemcpyAsync(d_size.p, &all_sizes[f0], 4 * nf, hipMemcpyHostToDevice, st), "copy sizes");
      hfz_ok(hfz_decompress_batch_dev(K, d_comp.p, bytes, (const uint64_t*)d_off.p,
                                      (const uint32_t*)d_size.p, nf, d_raw.p, n, frame_bytes,
                                      (int32_t*)d_status.p, st),
             "hfz_decompress_batch_dev");
      hip_ok(hipMemcpyAsync(status.data(), d_status.p, 4 * nf, hipMemcpyDeviceToHost, st), "copy status");
      if (out >= 0) hip_ok(hipMemcpyAsync(host.p, d_raw.p, n, hipMemcpyDeviceToHost, st), "copy out");
      hip_ok(hipStreamSynchronize(st), "synchronize");
      for (uint64_t i = 0; i < nf; ++i)
        if (status[i] != HFZ_OK) die("corrupt frame", hfz_strerror(status[i]));
      if (out >= 0 && n) write_at(out, host.p, n, raw0, out_path);
    }
    (void)hipStreamDestroy(st);
  };
  std::vector<std::thread> threads;
  for (int w = 1; w < plan.n_workers; ++w) threads.emplace_back(worker, w);
  worker(0);
  for (auto& t : threads) t.join();
  if (out >= 0) close(out);
  close(in);
  std::fprintf(stderr, "hfz: %" PRIu64 " frames, %" PRIu64 " raw bytes ok, %d GPU%s\n", n_frames,
               raw_bytes, plan.n_gpus, plan.n_gpus == 1 ? "" : "s");
  return 0;
}

int print_plan(uint64_t n_frames, uint64_t per_batch, int gpus, int per_gpu) {
  const Plan plan(n_frames, per_batch, gpus, per_gpu);
  std::printf("frames %" PRIu64 " per_batch %" PRIu64 " batches %" PRIu64 " gpus %d workers %d\n",
              plan.n_frames, plan.per_batch, plan.n_batches, plan.n_gpus, plan.n_workers);
  for (uint64_t b = 0; b < plan.n_batches; ++b)
    std::printf("batch %" PRIu64 " frames %" PRIu64 " %" PRIu64 " worker %d gpu %d\n", b, plan.first(b),
                plan.first(b) + plan.count(b), plan.worker(b), plan.gpu_of(plan.worker(b)));
  return 0;
}

void usage() {
  std::fprintf(stderr,
               "usage: hfz c [-k K] [-f frame_KiB] [-b batch_MiB] [-g gpus] [-j workers_per_gpu] <in> <out>\n"
               "       hfz d [-b batch_MiB] [-g gpus] [-j workers_per_gpu] <in> <out>\n"
               "       hfz t [-b batch_MiB] [-g gpus] [-j workers_per_gpu] <in>\n"
               "       hfz p <n_frames> <frames_per_batch> <gpus> <workers_per_gpu>\n");
  std::exit(2);
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) usage();
  const std::string mode = argv[1];
  if (mode == "p") {
    if (argc != 6) usage();
    return print_plan(std::strtoull(argv[2], nullptr, 10), std::strtoull(argv[3], nullptr, 10),
                      std::atoi(argv[4]), std::atoi(argv[5]));
  }
  int K = 32, gpus = 1, per_gpu = 1;
  uint64_t frame_kib = 256, batch_mib = 256;
  std::vector<const char*> files;
  for (int i = 2; i < argc; ++i) {
    const std::string a = argv[i];
    if ((a == "-k" || a == "-f" || a == "-b" || a == "-g" || a == "-j") && i + 1 < argc) {
      const long long v = std::atoll(argv[++i]);
      if (v <= 0) usage();
      if (a == "-k") K = (int)v;
      if (a == "-f") frame_kib = (uint64_t)v;
      if (a == "-b") batch_mib = (uint64_t)v;
      if (a == "-g") gpus = (int)v;
      if (a == "-j") per_gpu = (int)(v > 8 ? 8 : v);
    } else {
      files.push_back(argv[i]);
    }
  }
  const int have = hfz_device_count();
  if (have <= 0) die("no GPU", hfz_last_hip_error());
  if (gpus > have) {
    std::fprintf(stderr, "hfz: %d GPUs asked for, %d visible: using %d\n", gpus, have, have);
    gpus = have;
  }
  if (mode == "c") {
    if (files.size() != 2 || K > HFZ_MAX_STREAMS || (frame_kib << 10) > HFZ_MAX_FRAME_BYTES) usage();
    return compress_file(files[0], files[1], K, (uint32_t)(frame_kib << 10), batch_mib << 20, gpus, per_gpu);
  }
  if (mode == "d") {
    if (files.size() != 2) usage();
    return decompress_file(files[0], files[1], batch_mib << 20, gpus, per_gpu);
  }
  if (mode == "t") {
    if (files.size() != 1) usage();
    return decompress_file(files[0], nullptr, batch_mib << 20, gpus, per_gpu);
  }
  usage();
}
