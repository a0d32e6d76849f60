// Read bandwidth of one-wave workgroups that each walk a 256 KiB region, for
// several lane->address mappings of the 16-byte loads.  Diagnostic only.
//   G = lanes that share one 128-byte line per load instruction
//   G=1 : lane l walks its own 4 KiB slice (64 lines per instruction)
//   G=8 : 8 lanes cover one 128-byte line, 8 slices per instruction
//   G=64: fully coalesced (1 KiB contiguous per instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int G>
__global__ __launch_bounds__(64) void walk(const uint8_t* __restrict__ src, uint32_t* out, uint32_t region) {
  const uint8_t* base = src + (size_t)blockIdx.x * region;
  const uint32_t lane = threadIdx.x;
  const uint32_t slice = region / 64;          // bytes per stream
  const uint32_t sub = lane % G, grp = lane / G;
  constexpr uint32_t kStreamsPerInstr = 64 / G;
  uint4 acc = make_uint4(0, 0, 0, 0);
  // each iteration of `pos` covers G*16 bytes of every stream; streams are
  // visited kStreamsPerInstr at a time
  for (uint32_t pos = 0; pos < slice; pos += 128) {
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
      // 8 instructions move 128 bytes of each of the 64 streams
      const uint32_t unit = i * 64 + lane;                 // 16-byte unit index among 512
      uint32_t stream, off;
      if (G == 1) { stream = lane; off = i * 16; }
      else if (G == 64) { stream = 0; off = 0; }
      else {
        const uint32_t line_of = (i * kStreamsPerInstr + grp);   // which stream's line, 0..63 over... 
        // G lanes * 16 B = G*16 bytes of a line per instruction; 128/(G*16) instructions per line
        constexpr uint32_t kInstrPerLine = 128 / (G * 16);
        stream = (i / kInstrPerLine) * kStreamsPerInstr + grp;
        off = (i % kInstrPerLine) * (G * 16) + sub * 16;
        (void)line_of;
      }
      const uint8_t* p = (G == 64) ? base + (size_t)(pos / 128) * 8192 + unit * 16
                                   : base + (size_t)stream * slice + pos + off;
      const uint4 v = *(const uint4*)p;
      acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[blockIdx.x] = acc.x;
}
template <int G>
void run(const uint8_t* d, uint32_t* o, size_t total, uint32_t region) {
  const int blocks = (int)(total / region);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  walk<G><<<blocks, 64>>>(d, o, region);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) walk<G><<<blocks, 64>>>(d, o, region);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("G=%2d region %4u KiB blocks %5d: %.4f ms  %.0f GB/s\n", G, region >> 10, blocks, ms, total / ms / 1e6);
}
int main() {
  const size_t total = 1ull << 30;
  uint8_t* d; uint32_t* o;
  hipMalloc(&d, total); hipMalloc(&o, 1 << 20);
  hipMemset(d, 1, total);
  for (uint32_t region : {256u << 10, 64u << 10}) {
    for (size_t t : {size_t(256) << 20, total}) {
      run<1>(d, o, t, region); run<2>(d, o, t, region); run<4>(d, o, t, region);
      run<8>(d, o, t, region); run<64>(d, o, t, region);
    }
  }
  return 0;
}
