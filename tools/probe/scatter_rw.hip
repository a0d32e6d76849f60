// Read and write bandwidth of one-wave workgroups whose 64 lanes walk 64
// streams of a region (region/64 bytes apart), for several lane->address
// mappings of the 16-byte accesses.  Diagnostic only.
//   G = lanes that cover consecutive 16-byte pieces of ONE stream per instruction
//   G=1 : each lane its own stream (64 lines per instruction, 16 B each)
//   G=4 : 4 lanes cover 64 contiguous bytes of a stream, 16 streams per instruction
//   G=8 : 8 lanes cover one 128-byte line, 8 streams per instruction
//   G=64: 1 KiB contiguous per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int G, bool kWrite, bool kNt = false>
__global__ __launch_bounds__(64) void walk(uint8_t* __restrict__ mem, uint32_t* out, uint32_t region) {
  uint8_t* base = mem + (size_t)blockIdx.x * region;
  const uint32_t lane = threadIdx.x;
  const uint32_t slice = region / 64;  // bytes per stream
  const uint32_t sub = lane % G, grp = lane / G;
  constexpr uint32_t kStreams = 64 / G;           // streams per instruction
  constexpr uint32_t kPiece = G * 16;             // contiguous bytes per stream per instruction
  uint4 acc = make_uint4(lane, blockIdx.x, 3, 4);
  // one outer step moves 128 bytes of each of the 64 streams with 8 instructions
  for (uint32_t pos = 0; pos < slice; pos += 128) {
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
      uint32_t stream, off;
      if (G == 64) {
        stream = 0; off = 0;
      } else {
        // instruction i: piece (i % ppl) of the 128 bytes, stream set (i / ppl)
        constexpr uint32_t ppl = 128 / kPiece > 0 ? 128 / kPiece : 1;  // pieces per 128 bytes
        stream = (i / ppl) * kStreams + grp;
        off = (i % ppl) * kPiece + sub * 16;
      }
      uint8_t* p = (G == 64) ? base + (size_t)(pos / 128) * 8192 + (i * 64 + lane) * 16
                             : base + (size_t)stream * slice + pos + off;
      if (kWrite) {
        if (kNt) __builtin_nontemporal_store(acc.x, (uint32_t*)p), __builtin_nontemporal_store(acc.y, (uint32_t*)p + 1),
                 __builtin_nontemporal_store(acc.z, (uint32_t*)p + 2), __builtin_nontemporal_store(acc.w, (uint32_t*)p + 3);
        else *(uint4*)p = acc;
        acc.x += 1;
      } else {
        const uint4 v = *(const uint4*)p;
        acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[blockIdx.x] = acc.x;
}
template <int G, bool W, bool NT = false>
void run(uint8_t* d, uint32_t* o, size_t total, uint32_t region) {
  const int blocks = (int)(total / region);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  walk<G, W, NT><<<blocks, 64>>>(d, o, region);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) walk<G, W, NT><<<blocks, 64>>>(d, o, region);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double instr_per_cu = (double)total / 1024 / 256;
  printf("%s%s G=%2d total %5zu MiB region %4u KiB waves %5d: %.4f ms  %5.0f GB/s  %.1f ns per wave-instruction per CU\n",
         W ? "write" : "read ", NT ? " nt" : "   ", G, total >> 20, region >> 10, blocks, ms, total / ms / 1e6, ms * 1e6 / instr_per_cu);
}
int main() {
  const size_t cap = 2ull << 30;
  uint8_t* d; uint32_t* o;
  hipMalloc(&d, cap); hipMalloc(&o, 1 << 20);
  hipMemset(d, 1, cap);
  struct { size_t total; uint32_t region; } cfg[] = {{256u << 20, 128u << 10}, {256u << 20, 256u << 10}, {cap, 1u << 20}};
  for (auto c : cfg) {
    run<1, false>(d, o, c.total, c.region); run<2, false>(d, o, c.total, c.region); run<4, false>(d, o, c.total, c.region);
    run<8, false>(d, o, c.total, c.region); run<64, false>(d, o, c.total, c.region);
    run<1, true>(d, o, c.total, c.region); run<1, true, true>(d, o, c.total, c.region);
    run<2, true>(d, o, c.total, c.region); run<4, true>(d, o, c.total, c.region); run<4, true, true>(d, o, c.total, c.region);
    run<8, true>(d, o, c.total, c.region); run<64, true>(d, o, c.total, c.region);
  }
  return 0;
}
