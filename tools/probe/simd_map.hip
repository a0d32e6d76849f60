// Where do one-wave workgroups land?  Each block records its HW_ID register and
// spins on dependent VALU work; the host prints how the blocks of one launch
// spread over CUs and SIMDs.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
template <int LDS_BYTES, int THREADS>
__global__ __launch_bounds__(THREADS) void spin(uint32_t* hwid, uint32_t* sink, int iters) {
  __shared__ uint32_t m[LDS_BYTES / 4];
  m[threadIdx.x] = threadIdx.x;
  __syncthreads();
  uint32_t x = m[(threadIdx.x * 5) & 63] + blockIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
  if ((threadIdx.x & 63) == 0)
    hwid[blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  if (x == 0x12345678u) sink[0] = x;
}
template <int LDS_BYTES, int THREADS>
void run(int blocks, int iters) {
  const int waves = blocks * (THREADS / 64);
  uint32_t *d, *s;
  hipMalloc(&d, waves * 4);
  hipMalloc(&s, 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  spin<LDS_BYTES, THREADS><<<blocks, THREADS>>>(d, s, iters);
  hipEventRecord(a);
  spin<LDS_BYTES, THREADS><<<blocks, THREADS>>>(d, s, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<uint32_t> h(waves);
  hipMemcpy(h.data(), d, waves * 4, hipMemcpyDeviceToHost);
  std::map<uint32_t, int> per_simd, per_cu;
  int simd_hist[4] = {0, 0, 0, 0};
  for (uint32_t v : h) {
    const uint32_t simd = (v >> 4) & 3;
    const uint32_t cu_key = v & 0xfffff00u & ~0xc0u;  // everything above pipe/simd/wave
    per_cu[cu_key]++;
    per_simd[cu_key | (simd << 4)]++;
    simd_hist[simd]++;
  }
  int max_simd = 0, max_cu = 0;
  for (auto& kv : per_simd) max_simd = kv.second > max_simd ? kv.second : max_simd;
  for (auto& kv : per_cu) max_cu = kv.second > max_cu ? kv.second : max_cu;
  int dist[9] = {0};
  for (auto& kv : per_simd) dist[kv.second > 8 ? 8 : kv.second]++;
  printf("   SIMDs holding 1..8 waves: %d %d %d %d %d %d %d %d (of %zu seen)\n", dist[1], dist[2], dist[3], dist[4], dist[5], dist[6], dist[7], dist[8], per_simd.size());
  printf("lds %6d threads %3d blocks %5d: %.3f ms  distinct CUs %zu (max %d waves)  distinct SIMDs %zu (max %d waves)  simd hist %d %d %d %d  sample hwid %08x %08x %08x\n",
         LDS_BYTES, THREADS, blocks, ms, per_cu.size(), max_cu, per_simd.size(), max_simd,
         simd_hist[0], simd_hist[1], simd_hist[2], simd_hist[3], h[0], h[1], h[waves - 1]);
  hipFree(d);
  hipFree(s);
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  run<20480, 64>(1024, iters);
  run<20480, 64>(2048, iters);
  run<20480, 64>(1536, iters);
  run<29696, 64>(1024, iters);
  run<29696, 64>(2048, iters);
  run<29696, 64>(4096, iters);
  run<21504, 64>(1024, iters);
  run<21504, 64>(4096, iters);
  run<4096, 64>(1024, iters);
  run<4096, 64>(4096, iters);
  run<4096, 256>(256, iters);
  run<4096, 256>(1024, iters);
  return 0;
}
