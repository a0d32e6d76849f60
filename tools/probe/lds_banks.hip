// How many LDS banks does a wave64 atomic see?  Each lane adds to one fixed
// address; the patterns differ only in how lanes map to banks.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(const uint32_t* addr_tab, int pattern, int iters, uint32_t* out) {
  __shared__ uint32_t lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = 0;
  __syncthreads();
  const uint32_t a = addr_tab[pattern * 64 + threadIdx.x];  // dword index
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) atomicAdd(&lds[a], 1u);
  }
  __syncthreads();
  if (lds[threadIdx.x] == 0xffffffffu) out[0] = 1;
}
int main() {
  const char* names[] = {"linear: lane",
                         "32 columns, random rows (row stride 32 dwords): lanes l, l+32 share a column",
                         "64 columns, random rows (row stride 64 dwords)",
                         "32 columns, rows of equal parity",
                         "all lanes one bank, different rows (stride 32)",
                         "all lanes one bank, different rows (stride 64)",
                         "pairs 2j,2j+1 share a column of 32, random rows"};
  const int np = 7;
  uint32_t h[np * 64];
  uint32_t seed = 12345;
  for (int l = 0; l < 64; ++l) {
    seed = seed * 1664525u + 1013904223u;
    const uint32_t row = (seed >> 16) & 255;
    h[0 * 64 + l] = l;
    h[1 * 64 + l] = (row & 127) * 32 + (l & 31);
    h[2 * 64 + l] = (row & 127) * 64 + l;
    h[3 * 64 + l] = ((row & 63) * 2) * 32 + (l & 31);
    h[4 * 64 + l] = l * 32;
    h[5 * 64 + l] = l * 64;
    h[6 * 64 + l] = (row & 127) * 32 + (l >> 1);
  }
  uint32_t *d, *o;
  hipMalloc(&d, sizeof(h));
  hipMalloc(&o, 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 2000, blocks = 256 * 16;
  for (int p = 0; p < np; ++p) {
    k<<<blocks, 64>>>(d, p, 10, o);
    hipEventRecord(a);
    k<<<blocks, 64>>>(d, p, iters, o);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // per CU: 16 waves * iters * 16 atomics
    printf("%-82s %.2f ns per wave-atomic per CU\n", names[p], ms * 1e6 / (16.0 * iters * 16));
  }
  return 0;
}
