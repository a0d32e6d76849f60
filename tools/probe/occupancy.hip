// Prints LDS-related device properties and the resident-block limit for
// one-wave workgroups with a given static LDS footprint.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES>
__global__ __launch_bounds__(64) void k(uint32_t* out) {
  __shared__ uint32_t m[BYTES / 4];
  m[threadIdx.x] = threadIdx.x;
  __syncthreads();
  out[threadIdx.x] = m[(threadIdx.x * 7) % (BYTES / 4)];
}
template <int BYTES>
void show() {
  int n = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<BYTES>, 64, 0);
  printf("lds %6d B/block -> %d blocks/CU\n", BYTES, n);
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s CUs=%d sharedMemPerBlock=%zu sharedMemPerMultiprocessor=%zu maxSharedMemoryPerMultiProcessor=%zu regsPerMultiprocessor=%d\n",
         p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock, p.sharedMemPerMultiprocessor,
         p.maxSharedMemoryPerMultiProcessor, p.regsPerMultiprocessor);
  show<4096>(); show<8192>(); show<16384>(); show<21504>(); show<29696>(); show<32768>(); show<65536>();
  return 0;
}
