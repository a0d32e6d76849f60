// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the codec kernels' access
// pattern: one-wave workgroups in which every lane walks a stream of its own in
// aligned blocks (four or two 16-byte accesses per block), with a pause between
// blocks like the decoder's ~70 lookups.  Every kernel moves a known number of
// bytes; run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
// passes) and compare.  Diagnostic only.
//   read64 : 64-byte blocks read  (decoder input, count kernel uses 128)
//   read32 : 32-byte blocks read  (encoder input)
//   write32: 32-byte blocks written (decoder/encoder output, small batches)
//   write64: 64-byte blocks written (large batches)
//   readco : fully coalesced 16 B/lane read of the same bytes (the calibrated case)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void pause_a_while() {
  __builtin_amdgcn_s_sleep(32);  // 32 * 64 clocks
}

template <int kBlk>
__global__ __launch_bounds__(64) void read_blocks(const uint8_t* __restrict__ src, uint32_t* out, uint32_t slice) {
  const uint8_t* p = src + ((size_t)blockIdx.x * 64 + threadIdx.x) * slice;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint32_t pos = 0; pos < slice; pos += kBlk) {
#pragma unroll
    for (int q = 0; q < kBlk / 16; ++q) {
      const uint4 v = *(const uint4*)(p + pos + 16 * q);
      acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
    pause_a_while();
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[blockIdx.x] = acc.x;
}

template <int kBlk>
__global__ __launch_bounds__(64) void write_blocks(uint8_t* __restrict__ dst, uint32_t slice) {
  uint8_t* p = dst + ((size_t)blockIdx.x * 64 + threadIdx.x) * slice;
  for (uint32_t pos = 0; pos < slice; pos += kBlk) {
#pragma unroll
    for (int q = 0; q < kBlk / 16; ++q)
      *(uint4*)(p + pos + 16 * q) = make_uint4(pos, q, blockIdx.x, threadIdx.x);
    pause_a_while();
  }
}

__global__ __launch_bounds__(256) void read_coalesced(const uint8_t* __restrict__ src, uint32_t* out, size_t n) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; i < n; i += (size_t)gridDim.x * 256 * 16) {
    const uint4 v = *(const uint4*)(src + i);
    acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[blockIdx.x] = acc.x;
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 1024;  // one-wave workgroups
  const uint32_t slice = 2048;                        // bytes per lane: a stream of a 128 KiB frame
  const size_t total = (size_t)waves * 64 * slice;
  uint8_t *a, *b;
  uint32_t* o;
  if (hipMalloc(&a, total) != hipSuccess || hipMalloc(&b, total) != hipSuccess ||
      hipMalloc(&o, 1 << 20) != hipSuccess) return 1;
  hipMemset(a, 1, total);
  hipMemset(b, 0, total);
  hipDeviceSynchronize();
  printf("waves %d bytes per kernel %zu\n", waves, total);
  // the batch is first touched by another kernel, as in the codec (count -> encode -> decode)
  read_coalesced<<<2048, 256>>>(a, o, total);
  read_blocks<64><<<waves, 64>>>(a, o, slice);
  read_blocks<32><<<waves, 64>>>(a, o, slice);
  write_blocks<32><<<waves, 64>>>(b, slice);
  write_blocks<64><<<waves, 64>>>(b, slice);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  printf("done\n");
  return 0;
}
