// Issue cost of single instruction kinds on gfx950, one kind per kernel:
// 64 copies per loop iteration, W waves per SIMD.  Prints time per instruction
// per SIMD relative to v_add_u32 (which takes one quad-cycle).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define KERNEL(name, body, tail)                                                       \
  __global__ __launch_bounds__(64) void name(uint32_t* out, int iters) {               \
    __shared__ uint32_t lds[4096];                                                      \
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i * 2654435761u;              \
    __syncthreads();                                                                    \
    uint32_t a = threadIdx.x * 4, b = blockIdx.x, c = 3, d = 5;                         \
    uint32_t r = ((threadIdx.x * 2654435761u + blockIdx.x) >> 7) & 0x3ffc;              \
    uint64_t q = threadIdx.x, s = 7;                                                    \
    for (int i = 0; i < iters; ++i) {                                                   \
      asm volatile(REP64(body) tail                                                     \
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(q), "+v"(s), "+v"(r)      \
                   :: "memory");                                                        \
    }                                                                                   \
    if ((a ^ b ^ c ^ d ^ (uint32_t)q ^ (uint32_t)s) == 0x1234567u) out[0] = a;          \
  }

KERNEL(k_add_dep, "v_add_u32 %1, %1, %0\n", "")
KERNEL(k_add_ind, "v_add_u32 %1, %0, %2\n v_add_u32 %3, %0, %2\n", "")
KERNEL(k_shl64_dep, "v_lshlrev_b64 %4, %2, %4\n", "")
KERNEL(k_shl64_ind, "v_lshlrev_b64 %4, %2, %5\n", "")
KERNEL(k_or3, "v_or3_b32 %1, %1, %0, %2\n", "")
KERNEL(k_add3, "v_add3_u32 %1, %1, %0, %2\n", "")
KERNEL(k_lshl_add, "v_lshl_add_u32 %1, %1, 3, %2\n", "")
KERNEL(k_bfe, "v_bfe_u32 %1, %0, 8, 8\n", "")
KERNEL(k_and_or, "v_and_or_b32 %1, %1, %0, %2\n", "")
KERNEL(k_sdwa, "v_add_u32_sdwa %1, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n", "")
KERNEL(k_alignbyte, "v_alignbyte_b32 %1, %1, %0, %2\n", "")
KERNEL(k_mul_lo, "v_mul_lo_u32 %1, %1, %2\n", "")
KERNEL(k_mad24, "v_mad_u32_u24 %1, %1, %2, %0\n", "")
KERNEL(k_cndmask, "v_cndmask_b32 %1, %1, %0, vcc\n", "")
KERNEL(k_cmp, "v_cmp_gt_i32 vcc, 0, %1\n", "")
KERNEL(k_salu, "s_add_u32 s20, s20, 1\n", "")
KERNEL(k_lds_rd_lin, "ds_read_b32 %1, %0\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_rd_rand, "ds_read_b32 %1, %6\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_rd_u8, "ds_read_u8 %1, %6\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_rd_dep, "ds_read_b32 %6, %6\n s_waitcnt lgkmcnt(0)\n v_and_b32 %6, 0x3ffc, %6\n", "")
KERNEL(k_lds_wr_lin, "ds_write_b32 %0, %1\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_wr_rand, "ds_write_b32 %6, %1\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_add_lin, "ds_add_u32 %0, %2\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_add_rand, "ds_add_u32 %6, %2\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_rd64, "ds_read_b64 %4, %0\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_rd2, "ds_read2_b32 %4, %0 offset1:1\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_mix_valu_lds, "ds_read_b32 %1, %6\n v_add_u32 %3, %0, %2\n v_add_u32 %3, %0, %2\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_mix_valu_salu, "v_add_u32 %3, %0, %2\n s_add_u32 s20, s20, 1\n", "")
KERNEL(k_mix_wait, "v_add_u32 %3, %0, %2\n s_waitcnt lgkmcnt(0)\n", "")
KERNEL(k_lshr32_e32, "v_lshrrev_b32_e32 %1, 3, %1\n", "")
KERNEL(k_and_lit, "v_and_b32_e32 %1, 0x1ffc, %1\n", "")
KERNEL(k_alignbit, "v_alignbit_b32 %1, %1, %0, %2\n", "")
KERNEL(k_perm, "v_perm_b32 %1, %1, %0, %2\n", "")
KERNEL(k_lshr64, "v_lshrrev_b64 %4, %2, %4\n", "")
KERNEL(k_lds_wr16_rand, "ds_write_b16 %6, %1\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_wr16hi_rand, "ds_write_b16_d16_hi %6, %1\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_wr8_rand, "ds_write_b8 %6, %1\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_wr64_lin, "ds_write_b64 %0, %4\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_rd_u16_rand, "ds_read_u16 %1, %6\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_rd64_rand, "ds_read_b64 %4, %6\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_mix_rd_wr, "ds_read_b32 %1, %6\n ds_write_b16 %6, %2\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_saveexec, "v_cmp_gt_i32 vcc, 0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n", "")

KERNEL(k_cnd_cmp, "v_cmp_gt_i32 vcc, 0, %2\n v_cndmask_b32 %1, %1, %0, vcc\n", "")
KERNEL(k_cnd_e64, "v_cndmask_b32_e64 %1, %1, %0, s[20:21]\n", "")
KERNEL(k_cnd_ind, "v_cndmask_b32 %1, %2, %0, vcc\n v_cndmask_b32 %3, %2, %0, vcc\n", "")
KERNEL(k_bfi, "v_bfi_b32 %1, %0, %1, %2\n", "")
KERNEL(k_bfe_i, "v_bfe_i32 %1, %1, 5, 1\n", "")
KERNEL(k_bfrev, "v_bfrev_b32 %1, %1\n", "")
KERNEL(k_mix10_rd, "ds_read_b32 %3, %6\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_mix10_wr, "ds_write_b32 %6, %2\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n", "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_mix10_vop3, "v_or3_b32 %3, %3, %0, %2\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n", "")
KERNEL(k_mix_vop3_2, "v_or3_b32 %3, %3, %0, %2\n v_add_u32 %1, %1, %0\n", "")
KERNEL(k_add10, "v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n v_add_u32 %1, %1, %0\n", "")

typedef void (*kern_t)(uint32_t*, int);
struct Item { const char* name; kern_t k; int per_rep; };

int main(int argc, char** argv) {
  const int iters = 400;
  uint32_t* d;
  hipMalloc(&d, 4096);
  std::vector<Item> items = {
      {"v_add_u32 dep", k_add_dep, 1}, {"v_add_u32 indep", k_add_ind, 2},
      {"v_lshlrev_b64 dep", k_shl64_dep, 1}, {"v_lshlrev_b64 indep", k_shl64_ind, 1},
      {"v_or3_b32", k_or3, 1}, {"v_add3_u32", k_add3, 1}, {"v_lshl_add_u32", k_lshl_add, 1},
      {"v_bfe_u32", k_bfe, 1}, {"v_and_or_b32", k_and_or, 1}, {"v_add_u32_sdwa", k_sdwa, 1},
      {"v_alignbyte_b32", k_alignbyte, 1}, {"v_mul_lo_u32", k_mul_lo, 1},
      {"v_mad_u32_u24", k_mad24, 1}, {"v_cndmask_b32", k_cndmask, 1}, {"v_cmp (vcc)", k_cmp, 1},
      {"s_add_u32", k_salu, 1},
      {"ds_read_b32 linear", k_lds_rd_lin, 1}, {"ds_read_b32 random", k_lds_rd_rand, 1},
      {"ds_read_u8 random", k_lds_rd_u8, 1}, {"ds_read_b32 dependent chain (+and)", k_lds_rd_dep, 1},
      {"ds_write_b32 linear", k_lds_wr_lin, 1}, {"ds_write_b32 random", k_lds_wr_rand, 1},
      {"ds_add_u32 linear", k_lds_add_lin, 1}, {"ds_add_u32 random", k_lds_add_rand, 1},
      {"ds_read_b64 linear", k_lds_rd64, 1}, {"ds_read2_b32 linear", k_lds_rd2, 1},
      {"mix: 1 ds_read random + 2 v_add", k_mix_valu_lds, 1},
      {"mix: v_add + s_add", k_mix_valu_salu, 1}, {"mix: v_add + s_waitcnt", k_mix_wait, 1},
      {"v_cmp + s_and_saveexec + s_or exec", k_saveexec, 1},
      {"v_lshrrev_b32_e32", k_lshr32_e32, 1}, {"v_and_b32_e32 literal", k_and_lit, 1},
      {"v_alignbit_b32", k_alignbit, 1}, {"v_perm_b32", k_perm, 1}, {"v_lshrrev_b64 dep", k_lshr64, 1},
      {"ds_write_b16 random", k_lds_wr16_rand, 1}, {"ds_write_b16_d16_hi random", k_lds_wr16hi_rand, 1},
      {"ds_write_b8 random", k_lds_wr8_rand, 1}, {"ds_write_b64 linear", k_lds_wr64_lin, 1},
      {"ds_read_u16 random", k_lds_rd_u16_rand, 1}, {"ds_read_b64 random", k_lds_rd64_rand, 1},
      {"mix: ds_read_b32 + ds_write_b16 random", k_mix_rd_wr, 1},
      {"v_cmp + v_cndmask (vcc)", k_cnd_cmp, 1}, {"v_cndmask_b32_e64 sgpr mask", k_cnd_e64, 1},
      {"v_cndmask indep x2", k_cnd_ind, 2}, {"v_bfi_b32", k_bfi, 1}, {"v_bfe_i32", k_bfe_i, 1}, {"v_bfrev_b32", k_bfrev, 1},
      {"10 v_add", k_add10, 1}, {"mix: ds_read + 10 v_add", k_mix10_rd, 1}, {"mix: ds_write + 10 v_add", k_mix10_wr, 1},
      {"mix: v_or3 + 4 v_add", k_mix10_vop3, 1}, {"mix: v_or3 + v_add", k_mix_vop3_2, 1}};
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  double base[3] = {0, 0, 0};
  printf("%-40s %12s %12s %12s   (quad-cycles per SIMD per instruction group, at 1 / 2 / 4 waves per SIMD)\n",
         "instruction", "1 wave", "2 waves", "4 waves");
  for (auto& it : items) {
    double res[3];
    for (int wi = 0; wi < 3; ++wi) {
      const int waves = 1 << wi;
      const int blocks = 256 * 4 * waves;
      it.k<<<blocks, 64>>>(d, 10);
      hipEventRecord(a);
      it.k<<<blocks, 64>>>(d, iters);
      hipEventRecord(b);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      // time per group per SIMD: each SIMD runs `waves` waves * iters * 64 groups
      res[wi] = ms * 1e6 / ((double)waves * iters * 64);  // ns per group per SIMD
      if (it.k == k_add_dep) base[wi] = res[wi];
    }
    printf("%-40s %12.2f %12.2f %12.2f\n", it.name, res[0] / base[2], res[1] / base[2], res[2] / base[2]);
  }
  printf("v_add_u32 at 4 waves/SIMD: %.3f ns per instruction per SIMD\n", base[2]);
  return 0;
}
