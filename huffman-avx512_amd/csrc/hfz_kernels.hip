// gfx950 kernels of the K-stream Huffman codec: per-stream byte counts (and the
// plain histogram), canonical code construction, per-stream coded sizes, frame
// placement, encode, decode-table build + decode, and the synthetic data source.
// In the encoder and the decoder one wavefront (64 lanes) owns one frame and one
// lane one stream of it (two frames per wave for K <= 32 on large batches).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "hfz_internal.h"

namespace hfz {
namespace {

typedef uint32_t __attribute__((aligned(1))) u32_u;

__device__ __forceinline__ uint32_t load_u32u(const uint8_t* p) { return *(const u32_u*)p; }
__device__ __forceinline__ void store_u32u(uint8_t* p, uint32_t v) { *(u32_u*)p = v; }

// SliceSizes<K> (codec/huffman.cpp:98-108) in closed form: slice k of n bytes.
struct Slice {
  uint32_t begin, size;
};
__device__ __forceinline__ Slice slice_of(uint32_t n, uint32_t K, uint32_t k) {
  const uint32_t q = n / K, r = n % K;
  Slice s;
  s.begin = k * q + (k < r ? k : r);
  s.size = q + (k < r ? 1u : 0u);
  return s;
}

// Diagnostic build only (-DHFZ_STAMP): per-phase cycle sums of the decode and
// encode kernels and of the code builder, read back with hfz_debug_stamps().
// Never compiled into libhfz.so.
#ifdef HFZ_STAMP
__device__ unsigned long long g_stamps[32];
#define STAMP_DECL() unsigned long long st_prev_ = __builtin_amdgcn_s_memtime(), st_acc_[8] = {}
#define STAMP(i)                                                    \
  do {                                                              \
    __builtin_amdgcn_s_waitcnt(0xC07F);                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
    st_acc_[i] += now_ - st_prev_;                                  \
    st_prev_ = now_;                                                \
  } while (0)
#define STAMP_FLUSH(base)                                           \
  do {                                                              \
    if (threadIdx.x == 0)                                           \
      for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_stamps[(base) + i_], st_acc_[i_]); \
  } while (0)
#else
#define STAMP_DECL() do {} while (0)
#define STAMP(i) do {} while (0)
#define STAMP_FLUSH(base) do {} while (0)
#endif

// Ordering point between LDS accesses of different lanes of ONE wave.  The
// codec kernels run one wave per workgroup; a wave's LDS instructions execute in
// issue order, so all that is needed is that the compiler keeps the order.
// __syncthreads() would also wait for every global load and store in flight
// (s_waitcnt vmcnt(0)), which serialises each round on HBM latency.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Stores the part of the aligned 32-byte block [a, a+32) (offsets from base,
// data in src[0..7]) that lies inside [lo, hi): whole dwords where possible,
// single bytes only in the two dwords that straddle lo or hi.
__device__ __forceinline__ void store_block_clipped(uint8_t* base, uint32_t a, const uint32_t* src,
                                                    uint32_t lo, uint32_t hi) {
#pragma unroll
  for (uint32_t d = 0; d < 8; ++d) {
    const uint32_t p = a + 4 * d;
    const uint32_t w = src[d];
    if (p >= lo && p + 4 <= hi) {
      *(uint32_t*)(base + p) = w;
    } else if (p < hi && p + 4 > lo) {
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j)
        if (p + j >= lo && p + j < hi) base[p + j] = (uint8_t)(w >> (8 * j));
    }
  }
}

__device__ __forceinline__ uint32_t frame_len(uint64_t total, uint64_t frame_bytes, uint64_t f) {
  const uint64_t fb = f * frame_bytes;
  const uint64_t left = total > fb ? total - fb : 0;
  return (uint32_t)(left < frame_bytes ? left : frame_bytes);
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v += o;
  }
  return v;
}

// ---------------------------------------------------------------------------
// K1: byte histogram (codec/histogram.cpp:193-201 computes the same counts).
// 8 interleaved sub-histograms per block (bin*8 + (lane&7)) keep equal bytes
// in neighbouring lanes off one LDS address; the reference spreads its counts
// over 8 arrays for the same reason (codec/histogram.cpp:18).
__global__ __launch_bounds__(256) void hist_kernel(const uint8_t* __restrict__ raw,
                                                   uint64_t total, uint64_t frame_bytes,
                                                   uint32_t bpf, uint32_t accumulate,
                                                   uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256 * 8];
  const uint32_t tid = threadIdx.x;
  const uint64_t f = blockIdx.x / bpf;
  const uint32_t b = blockIdx.x % bpf;
  for (uint32_t i = tid; i < 2048; i += 256) h[i] = 0;
  __syncthreads();

  const uint64_t n = frame_len(total, frame_bytes, f);
  const uint64_t piece = ((n + bpf - 1) / bpf + 15) & ~uint64_t(15);
  const uint64_t lo = (uint64_t)b * piece < n ? (uint64_t)b * piece : n;
  const uint64_t hi = lo + piece < n ? lo + piece : n;
  const uint8_t* s = raw + f * frame_bytes + lo;
  const uint8_t* e = raw + f * frame_bytes + hi;
  const uint8_t* a0 = s + ((16 - ((uintptr_t)s & 15)) & 15);  // pointer arithmetic keeps
  const uint8_t* a1 = e - ((uintptr_t)e & 15);               // the global address space
  const uint32_t rep = tid & 7;
  if (a0 >= a1) {
    for (const uint8_t* q = s + tid; q < e; q += 256) atomicAdd(&h[(uint32_t)*q * 8 + rep], 1u);
  } else {
    for (const uint8_t* q = s + tid; q < a0; q += 256) atomicAdd(&h[(uint32_t)*q * 8 + rep], 1u);
    for (const uint8_t* q = a1 + tid; q < e; q += 256) atomicAdd(&h[(uint32_t)*q * 8 + rep], 1u);
    for (const uint8_t* q = a0 + (size_t)tid * 16; q < a1; q += 256 * 16) {
      const uint4 v = *(const uint4*)q;
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          atomicAdd(&h[((w[j] >> (8 * k)) & 0xff) * 8 + rep], 1u);
      }
    }
  }
  __syncthreads();
  uint32_t sum = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) sum += h[tid * 8 + r];
  if (bpf == 1 && !accumulate)
    hist[f * 256 + tid] = sum;
  else if (sum)
    atomicAdd(&hist[f * 256 + tid], sum);
}

// ---------------------------------------------------------------------------
// Canonical code construction on the device, one wave per frame:
// MakeCanonicalCoding (codec/huffman.cpp:339-437) from the frame's histogram.
//
// The order of equal-count symbols is part of the frame (it decides
// sorted_syms and with it every code), and the reference gets it from
// libstdc++'s std::sort (:353-354), so that algorithm is followed step by
// step: the introsort partition loop (median of three to the front, unguarded
// Hoare partition, segments of at most 16 left alone, heapsort once the depth
// budget 2*floor(log2 n) is spent) is reproduced swap for swap, each partition
// evaluated by the whole wave (see introsort_loop); the insertion sort
// that finishes std::sort is stable, so its result is the stable order of the
// partitioned array and all lanes compute it as ranks.  The two-queue Huffman
// merge (:364-416, a leaf wins a tie, :375) is serial as well; depths come from
// parent links by pointer jumping.
struct SortArrays {
  uint8_t* sym;
  uint32_t* key;
};
__device__ __forceinline__ bool sort_less(const SortArrays& a, int i, int j) {
  return a.key[i] > a.key[j];  // comparator of codec/huffman.cpp:354
}
__device__ __forceinline__ void sort_swap(const SortArrays& a, int i, int j) {
  const uint8_t s = a.sym[i];
  const uint32_t k = a.key[i];
  a.sym[i] = a.sym[j];
  a.key[i] = a.key[j];
  a.sym[j] = s;
  a.key[j] = k;
}
__device__ __forceinline__ void sort_move(const SortArrays& a, int dst, int src) {
  a.sym[dst] = a.sym[src];
  a.key[dst] = a.key[src];
}

// std::__adjust_heap + std::__push_heap on [first, first+len), value (vs, vk)
__device__ void heap_adjust(const SortArrays& a, int first, int hole, int len, uint8_t vs,
                            uint32_t vk) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (sort_less(a, first + child, first + child - 1)) child--;
    sort_move(a, first + hole, first + child);
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    sort_move(a, first + hole, first + child - 1);
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && a.key[first + parent] > vk) {
    sort_move(a, first + hole, first + parent);
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a.sym[first + hole] = vs;
  a.key[first + hole] = vk;
}

// std::__partial_sort(first, last, last): make_heap then sort_heap
__device__ void heap_sort_range(const SortArrays& a, int first, int last) {
  const int len = last - first;
  if (len >= 2) {
    for (int parent = (len - 2) / 2;; --parent) {
      heap_adjust(a, first, parent, len, a.sym[first + parent], a.key[first + parent]);
      if (parent == 0) break;
    }
  }
  while (last - first > 1) {
    --last;
    const uint8_t vs = a.sym[last];
    const uint32_t vk = a.key[last];
    sort_move(a, last, first);
    heap_adjust(a, first, 0, last - first, vs, vk);
  }
}

// std::__introsort_loop with its recursion on an explicit stack, run by the
// whole wave: first, last, depth and sp are wave-uniform.  The unguarded Hoare
// partition (std::__unguarded_partition) is evaluated in parallel.  Serially it
// alternates "i up to the next element not before the pivot" / "j down to the
// next element not after the pivot" and swaps while i < j; elements it has
// swapped are never looked at again before the loop ends, so the t-th swap pairs
// the t-th such element from the left (llist[t]) with the t-th from the right
// (rlist[t]) of the ORIGINAL segment, for exactly the t with llist[t] < rlist[t]
// (a prefix: one list ascends, the other descends).  When the loop ends after T
// swaps, i stands on the next left candidate or on the last swapped right
// position, whichever comes first.
__device__ void introsort_loop(const SortArrays& a, int n, int* stk, uint32_t* llist,
                               uint32_t* rlist, uint32_t lane) {
  int depth = 0;
  for (int v = n; v > 1; v >>= 1) ++depth;
  depth *= 2;
  int sp = 0, first = 0, last = n;
  const uint64_t below_me = (1ull << lane) - 1;
  for (;;) {
    while (last - first > 16) {
      if (depth == 0) {
        if (lane == 0) heap_sort_range(a, first, last);
        wave_sync();
        break;
      }
      --depth;
      // __move_median_to_first(first, first+1, mid, last-1)
      const int x = first + 1, y = first + (last - first) / 2, z = last - 1;
      const uint32_t kx = a.key[x], ky = a.key[y], kz = a.key[z];
      int m;
      if (kx > ky) {
        m = ky > kz ? y : (kx > kz ? z : x);
      } else {
        m = kx > kz ? x : (ky > kz ? z : y);
      }
      m = __builtin_amdgcn_readfirstlane(m);
      wave_sync();
      if (lane == 0) sort_swap(a, first, m);
      wave_sync();
      const uint32_t pk = __builtin_amdgcn_readfirstlane(a.key[first]);
      // __unguarded_partition(first+1, last, first)
      const int base = first + 1;
      const int len = last - base;  // 16..255
      uint64_t lb[4] = {0, 0, 0, 0}, rb[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (64 * q < len) {
          const int p = base + 64 * q + (int)lane;
          const bool in = p < last;
          const uint32_t k = in ? a.key[p] : 0;
          lb[q] = __ballot(in && k <= pk);  // not before the pivot: stops i
          rb[q] = __ballot(in && k >= pk);  // not after the pivot: stops j
        }
      }
      uint32_t lpre = 0;
      uint32_t rabove = __popcll(rb[0]) + __popcll(rb[1]) + __popcll(rb[2]) + __popcll(rb[3]);
      const uint32_t n_r = rabove;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (64 * q < len) {
          const uint32_t p = (uint32_t)(base + 64 * q) + lane;
          rabove -= __popcll(rb[q]);  // candidates in the chunks above this one
          if ((lb[q] >> lane) & 1) llist[lpre + __popcll(lb[q] & below_me)] = p;
          if ((rb[q] >> lane) & 1) rlist[rabove + __popcll(rb[q] & ~below_me) - 1] = p;
          lpre += __popcll(lb[q]);
        }
      }
      const uint32_t n_l = lpre;
      wave_sync();
      const uint32_t n_min = n_l < n_r ? n_l : n_r;
      uint32_t T = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if ((uint32_t)(64 * q) < n_min) {
          const uint32_t t = 64 * q + lane;
          const bool c = t < n_min && llist[t] < rlist[t];
          T += __popcll(__ballot(c));
        }
      }
      uint32_t cut = 0xffffffffu;
      if (T < n_l) cut = llist[T];
      if (T > 0) {
        const uint32_t r = rlist[T - 1];
        cut = r < cut ? r : cut;
      }
      cut = __builtin_amdgcn_readfirstlane(cut);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if ((uint32_t)(64 * q) < T) {
          const uint32_t t = 64 * q + lane;
          if (t < T) sort_swap(a, (int)llist[t], (int)rlist[t]);
        }
      }
      wave_sync();
      // recurse on [cut, last) with the reduced depth, continue with [first, cut)
      if (lane == 0) {
        stk[3 * sp] = (int)cut;
        stk[3 * sp + 1] = last;
        stk[3 * sp + 2] = depth;
      }
      ++sp;
      last = (int)cut;
    }
    if (sp == 0) return;
    --sp;
    wave_sync();
    first = __builtin_amdgcn_readfirstlane(stk[3 * sp]);
    last = __builtin_amdgcn_readfirstlane(stk[3 * sp + 1]);
    depth = __builtin_amdgcn_readfirstlane(stk[3 * sp + 2]);
  }
}

// Register r (wave-uniform) of four; used where r changes rarely.
__device__ __forceinline__ uint32_t pick4(const uint32_t (&a)[4], int r) {
  return r <= 0 ? a[0] : r == 1 ? a[1] : r == 2 ? a[2] : a[3];
}
__device__ __forceinline__ void put4(uint32_t (&a)[4], int r, uint32_t v) {
  a[0] = r == 0 ? v : a[0];
  a[1] = r == 1 ? v : a[1];
  a[2] = r == 2 ? v : a[2];
  a[3] = r == 3 ? v : a[3];
}

// Builds the code of one frame from its histogram hist[256] (global or LDS)
// and writes the FrameCode record; returns the LDS copy of the enc table,
// (code << 4) | len.  Runs on ONE wave (the barriers inside are wave-level), so
// a larger workgroup calls it from a single wave and synchronises afterwards.
__device__ __forceinline__ const uint16_t* build_code(const uint32_t* hist, FrameCode* __restrict__ out_p,
                                                      uint32_t lane) {
  __shared__ uint32_t key[256], key2[256], tcount[256], lcount[256];  // tcount: sort scratch
  __shared__ uint8_t sym[256], sym2[256];
  __shared__ uint16_t parent[2][256], depthv[2][256], lparent[256], encs[256];
  __shared__ uint32_t hdrw[70];
  __shared__ int stk[3 * 40];
  STAMP_DECL();

  // symbols with a nonzero count, ascending (:342-347)
  const uint4 h4 = *(const uint4*)(hist + 4 * lane);
  const uint32_t hv[4] = {h4.x, h4.y, h4.z, h4.w};
  uint32_t nz = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) nz += hv[j] != 0;
  const uint32_t incl = wave_inclusive_scan(nz, lane);
  // n is the same in every lane; saying so keeps everything derived from it in scalar registers
  const int n = __builtin_amdgcn_readfirstlane((int)__shfl(incl, 63, 64));
  uint32_t pos = incl - nz;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    lcount[4 * lane + j] = 0;
    encs[4 * lane + j] = 0;
    if (hv[j] != 0) {
      sym[pos] = (uint8_t)(4 * lane + j);
      key[pos] = hv[j];
      ++pos;
    }
  }
  wave_sync();

  STAMP(0);  // compaction
  const SortArrays arr = {sym, key};
  if (n > 16) introsort_loop(arr, n, stk, tcount, key2, lane);  // both lists are free until later
  wave_sync();
  STAMP(1);  // partition loop
  // Stable order by descending count = what the final insertion sort leaves.
  // The partition loop leaves runs of at most 16 elements that are already in
  // order among each other, so an element moves fewer than 16 places and only
  // its 15 neighbours on either side decide its rank.
  for (int i = (int)lane; i < n; i += 64) {
    const uint32_t k = key[i];
    int rank = i > 15 ? i - 15 : 0;
#pragma unroll
    for (int d = -15; d <= 15; ++d) {
      const int m = i + d;
      const bool in = m >= 0 && m < n;
      const int mc = m < 0 ? 0 : (m >= n ? n - 1 : m);  // always a valid index: no branch around the read
      const uint32_t km = key[mc];
      rank += in && (km > k || (km == k && d < 0));
    }
    sym2[rank] = sym[i];
    key2[rank] = k;
  }
  wave_sync();

  STAMP(2);  // rank sort
  // Two-queue Huffman merge (:364-416); records each leaf's and node's parent.
  // Inherently serial, so it runs as wave-uniform scalar code with both queues in
  // registers: element i is lane i & 63 of register i >> 6.  Each queue is walked
  // one way, so the register that holds its head changes every 64 elements only;
  // a copy of that register is kept at hand and a step needs one v_readlane per
  // element taken and one lane-select per node made, no LDS round trip.
  //   leaves: kq[], head ns going down, kcur = kq[ns >> 6]
  //   nodes:  made at ts, taken at nt, both going up; tw collects block ts >> 6
  //           and moves to tq[] when it is full; tr = block nt >> 6 once full
  uint32_t kq[4], tq[4] = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) kq[r] = key2[64 * r + lane];
  int ts = 0;
  if (n <= 64) {
    // up to 64 symbols (any skewed source): one register per queue, no blocks
    int ns = n - 1, nt = 0;
    uint32_t tw = 0;
    uint32_t lk = __builtin_amdgcn_readlane(kq[0], ns & 63), nk = 0;
    while ((ts - nt) + (ns + 1) > 1) {
      uint32_t sum = 0;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bool leaf = ns >= 0 && (nt == ts || lk <= nk);
        if (leaf) {
          sum += lk;
          lparent[ns] = (uint16_t)ts;
          --ns;
          lk = __builtin_amdgcn_readlane(kq[0], ns & 63);
        } else {
          sum += nk;
          parent[0][nt] = (uint16_t)ts;
          ++nt;
          nk = __builtin_amdgcn_readlane(tw, nt & 63);
        }
      }
      tw = lane == (uint32_t)ts ? sum : tw;
      if (nt == ts) nk = sum;
      ++ts;
    }
    if (ts > 0) parent[0][ts - 1] = (uint16_t)(ts - 1);
  } else {
    int ns = n - 1, nt = 0;
    uint32_t kcur = pick4(kq, ns >> 6), tw = 0, tr = 0;
    uint32_t lk = __builtin_amdgcn_readlane(kcur, ns & 63);  // head of the leaf queue
    uint32_t nk = 0;                                         // head of the node queue (valid if nt < ts)
    while ((ts - nt) + (ns + 1) > 1) {
      uint32_t sum = 0;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bool leaf = ns >= 0 && (nt == ts || lk <= nk);  // a leaf wins a tie (:375)
        if (leaf) {
          sum += lk;
          lparent[ns] = (uint16_t)ts;  // the same store from every lane
          --ns;
          if ((ns & 63) == 63 && ns >= 0) kcur = pick4(kq, ns >> 6);
          lk = __builtin_amdgcn_readlane(kcur, ns & 63);  // unused once ns < 0
        } else {
          sum += nk;
          parent[0][nt] = (uint16_t)ts;
          ++nt;
          if ((nt & 63) == 0) tr = pick4(tq, nt >> 6);  // stale if that block is still tw: not read then
          nk = __builtin_amdgcn_readlane((nt >> 6) == (ts >> 6) ? tw : tr, nt & 63);  // unused if nt == ts
        }
      }
      tw = lane == (uint32_t)(ts & 63) ? sum : tw;
      if (nt == ts) nk = sum;  // the queue was empty: the new node is its head
      ++ts;
      if ((ts & 63) == 0) {  // block full (at most three times)
        const int b = (ts >> 6) - 1;
        put4(tq, b, tw);
        if ((nt >> 6) == b) tr = tw;
        tw = 0;
      }
    }
    if (ts > 0) parent[0][ts - 1] = (uint16_t)(ts - 1);  // root points to itself
  }
  wave_sync();
  STAMP(3);  // merge
  if (ts == 0) {
    if (lane == 0 && n == 1) lcount[0] = 1;  // a single symbol: zero-length code
  } else {
    // depth of every internal node by pointer jumping (root = ts-1, depth 0)
    for (int i = (int)lane; i < ts; i += 64) depthv[0][i] = i == ts - 1 ? 0 : 1;
    wave_sync();
    int cur = 0;
    for (int it = 0; it < 8; ++it) {
      for (int i = (int)lane; i < ts; i += 64) {
        const int p = parent[cur][i];
        depthv[cur ^ 1][i] = (uint16_t)(depthv[cur][i] + depthv[cur][p]);
        parent[cur ^ 1][i] = parent[cur][p];
      }
      wave_sync();
      cur ^= 1;
    }
    for (int i = (int)lane; i < n; i += 64)
      atomicAdd(&lcount[depthv[cur][lparent[i]] + 1], 1u);
  }
  wave_sync();

  STAMP(4);  // depths
  // LimitCodeLengths (:297-327) on wave-uniform registers: lengths beyond 12
  // fold into 12, then MiniZ's fix-up loop
  uint32_t lc[kMaxCodeLen + 1];
  {
    uint32_t over = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * lane + j > (uint32_t)kMaxCodeLen) over += lcount[4 * lane + j];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) over += __shfl_xor(over, d, 64);
#pragma unroll
    for (int L = 0; L <= kMaxCodeLen; ++L) lc[L] = __builtin_amdgcn_readfirstlane(lcount[L]);
    lc[kMaxCodeLen] += __builtin_amdgcn_readfirstlane(over);
    uint32_t kraft = 0;
#pragma unroll
    for (int L = 0; L <= kMaxCodeLen; ++L) kraft += lc[L] << (kMaxCodeLen - L);
    // One round of the fix-up loop takes a code of the longest length j < 12 in
    // use, lengthens it, and drops the Kraft sum by one.  While length 11 is in
    // use every round does the same thing (lc[11] - 1, lc[12] - 1 + 2), so those
    // rounds are applied together; any other j is a round of its own and leaves
    // two codes at j + 1.
    while (kraft > (1u << kMaxCodeLen)) {
      const uint32_t excess = kraft - (1u << kMaxCodeLen);
      if (lc[kMaxCodeLen - 1] > 0) {
        const uint32_t t = lc[kMaxCodeLen - 1] < excess ? lc[kMaxCodeLen - 1] : excess;
        lc[kMaxCodeLen - 1] -= t;
        lc[kMaxCodeLen] += t;
        kraft -= t;
        continue;
      }
      --lc[kMaxCodeLen];
      bool done = false;
#pragma unroll
      for (int j = kMaxCodeLen - 2; j >= 0; --j) {
        if (!done && lc[j] > 0) {
          --lc[j];
          lc[j + 1] += 2;
          done = true;
        }
      }
      --kraft;
    }
  }
  STAMP(5);  // length limit
  // canonical codes in sorted order (ForallCodes, :260-284) and header bytes
  uint32_t len_mask = 0, nlen = 0;
#pragma unroll
  for (int L = 0; L <= kMaxCodeLen; ++L) {
    if (lc[L]) {
      len_mask |= 1u << L;
      ++nlen;
    }
  }
  for (int i = (int)lane; i < n; i += 64) {
    uint32_t start = 0, codev = 0, L = 0, base = 0, fc = 0;
#pragma unroll
    for (int l = 0; l <= kMaxCodeLen; ++l) {
      if ((uint32_t)i >= start && lc[l] != 0) {
        L = l;
        base = start;
        fc = codev;
      }
      start += lc[l];
      codev += lc[l] << (kMaxCodeLen - l);
    }
    const uint32_t c12 = fc + (((uint32_t)i - base) << (kMaxCodeLen - L));
    encs[sym2[i]] = (uint16_t)((c12 << 4) | L);
  }
  uint8_t* hb = (uint8_t*)hdrw;
  for (uint32_t i = lane; i < 70; i += 64) hdrw[i] = 0;
  wave_sync();
  if (lane == 0) {
    hdrw[0] = len_mask;
    uint32_t p = 4;
#pragma unroll
    for (int L = 0; L <= kMaxCodeLen; ++L)
      if (lc[L]) hb[p++] = (uint8_t)lc[L];  // 256 is stored as 0x00 (:801-806)
  }
  for (int i = (int)lane; i < n; i += 64) hb[4 + nlen + i] = sym2[i];
  wave_sync();

  STAMP(6);  // canonical codes, header
  FrameCode& out = *out_p;
  uint32_t* enc_out = (uint32_t*)out.enc;
  const uint32_t* encs32 = (const uint32_t*)encs;
  for (uint32_t i = lane; i < 128; i += 64) enc_out[i] = encs32[i];
  uint32_t* hdr_out = (uint32_t*)out.hdr;
  for (uint32_t i = lane; i < 69; i += 64) hdr_out[i] = hdrw[i];
  if (lane == 0) out.hdr_len = 4 + nlen + (uint32_t)n;
  STAMP(7);  // write-out
  STAMP_FLUSH(16);
  return encs;
}

__global__ __launch_bounds__(64) void code_kernel(const uint32_t* __restrict__ hist,
                                                  FrameCode* __restrict__ code) {
  (void)build_code(hist + (uint64_t)blockIdx.x * 256, code + blockIdx.x, threadIdx.x);
}

// ---------------------------------------------------------------------------
// K2: the frame's code (see build_code) and the coded size of each of its
// streams (codec/huffman.cpp:770-786): size_k = ceil(sum of code lengths over
// slice k / 8) + 8, then prefix sums.  Lane k walks slice k in aligned 32-byte
// blocks, four blocks per group with the next group already in flight.
constexpr uint32_t kSzGroup = 4;

__device__ __forceinline__ uint32_t sum_code_len(const uint8_t* __restrict__ ibase, uint32_t in_lo,
                                                 uint32_t in_hi, const uint8_t* lens) {
  if (in_hi <= in_lo) return 0;
  const uint32_t first_blk = in_lo & ~31u;
  const uint32_t nblk = (in_hi + 31) / 32 - in_lo / 32;
  const uint32_t last_blk = first_blk + 32 * (nblk - 1);
  const uint32_t ngroups = (nblk + kSzGroup - 1) / kSzGroup;
  uint4 cur[2 * kSzGroup], nxt[2 * kSzGroup];
#define HFZ_LOAD_GROUP(dst, g)                                        \
  do {                                                                \
    _Pragma("unroll") for (uint32_t q_ = 0; q_ < kSzGroup; ++q_) {    \
      uint32_t a_ = first_blk + 32 * ((g) * kSzGroup + q_);           \
      a_ = a_ < last_blk ? a_ : last_blk;                             \
      dst[2 * q_] = *(const uint4*)(ibase + a_);                      \
      dst[2 * q_ + 1] = *(const uint4*)(ibase + a_ + 16);             \
    }                                                                 \
  } while (0)
  uint32_t bits = 0;
  HFZ_LOAD_GROUP(nxt, 0);
  for (uint32_t g = 0; g < ngroups; ++g) {
#pragma unroll
    for (uint32_t q = 0; q < 2 * kSzGroup; ++q) cur[q] = nxt[q];
    if (g + 1 < ngroups) HFZ_LOAD_GROUP(nxt, g + 1);
#pragma unroll
    for (uint32_t q = 0; q < kSzGroup; ++q) {
      const uint32_t t = g * kSzGroup + q;
      if (t >= nblk) break;
      const uint32_t w[8] = {cur[2 * q].x,     cur[2 * q].y,     cur[2 * q].z,     cur[2 * q].w,
                             cur[2 * q + 1].x, cur[2 * q + 1].y, cur[2 * q + 1].z, cur[2 * q + 1].w};
      const uint32_t b = first_blk + 32 * t;
      if (b >= in_lo && b + 32 <= in_hi) {
#pragma unroll
        for (uint32_t d = 0; d < 8; ++d) {
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j) bits += lens[(w[d] >> (8 * j)) & 0xff];
        }
      } else {
#pragma unroll
        for (uint32_t d = 0; d < 8; ++d) {
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t p = b + 4 * d + j;
            if (p >= in_lo && p < in_hi) bits += lens[(w[d] >> (8 * j)) & 0xff];
          }
        }
      }
    }
  }
#undef HFZ_LOAD_GROUP
  return bits;
}

// build != 0: the kernel first derives the frame's code from hist[f] and writes
// code[f]; build == 0: code[f * code_step] is given (code_step = 0: one code for
// every frame).
__global__ __launch_bounds__(64) void sizes_kernel(uint32_t K, const uint8_t* __restrict__ raw,
                                                   uint64_t total, uint64_t frame_bytes,
                                                   const uint32_t* __restrict__ hist,
                                                   FrameCode* __restrict__ code, uint32_t build,
                                                   uint32_t code_step,
                                                   uint32_t* __restrict__ stream_end,
                                                   uint32_t* __restrict__ csize) {
  __shared__ uint8_t lens[256];
  const uint64_t f = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  uint32_t hdr_len;
  if (build) {
    const uint16_t* encs = build_code(hist + f * 256, code + f, lane);
    for (uint32_t i = lane; i < 256; i += 64) lens[i] = (uint8_t)(encs[i] & 15);
    __syncthreads();
    hdr_len = code[f].hdr_len;  // written by lane 0 above; same wave, after the barrier
  } else {
    const FrameCode& fc = code[f * code_step];
    for (uint32_t i = lane; i < 256; i += 64) lens[i] = (uint8_t)(fc.enc[i] & 15);
    __syncthreads();
    hdr_len = fc.hdr_len;
  }
  const uint32_t n = frame_len(total, frame_bytes, f);
  uint32_t bytes = 0;
  if (lane < K) {
    const Slice s = slice_of(n, K, lane);
    const uint8_t* const frame_in = raw + f * frame_bytes;
    const uint8_t* const ibase = frame_in - ((uintptr_t)frame_in & 31);
    const uint32_t in_lo = (uint32_t)((uintptr_t)frame_in - (uintptr_t)ibase) + s.begin;
    const uint32_t bits = sum_code_len(ibase, in_lo, in_lo + s.size, lens);
    bytes = (bits + 7) / 8 + kSlop;
  }
  const uint32_t incl = wave_inclusive_scan(bytes, lane);
  if (lane < K) stream_end[f * 64 + lane] = incl;
  if (lane == K - 1) csize[f] = 4 + hdr_len + 4 * (K - 1) + incl;
}

// ---------------------------------------------------------------------------
// K1+K2 in one pass over the input: per-stream byte counts in LDS, from which
// come the frame histogram (codec/histogram.cpp:193-201), the canonical code
// (build_code) and the coded size of every stream as
// sum over symbols of count_k[sym] * len[sym] (codec/huffman.cpp:770-786) -- the
// input is read once and no per-byte code-length lookup is needed.
//
// One workgroup of 256 threads per frame.  Thread t counts part t / K of
// stream t % K.  The counters form 256 rows (one per byte value) of 32 words:
//   K > 32 : word = stream % 32, 16-bit count number stream / 32 in it (slices of
//            at most 65535 bytes; the launcher falls back to hist_kernel +
//            sizes_kernel beyond that)
//   K <= 32: word = stream + K * (part % (32 / K)), a 32-bit count
// Lanes l and l + 32 of a wave share a word (or, with K <= 32, at least a
// bank) and all other lanes are on other banks: each half of an LDS atomic
// hits 32 different banks whatever the data looks like.  The kernel ends up
// limited by its reads (every lane walks its own kilobyte: about 3.6 TB/s).
constexpr uint32_t kCntThreads = 256;
constexpr uint32_t kCntBlk = 128;  // bytes per lane per load group (the next group is in flight)

// The lanes of a wave walk streams n/K bytes apart, a power of two for the
// usual frame sizes, so if they all stood at the same offset of their streams
// every load would send its 64 lines to the same few memory channels (counting
// 2 GiB in 1 MiB frames: 0.54 ms that way, 0.42 with streams 16.25 KiB apart).
// Counting does not care about order, so every lane starts at a block of its
// own (`rot`) and wraps around.
__device__ __forceinline__ void count_range(const uint8_t* __restrict__ ibase, uint32_t lo,
                                            uint32_t hi, uint32_t rot, uint32_t* col, uint32_t inc) {
  if (hi <= lo) return;
  constexpr uint32_t kQ = kCntBlk / 16;  // 16-byte loads per block
  const uint32_t first_blk = lo & ~(kCntBlk - 1);
  const uint32_t nblk = (hi + kCntBlk - 1) / kCntBlk - lo / kCntBlk;
  uint32_t cur_t = rot % nblk;  // block being counted; the next one is in flight
  uint4 cur[kQ], nxt[kQ];
#pragma unroll
  for (uint32_t q = 0; q < kQ; ++q) nxt[q] = *(const uint4*)(ibase + first_blk + kCntBlk * cur_t + 16 * q);
  for (uint32_t t = 0; t < nblk; ++t) {
    const uint32_t b = first_blk + kCntBlk * cur_t;
#pragma unroll
    for (uint32_t q = 0; q < kQ; ++q) cur[q] = nxt[q];
    cur_t = cur_t + 1 == nblk ? 0 : cur_t + 1;
    if (t + 1 < nblk) {
#pragma unroll
      for (uint32_t q = 0; q < kQ; ++q)
        nxt[q] = *(const uint4*)(ibase + first_blk + kCntBlk * cur_t + 16 * q);
    }
    if (b >= lo && b + kCntBlk <= hi) {
#pragma unroll
      for (uint32_t q = 0; q < kQ; ++q) {
        const uint32_t w[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j) atomicAdd(&col[((w[d] >> (8 * j)) & 0xff) * 32], inc);
        }
      }
    } else {
#pragma unroll
      for (uint32_t q = 0; q < kQ; ++q) {
        const uint32_t w[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t p = b + 16 * q + 4 * d + j;
            if (p >= lo && p < hi) atomicAdd(&col[((w[d] >> (8 * j)) & 0xff) * 32], inc);
          }
        }
      }
    }
  }
}

// counts_out != null: the counters and the frame histogram go to memory and
// code_sizes_kernel finishes the job (the code construction has long serial
// stretches, which want many frames per CU in flight, and this kernel's 33 KiB
// of LDS allows only four).
// counts_out == null: code[f * code_step] is given (code_step = 0: one code for
// every frame) and the stream sizes
// are computed right here.  hist_out may be null then.
// split_log2 > 0 (counts_out only): 2^split_log2 workgroups per frame; workgroup sp
// of frame f counts piece sp of every stream, the pieces cut like SliceSizes cuts
// a frame (the cut of the several-threads-per-stream encoder), and writes counters
// and histogram of unit f * 2^split_log2 + sp.  Few-frame batches get more
// workgroups that way, and the encoder learns where every piece's bits begin
// without adding up code lengths in a pass of its own.
__global__ __launch_bounds__(kCntThreads) void count_kernel(
    uint32_t K, const uint8_t* __restrict__ raw, uint64_t total, uint64_t frame_bytes,
    const FrameCode* __restrict__ code, uint32_t code_step, uint32_t* __restrict__ counts_out,
    uint32_t* __restrict__ hist_out, uint32_t* __restrict__ stream_end,
    uint32_t* __restrict__ csize, uint32_t split_log2) {
  __shared__ __align__(16) uint32_t cnt[256 * 32];
  __shared__ uint32_t lens4[64];  // code length of symbol s in byte s & 3 of word s >> 2
  __shared__ uint32_t part_bits[kCntThreads / 64][64];
  const uint64_t unit = blockIdx.x;  // where counters and histogram go
  const uint64_t f = unit >> split_log2;
  const uint32_t split_piece = (uint32_t)(unit & ((1u << split_log2) - 1));
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint32_t i = tid; i < 256 * 32; i += kCntThreads) cnt[i] = 0;
  __syncthreads();

  const uint32_t n = frame_len(total, frame_bytes, f);
  const bool packed = K > 32;
  const uint32_t parts = kCntThreads / K;  // >= 4
  const uint32_t reps = packed ? 1 : 32 / K;
  {
    const uint32_t stream = tid % K, part = tid / K;
    if (part < parts) {
      Slice s = slice_of(n, K, stream);
      if (split_log2) {  // this workgroup's piece of the stream
        const Slice ps = slice_of(s.size, 1u << split_log2, split_piece);
        s.begin += ps.begin;
        s.size = ps.size;
      }
      const uint32_t chunk = ((s.size + parts - 1) / parts + kCntBlk - 1) & ~(kCntBlk - 1);
      const uint32_t rel_lo = part * chunk < s.size ? part * chunk : s.size;
      const uint32_t rel_hi = rel_lo + chunk < s.size ? rel_lo + chunk : s.size;
      const uint8_t* const frame_in = raw + f * frame_bytes;
      const uint8_t* const ibase = frame_in - ((uintptr_t)frame_in & (kCntBlk - 1));
      const uint32_t in0 = (uint32_t)((uintptr_t)frame_in - (uintptr_t)ibase) + s.begin;
      const uint32_t colidx = packed ? stream & 31 : stream + K * (part % reps);
      const uint32_t inc = packed ? 1u << (16 * (stream >> 5)) : 1u;
      count_range(ibase, in0 + rel_lo, in0 + rel_hi, 2 * stream + part, cnt + colidx, inc);
    }
  }
  __syncthreads();

  // frame histogram: thread t sums row t, starting at a different word per
  // thread so that the 64 lanes read 32 different banks
  uint32_t sum = 0;
#pragma unroll 8
  for (uint32_t c = 0; c < 32; ++c) {
    const uint32_t w = cnt[tid * 32 + ((c + tid) & 31)];
    sum += packed ? (w & 0xffff) + (w >> 16) : w;
  }
  if (hist_out) hist_out[unit * 256 + tid] = sum;
  if (counts_out) {
    // Only the rows of byte values that occur go to memory, packed in ascending
    // order of the value (a fifth of the rows on the reference's biased source);
    // code_sizes_kernel finds out which they are from the histogram.
    const bool occurs = sum != 0;
    const uint64_t bal = __ballot(occurs);
    if (lane == 0) part_bits[wave][0] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t at = (uint32_t)__popcll(bal & ((1ull << lane) - 1)), n_rows = 0;
#pragma unroll
    for (uint32_t w = 0; w < kCntThreads / 64; ++w) {
      const uint32_t c = part_bits[w][0];
      if (w < wave) at += c;
      n_rows += c;
    }
    uint8_t* const row_of = (uint8_t*)lens4;  // 256 bytes
    if (occurs) row_of[at] = (uint8_t)tid;
    __syncthreads();
    uint4* dst = (uint4*)(counts_out + unit * (256 * 32));
    const uint4* src = (const uint4*)cnt;
    for (uint32_t i = tid; i < 8 * n_rows; i += kCntThreads) dst[i] = src[8 * (uint32_t)row_of[i >> 3] + (i & 7)];
    return;
  }
  if (wave == 0) {
    const uint16_t* enc = code[f * code_step].enc;
    lens4[lane] = (uint32_t)(enc[4 * lane] & 15) | (uint32_t)(enc[4 * lane + 1] & 15) << 8 |
                  (uint32_t)(enc[4 * lane + 2] & 15) << 16 | (uint32_t)(enc[4 * lane + 3] & 15) << 24;
  }
  __syncthreads();

  // coded bits of stream `lane`: each wave covers a quarter of the symbols
  {
    uint32_t bits = 0;
    if (lane < K) {
      const uint32_t sh = packed ? 16 * (lane >> 5) : 0;
      const uint32_t c0 = packed ? lane & 31 : lane;
      for (uint32_t q = 0; q < 16; ++q) {
        const uint32_t sym0 = wave * 64 + 4 * q;
        const uint32_t l4 = lens4[sym0 >> 2];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
          uint32_t c;
          if (packed) {
            c = (cnt[(sym0 + j) * 32 + c0] >> sh) & 0xffff;
          } else {
            c = 0;
            for (uint32_t r = 0; r < reps; ++r) c += cnt[(sym0 + j) * 32 + c0 + K * r];
          }
          bits += c * ((l4 >> (8 * j)) & 0xff);
        }
      }
    }
    part_bits[wave][lane] = bits;
  }
  __syncthreads();
  if (wave == 0) {
    uint32_t bytes = 0;
    if (lane < K) {
      uint32_t bits = 0;
#pragma unroll
      for (uint32_t q = 0; q < kCntThreads / 64; ++q) bits += part_bits[q][lane];
      bytes = (bits + 7) / 8 + kSlop;
    }
    const uint32_t incl = wave_inclusive_scan(bytes, lane);
    if (lane < K) stream_end[f * 64 + lane] = incl;
    if (lane == K - 1) csize[f] = 4 + code[f * code_step].hdr_len + 4 * (K - 1) + incl;
  }
}

// Second half of the counting path, one wave per frame: the frame's code from
// its histogram (build_code), then the coded size of every stream from the
// per-stream counts count_kernel left in memory: the rows of its LDS array for
// the byte values that occur (nonzero histogram bins), packed in ascending order.
// Lane l walks the packed rows of parity l / 32 and owns counter word l % 32.
// given != null: no code is built; the sizes are those under given[f * given_step].
__device__ __forceinline__ void code_sizes_body(uint32_t K, const uint32_t* __restrict__ hist,
                                                const uint32_t* __restrict__ counts,
                                                FrameCode* __restrict__ code,
                                                const FrameCode* __restrict__ given,
                                                uint32_t given_step,
                                                uint32_t* __restrict__ stream_end,
                                                uint32_t* __restrict__ csize) {
  __shared__ uint32_t lens[256];
  __shared__ uint32_t row_len[256];  // code length of the byte value of packed row j
  __shared__ uint32_t sums[64];
  const uint64_t f = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const uint32_t* rows = counts + f * (256 * 32) + lane;  // word (packed row l/32, column l%32)
  // the counters do not depend on the code: start fetching them now (rows past
  // the last packed one hold stale data and are skipped below)
  constexpr uint32_t kAhead = 8;
  uint32_t nxt[kAhead];
#pragma unroll
  for (uint32_t i = 0; i < kAhead; ++i) nxt[i] = rows[64 * i];
  // which byte values occur: lane l looks at values 4l .. 4l+3
  const uint4 h4 = *(const uint4*)(hist + f * 256 + 4 * lane);
  const uint32_t hv[4] = {h4.x, h4.y, h4.z, h4.w};
  uint32_t nz = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) nz += hv[j] != 0;
  const uint32_t incl = wave_inclusive_scan(nz, lane);
  const uint32_t n_rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(incl, 63, 64));
  uint32_t hdr_len;
  if (given) {
    const FrameCode& fc = given[f * given_step];
    for (uint32_t i = lane; i < 256; i += 64) lens[i] = fc.enc[i] & 15;
    hdr_len = fc.hdr_len;
  } else {
    const uint16_t* encs = build_code(hist + f * 256, code + f, lane);
    for (uint32_t i = lane; i < 256; i += 64) lens[i] = encs[i] & 15;
    hdr_len = 0;  // read back below, after the barrier
  }
  sums[lane] = 0;
  wave_sync();
  {
    uint32_t at = incl - nz;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (hv[j] != 0) row_len[at++] = lens[4 * lane + j];
  }
  wave_sync();
  if (!given) hdr_len = code[f].hdr_len;  // written by lane 0 of this wave inside build_code
  const bool packed = K > 32;
  const uint32_t reps = packed ? 1 : 32 / K;
  const uint32_t col = lane & 31, par = lane >> 5;
  uint32_t acc_lo = 0, acc_hi = 0;
  const uint32_t n_groups = ((n_rows + 1) / 2 + kAhead - 1) / kAhead;  // kAhead row pairs each
  for (uint32_t g = 0; g < n_groups; ++g) {
    uint32_t cur[kAhead];
#pragma unroll
    for (uint32_t i = 0; i < kAhead; ++i) cur[i] = nxt[i];
    if (g + 1 < n_groups) {
#pragma unroll
      for (uint32_t i = 0; i < kAhead; ++i) nxt[i] = rows[64 * (kAhead * (g + 1) + i)];
    }
#pragma unroll
    for (uint32_t i = 0; i < kAhead; ++i) {
      const uint32_t r = 2 * (kAhead * g + i) + par;
      const uint32_t len = r < n_rows ? row_len[r] : 0;
      const uint32_t c = r < n_rows ? cur[i] : 0;
      if (packed) {
        acc_lo += (c & 0xffff) * len;
        acc_hi += (c >> 16) * len;
      } else {
        acc_lo += c * len;
      }
    }
  }
  if (packed) {
    atomicAdd(&sums[col], acc_lo);
    atomicAdd(&sums[col + 32], acc_hi);
  } else if (col < K * reps) {
    atomicAdd(&sums[col % K], acc_lo);
  }
  wave_sync();
  uint32_t bytes = 0;
  if (lane < K) bytes = (sums[lane] + 7) / 8 + kSlop;
  const uint32_t incl_b = wave_inclusive_scan(bytes, lane);
  if (lane < K) stream_end[f * 64 + lane] = incl_b;
  if (lane == K - 1) csize[f] = 4 + hdr_len + 4 * (K - 1) + incl_b;
}

// Two builds of the same body.  The code construction is long chains of serial
// work per frame, so what counts is how many frames a SIMD holds: four per SIMD
// (128 registers, a few spills) for batches that need the room, two (all the
// registers the body wants) while two per SIMD hold the whole batch.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void code_sizes_kernel(
    uint32_t K, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ counts,
    FrameCode* __restrict__ code, const FrameCode* __restrict__ given, uint32_t given_step,
    uint32_t* __restrict__ stream_end, uint32_t* __restrict__ csize) {
  code_sizes_body(K, hist, counts, code, given, given_step, stream_end, csize);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void code_sizes_kernel_wide(
    uint32_t K, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ counts,
    FrameCode* __restrict__ code, const FrameCode* __restrict__ given, uint32_t given_step,
    uint32_t* __restrict__ stream_end, uint32_t* __restrict__ csize) {
  code_sizes_body(K, hist, counts, code, given, given_step, stream_end, csize);
}

// The same for a batch counted in pieces (count_kernel with split_log2 > 0): the
// frame's histogram is the sum of its units', the code is built from that, and
// the coded bits of piece sp of stream k come from unit f * P + sp's counters.
// Their running sums are the first bit of every piece, which go to
// part_first[(f * 64 + k) * 8 + sp] for the several-threads-per-stream encoder
// (P = 2^split_log2 <= 8 pieces, the encoder's parts), and the last one is the
// stream's size.  One wave per piece: all of them fetch their unit's histogram
// and the first of its counters, wave 0 builds the code while those loads are in
// flight, then every wave weighs its unit's counters and wave 0 adds up.
constexpr uint32_t kMaxSplit = 1u << kMaxSplitLog2;
__global__ __launch_bounds__(64 * kMaxSplit) void code_sizes_split_kernel(
    uint32_t K, const uint32_t* __restrict__ hist_u, const uint32_t* __restrict__ counts_u,
    FrameCode* __restrict__ code, uint32_t split_log2, uint32_t* __restrict__ part_first,
    uint32_t* __restrict__ stream_end, uint32_t* __restrict__ csize) {
  __shared__ uint32_t lens[256];
  __shared__ __align__(16) uint32_t hu[kMaxSplit][256];  // the units' histograms
  __shared__ __align__(16) uint32_t hsum[256];           // the frame's
  __shared__ uint32_t row_len[kMaxSplit][256];  // code length of the byte value of packed row j of a unit
  __shared__ uint32_t sums[kMaxSplit][64];      // coded bits of (piece, stream)
  const uint64_t f = blockIdx.x;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t P = 1u << split_log2;  // = waves of this workgroup
  const uint64_t u = f * P + wave;
  const uint32_t* rows = counts_u + u * (256 * 32) + lane;  // word (packed row l/32, column l%32)
  // the counters do not depend on the code: start fetching them now (row pairs past
  // the last packed one hold stale data and are skipped below)
  constexpr uint32_t kAhead = 8;
  uint32_t nxt[kAhead];
#pragma unroll
  for (uint32_t i = 0; i < kAhead; ++i) nxt[i] = rows[64 * i];
  // which byte values occur in this unit: lane l looks at values 4l .. 4l+3
  const uint4 h4 = *(const uint4*)(hist_u + u * 256 + 4 * lane);
  *(uint4*)(&hu[wave][4 * lane]) = h4;
  sums[wave][lane] = 0;
  __syncthreads();
  if (wave == 0) {
    uint4 a = make_uint4(0, 0, 0, 0);
    for (uint32_t sp = 0; sp < P; ++sp) {
      const uint4 h = *(const uint4*)(&hu[sp][4 * lane]);
      a.x += h.x;
      a.y += h.y;
      a.z += h.z;
      a.w += h.w;
    }
    *(uint4*)(hsum + 4 * lane) = a;
    wave_sync();
    const uint16_t* encs = build_code(hsum, code + f, lane);
    for (uint32_t i = lane; i < 256; i += 64) lens[i] = encs[i] & 15;
  }
  __syncthreads();
  const uint32_t hv[4] = {h4.x, h4.y, h4.z, h4.w};
  uint32_t nz = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) nz += hv[j] != 0;
  const uint32_t incl = wave_inclusive_scan(nz, lane);
  const uint32_t n_rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(incl, 63, 64));
  {
    uint32_t at = incl - nz;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (hv[j] != 0) row_len[wave][at++] = lens[4 * lane + j];
  }
  wave_sync();
  const bool packed = K > 32;
  const uint32_t reps = packed ? 1 : 32 / K;
  const uint32_t col = lane & 31, par = lane >> 5;
  uint32_t acc_lo = 0, acc_hi = 0;
  const uint32_t n_groups = ((n_rows + 1) / 2 + kAhead - 1) / kAhead;  // kAhead row pairs each
  for (uint32_t g = 0; g < n_groups; ++g) {
    uint32_t cur[kAhead];
#pragma unroll
    for (uint32_t i = 0; i < kAhead; ++i) cur[i] = nxt[i];
    if (g + 1 < n_groups) {
#pragma unroll
      for (uint32_t i = 0; i < kAhead; ++i) nxt[i] = rows[64 * (kAhead * (g + 1) + i)];
    }
#pragma unroll
    for (uint32_t i = 0; i < kAhead; ++i) {
      const uint32_t r = 2 * (kAhead * g + i) + par;
      const uint32_t len = r < n_rows ? row_len[wave][r] : 0;
      const uint32_t c = r < n_rows ? cur[i] : 0;
      if (packed) {
        acc_lo += (c & 0xffff) * len;
        acc_hi += (c >> 16) * len;
      } else {
        acc_lo += c * len;
      }
    }
  }
  if (packed) {
    atomicAdd(&sums[wave][col], acc_lo);
    atomicAdd(&sums[wave][col + 32], acc_hi);
  } else if (col < K * reps) {
    atomicAdd(&sums[wave][col % K], acc_lo);
  }
  __syncthreads();
  if (wave != 0) return;
  const uint32_t hdr_len = code[f].hdr_len;  // written by lane 0 of this wave inside build_code
  uint32_t run = 0;  // lane k: bits of stream k before the piece at hand
  if (lane < K) {
    for (uint32_t sp = 0; sp < P; ++sp) {
      part_first[(f * 64 + lane) * 8 + sp] = run;
      run += sums[sp][lane];
    }
  }
  uint32_t bytes = 0;
  if (lane < K) bytes = (run + 7) / 8 + kSlop;
  const uint32_t incl_b = wave_inclusive_scan(bytes, lane);
  if (lane < K) stream_end[f * 64 + lane] = incl_b;
  if (lane == K - 1) csize[f] = 4 + hdr_len + 4 * (K - 1) + incl_b;
}

// Sum of the per-frame histograms of a batch, 64-bit: total[c] = sum over f of hist[f][c].
__global__ __launch_bounds__(256) void hist_total_kernel(const uint32_t* __restrict__ hist,
                                                         uint64_t n_frames,
                                                         unsigned long long* __restrict__ total) {
  const uint32_t c = threadIdx.x;
  unsigned long long sum = 0;
  for (uint64_t f = blockIdx.x; f < n_frames; f += gridDim.x) sum += hist[f * 256 + c];
  if (sum) atomicAdd(&total[c], sum);
}

// ---------------------------------------------------------------------------
// Frame placement: exclusive prefix sum of the frame sizes (packed layout) or
// i*stride (slot layout).  One block; frames are few compared with bytes.
__global__ __launch_bounds__(1024) void offsets_kernel(const uint32_t* __restrict__ csize,
                                                       uint64_t n_frames, uint64_t stride,
                                                       uint64_t cap,
                                                       uint64_t* __restrict__ comp_off,
                                                       uint32_t* __restrict__ comp_size,
                                                       int32_t* __restrict__ status) {
  __shared__ uint64_t wave_tot[16];
  __shared__ uint64_t carry_s;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t base = 0; base < n_frames; base += 1024) {
    const uint64_t i = base + tid;
    const uint64_t v = i < n_frames ? csize[i] : 0;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint64_t o = __shfl_up(incl, d, 64);
      if (lane >= (uint32_t)d) incl += o;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    uint64_t before = carry_s;
    for (uint32_t w = 0; w < wid; ++w) before += wave_tot[w];
    if (i < n_frames) {
      const uint64_t off = stride ? i * stride : before + incl - v;
      const bool ok = off <= cap && v <= cap - off && (stride == 0 || v <= stride);
      comp_off[i] = off;
      comp_size[i] = (uint32_t)v;
      if (status) status[i] = ok ? HFZ_OK : HFZ_E_CAPACITY;
    }
    __syncthreads();
    if (tid == 1023) carry_s = before + incl;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// K3: encode.  Lane k packs slice k MSB-first into a 64-bit accumulator and
// emits finished 32-bit pieces towards lower addresses, which yields the bytes
// CodeWriter (codec/huffman.cpp:439-500) leaves behind: bit i of the stream is
// bit 7-(i&7) of byte end-1-(i>>3), and the 8 bytes at the start of the stream
// region are zero.
//
// Every lane reads its slice in aligned 32-byte blocks straight into registers
// (the next block is always in flight) and collects its output in a 128-byte
// LDS ring, from which whole aligned 32-byte blocks are stored.  The LDS table
// holds (code << 20) | len per symbol, so two codes are merged with 32-bit
// operations before one 64-bit shift places them in the accumulator.
constexpr uint32_t kEncBlk = 32;     // bytes per input block
constexpr uint32_t kEncRingDw = 32;  // output ring per lane: 128 bytes
constexpr uint32_t kEncGroup = 4;    // input blocks loaded at once per lane

// The output ring is stored transposed and backwards: the dword at output
// offset p (a multiple of 4) of lane l lives at ring row ~(p >> 2) & 31, LDS
// dword [row * 64 + l].  Lane l is always on bank l % 32 (no conflicts, no
// padding), the ring address of the dword being filled only ever grows (by one
// row per finished dword), and an aligned 32- or 64-byte output block is one
// aligned group of eight or sixteen rows in reverse order.
//
// kF = frames per wave.  With K <= 32 a frame would leave half the lanes idle,
// so lanes 0..31 take frame 2b and lanes 32..63 frame 2b+1 of workgroup b (each
// half with its own code table); everything past the prologue is per lane anyway.
//
// kEncOutBlk = bytes per stored output block: 64 on large batches, where fewer,
// larger pieces of a line reach memory (18 % faster at 1 GiB and beyond), 32
// when every SIMD has a single wave and the shorter wait for the ring reads
// matters more (9 % faster at 1024 frames).
//
// kParts: several threads per stream, for batches with too few frames to fill
// the chip one wave per frame (and for single frames).  A workgroup of up to
// 512 threads takes one frame; thread t encodes part t % P of stream t / P
// (P = 2^log2_parts parts per stream, cut like SliceSizes cuts the frame).  A
// part's codes have to start at the bit where the previous part's end: every
// thread first adds up the code lengths of its part, a prefix sum over the P
// threads of a stream (neighbouring lanes of one wave) gives each part its
// first bit B, and then the part is written like a stream of its own whose
// bytes run from end - (B >> 3) down to end - (B' >> 3), B' being the first
// bit of the next part.  A byte in which two parts meet belongs to the later
// one, which fetches the bits above its own from the symbols before its part
// (at most seven codes); the earlier part's bits of that byte, like the
// padding and the slop that every writer appends, fall outside its byte range
// and are clipped by the block stores.  The result is the same bytes as one
// lane per stream (codec/huffman.cpp:825-843).
template <uint32_t kF, uint32_t kEncOutBlk, bool kParts>
__global__ __launch_bounds__(kParts ? 512 : 64) void encode_kernel(
    uint32_t K, const uint8_t* __restrict__ raw, uint64_t total, uint64_t frame_bytes,
    uint64_t n_frames, const FrameCode* __restrict__ code, uint32_t code_step,
    const uint32_t* __restrict__ stream_end,
    const uint64_t* __restrict__ comp_off, const uint32_t* __restrict__ comp_size,
    uint8_t* __restrict__ comp, uint64_t cap, uint32_t log2_parts,
    const uint32_t* __restrict__ part_first) {
  static_assert(kF == 1 || kF == 2, "one or two frames per wave");
  static_assert(!kParts || kF == 1, "parts: one frame per workgroup");
  static_assert(kEncOutBlk == 32 || kEncOutBlk == 64, "the ring holds four or two blocks");
  __shared__ uint32_t enc_all[kF * 256];
  // one output ring per wave: with parts the launch sizes it (dynamic LDS), so
  // that small workgroups do not hold the LDS of the largest
  __shared__ uint32_t oring_one[kParts ? 1 : 64 * kEncRingDw];
  extern __shared__ uint32_t oring_dyn[];
  uint32_t* const oring = kParts ? oring_dyn : oring_one;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  const uint32_t half = kF == 1 ? 0 : lane >> 5;
  // stream index within the frame, part index within the stream
  const uint32_t k = kParts ? tid >> log2_parts : (kF == 1 ? lane : lane & 31);
  const uint32_t n_parts = kParts ? 1u << log2_parts : 1u;
  const uint32_t part = kParts ? tid & (n_parts - 1) : 0u;
  // threads that share the frame's table and header work, and this thread's index among them
  const uint32_t n_sub = kParts ? blockDim.x : 64 / kF;
  const uint32_t ti = kParts ? tid : k;
  const uint64_t f_real = (uint64_t)blockIdx.x * kF + half;
  const bool present = f_real < n_frames;
  const uint64_t f = present ? f_real : 0;
  uint32_t* const enc = enc_all + 256 * half;
  STAMP_DECL();
  const FrameCode& fc = code[f * code_step];
  for (uint32_t i = ti; i < 256; i += n_sub) {
    const uint32_t e = fc.enc[i];
    enc[i] = ((e >> 4) << 20) | (e & 15);
  }

  const uint64_t off = comp_off[f];
  const uint32_t size = comp_size[f];
  const bool ok = present && !(off > cap || size > cap - off);  // else reported by offsets_kernel
  uint8_t* out = comp + (ok ? off : 0);
  const uint32_t n = frame_len(total, frame_bytes, f);
  const uint32_t hdr_len = fc.hdr_len;
  const uint32_t hdr_total = 4 + hdr_len + 4 * (K - 1);

  // header: raw_size | len_mask | counts | sorted symbols | K-1 end offsets
  if (ok) {
    if (ti < 4) out[ti] = (uint8_t)(n >> (8 * ti));
    for (uint32_t i = ti; i < hdr_len; i += n_sub) out[4 + i] = fc.hdr[i];
    if (ti < K - 1) store_u32u(out + 4 + hdr_len + 4 * ti, stream_end[f * 64 + ti]);
  }
  if (kParts) __syncthreads();
  else wave_sync();  // enc[] complete
  STAMP(0);  // table + header
  if (!ok || k >= K) return;

  // input: offsets from ibase, the 32-byte block holding the frame's first raw
  // byte; output: offsets from obase likewise (pointer arithmetic keeps both in
  // the global address space)
  const uint8_t* const frame_in = raw + f * frame_bytes;
  const uint8_t* const ibase = frame_in - ((uintptr_t)frame_in & 31);
  uint8_t* const obase = out - ((uintptr_t)out & (kEncOutBlk - 1));
  const Slice s = slice_of(n, K, k);
  const uint32_t area = (uint32_t)((uintptr_t)out - (uintptr_t)obase) + hdr_total;
  uint32_t r_hi = area + stream_end[f * 64 + k];
  uint32_t r_lo = area + (k ? stream_end[f * 64 + k - 1] : 0);
  uint32_t in_lo = (uint32_t)((uintptr_t)frame_in - (uintptr_t)ibase) + s.begin;
  uint32_t in_hi = in_lo + s.size;
  uint32_t lead = 0, lead_n = 0;  // parts: the bits above this part's in its first byte
  if (kParts) {
    const Slice ps = slice_of(s.size, n_parts, part);
    const uint32_t slice_lo = in_lo;
    in_lo += ps.begin;
    in_hi = in_lo + ps.size;
    // code bits of the part: whole table entries are added up, sixteen at a
    // time (bits 4..19 of an entry are zero, so the low bits are the lengths'
    // sum).  The part is read in groups of four aligned 16-byte blocks with the
    // next group in flight; addresses past the part are clamped to its last block.
    uint32_t bits = 0;
    if (!part_first) {
      const uint32_t first16 = in_lo & ~15u;
      const uint32_t n16 = ps.size ? ((in_hi + 15) >> 4) - (in_lo >> 4) : 0;
      const uint32_t last16 = n16 ? first16 + 16 * (n16 - 1) : first16;
      const uint32_t n_groups = (n16 + 3) / 4;
      uint4 cur4[4], nxt4[4];
#define HFZ_LOAD4(dst, g)                                            \
  do {                                                               \
    _Pragma("unroll") for (uint32_t q_ = 0; q_ < 4; ++q_) {          \
      uint32_t a_ = first16 + 16 * (4 * (g) + q_);                   \
      a_ = a_ < last16 ? a_ : last16;                                \
      dst[q_] = *(const uint4*)(ibase + a_);                         \
    }                                                                \
  } while (0)
      if (n_groups) HFZ_LOAD4(nxt4, 0);
      for (uint32_t g = 0; g < n_groups; ++g) {
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) cur4[q] = nxt4[q];
        if (g + 1 < n_groups) HFZ_LOAD4(nxt4, g + 1);
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
          const uint32_t a = first16 + 16 * (4 * g + q);
          if (a > last16) break;
          const uint32_t w[4] = {cur4[q].x, cur4[q].y, cur4[q].z, cur4[q].w};
          uint32_t sum = 0;
          if (a >= in_lo && a + 16 <= in_hi) {
#pragma unroll
            for (uint32_t d = 0; d < 4; ++d)
              sum += enc[w[d] & 0xff] + enc[(w[d] >> 8) & 0xff] + enc[(w[d] >> 16) & 0xff] + enc[w[d] >> 24];
          } else {
#pragma unroll
            for (uint32_t d = 0; d < 4; ++d) {
#pragma unroll
              for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t pos = a + 4 * d + j;
                if (pos >= in_lo && pos < in_hi) sum += enc[(w[d] >> (8 * j)) & 0xff];
              }
            }
          }
          bits += sum & 0xfffffu;
        }
      }
#undef HFZ_LOAD4
    }
    // first bit of every part: prefix sum over the threads of the stream, or, for a
    // batch that was counted piece by piece (count_kernel with split_log2 =
    // log2_parts), read from what code_sizes_split_kernel left: no pass over the
    // input was needed then
    uint32_t incl = bits;
    uint32_t first_bit;
    if (part_first) {
      const uint32_t* pf = part_first + (f * 64 + k) * 8;
      first_bit = pf[part];
      incl = part + 1 < n_parts ? pf[part + 1] : first_bit;  // the last part's end is the stream's
    } else {
      for (uint32_t d = 1; d < n_parts; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, (int)n_parts);
        if (part >= d) incl += o;
      }
      first_bit = incl - bits;
    }
    const uint32_t stream_hi = r_hi;
    r_hi = stream_hi - (first_bit >> 3);
    if (part != n_parts - 1) r_lo = stream_hi - (incl >> 3);
    lead_n = first_bit & 7;
    // the lead_n bits before the part: the ends of the codes of the symbols before it
    uint32_t have = 0;
    for (uint32_t q = in_lo; have < lead_n && q > slice_lo;) {
      --q;
      const uint32_t e = enc[ibase[q]];
      const uint32_t len = e & 15;
      if (len) lead |= ((e >> 20) >> (12 - len)) << have;
      have += len;
    }
    lead &= (1u << lead_n) - 1;
    // a part whose bits all lie in bytes that later parts own has nothing to write
    if (r_lo >= r_hi) return;
  }

  uint8_t* const ocol = (uint8_t*)(oring + (kParts ? (tid >> 6) * 64 * kEncRingDw : 0u) + lane);  // row r of this lane: ocol + 256 * r
  // bit writer: wcur holds the pending bits of the dword being filled, MSB first.
  // S counts the stream's bits: its low five bits are the number of pending bits
  // and bits 5..9 the ring row of the dword being filled (one row per finished
  // dword), so (S << 3) & kRowMask is that row's byte offset.  Whole table
  // entries are added to S: bits 4..19 of an entry are zero, so the low twenty
  // bits of S stay exact while the code bits pile up above them; once per input
  // block S is cut back to those twenty bits and wp, the output offset just above
  // the dword being filled, is brought up to date.  If the stream does not end on
  // a dword boundary the bytes above its end count as already written.
  uint32_t wcur = 0;
  uint32_t nb0 = (r_hi & 3) ? 8 * (4 - (r_hi & 3)) : 0;
  if (kParts && lead_n) {
    wcur = (lead << (32 - lead_n)) >> nb0;  // nb0 + lead_n <= 31
    nb0 += lead_n;
  }
  const uint32_t wp0 = (r_hi + 3) & ~3u;
  constexpr uint32_t kRowMask = (kEncRingDw - 1) << 8;
  uint32_t S = (((~((wp0 - 4) >> 2)) & (kEncRingDw - 1)) << 5) + nb0;
  uint32_t s_seen = S;  // S when wp was last brought up to date
  uint32_t wp = wp0;
  uint32_t nf = (r_hi - 1) & ~(kEncOutBlk - 1);  // next block to store (descending)
  const uint32_t nf_last = r_lo & ~(kEncOutBlk - 1);
  bool nf_live = r_lo < r_hi;  // always true for a whole stream (it has its 8 bytes of slop)

  // Appends the left-aligned code bits v_ (at most 24) and moves on by the entries
  // sum_: the current dword is written every time (it is simply rewritten until it
  // is complete); when bit 5 of S flips the dword is complete and the bits that
  // did not fit become the next one.
#define HFZ_PUT(v_, sum_)                                                     \
  do {                                                                        \
    wcur |= (v_) >> (S & 31);                                                  \
    *(uint32_t*)(ocol + ((S << 3) & kRowMask)) = wcur;                         \
    const uint32_t sn_ = S + (sum_);                                          \
    const uint32_t full_ = (sn_ ^ S) & 32;                                    \
    wcur = (uint32_t)((((uint64_t)__builtin_amdgcn_alignbit((v_), 0u, S) << 32) | wcur) >> full_); \
    S = sn_;                                                                  \
  } while (0)
  // Two table entries merged: the second code moves down by the length of the
  // first; what the lengths leave in bits 0..3 is cut off.
#define HFZ_PAIR(b0, b1)                                                      \
  do {                                                                        \
    const uint32_t e0_ = enc[(b0)], e1_ = enc[(b1)];                          \
    const uint32_t pair_ = (e0_ | (e1_ >> (e0_ & 31))) & 0xffffff00u;         \
    HFZ_PUT(pair_, e0_ + e1_);                                                \
  } while (0)
#define HFZ_ONE(b0)                                                           \
  do {                                                                        \
    const uint32_t e0_ = enc[(b0)];                                           \
    const uint32_t one_ = e0_ & 0xfff00000u;                                  \
    HFZ_PUT(one_, e0_);                                                       \
  } while (0)
  // stores the ring block holding output offsets [blk, blk+kEncOutBlk); only bytes of
  // this stream's region [r_lo, r_hi) are written
#define HFZ_STORE_BLOCK(blk)                                                  \
  do {                                                                        \
    constexpr int kDw_ = kEncOutBlk / 4;                                      \
    const uint8_t* g_ = ocol + (((~((blk) >> 2)) & (kEncRingDw - kDw_)) << 8); \
    uint32_t sb_[kDw_];                                                       \
    _Pragma("unroll") for (int q_ = 0; q_ < kDw_; ++q_)                       \
      sb_[q_] = *(const uint32_t*)(g_ + 256 * (kDw_ - 1 - q_));               \
    if ((blk) >= r_lo && (blk) <= r_hi && r_hi - (blk) >= kEncOutBlk) {       \
      _Pragma("unroll") for (int q_ = 0; q_ < kDw_ / 4; ++q_)                 \
        *(uint4*)(obase + (blk) + 16 * q_) =                                  \
            make_uint4(sb_[4 * q_], sb_[4 * q_ + 1], sb_[4 * q_ + 2], sb_[4 * q_ + 3]); \
    } else {                                                                  \
      _Pragma("unroll") for (int q_ = 0; q_ < kDw_ / 8; ++q_)                 \
        store_block_clipped(obase, (blk) + 32 * q_, sb_ + 8 * q_, r_lo, r_hi); \
    }                                                                         \
  } while (0)

  // The slice is read in groups of kEncGroup aligned 32-byte blocks; the loads
  // of the next group are issued before the current one is encoded, so about
  // 2*kEncGroup*32 bytes per lane are in flight, which is what it takes to
  // cover HBM latency at this occupancy.  Addresses past the slice are clamped
  // to its last block (the data is ignored).
  const uint32_t nblk = s.size ? ((in_hi + kEncBlk - 1) / kEncBlk - in_lo / kEncBlk) : 0;
  const uint32_t first_blk = in_lo & ~(kEncBlk - 1);
  const uint32_t last_blk = nblk ? first_blk + kEncBlk * (nblk - 1) : first_blk;
  uint4 cur[2 * kEncGroup], nxt[2 * kEncGroup];
#define HFZ_LOAD_GROUP(dst, g)                                                \
  do {                                                                        \
    _Pragma("unroll") for (uint32_t q_ = 0; q_ < kEncGroup; ++q_) {           \
      uint32_t a_ = first_blk + kEncBlk * ((g) * kEncGroup + q_);             \
      a_ = a_ < last_blk ? a_ : last_blk;                                     \
      dst[2 * q_] = *(const uint4*)(ibase + a_);                              \
      dst[2 * q_ + 1] = *(const uint4*)(ibase + a_ + 16);                     \
    }                                                                         \
  } while (0)
  const uint32_t ngroups = (nblk + kEncGroup - 1) / kEncGroup;
  if (ngroups) HFZ_LOAD_GROUP(nxt, 0);
  STAMP(1);  // setup
  for (uint32_t g = 0; g < ngroups; ++g) {
#pragma unroll
    for (uint32_t q = 0; q < 2 * kEncGroup; ++q) cur[q] = nxt[q];
    if (g + 1 < ngroups) HFZ_LOAD_GROUP(nxt, g + 1);
#pragma unroll
    for (uint32_t q = 0; q < kEncGroup; ++q) {
      const uint32_t t = g * kEncGroup + q;
      if (t >= nblk) break;
      const uint32_t w[8] = {cur[2 * q].x,     cur[2 * q].y,     cur[2 * q].z,     cur[2 * q].w,
                             cur[2 * q + 1].x, cur[2 * q + 1].y, cur[2 * q + 1].z, cur[2 * q + 1].w};
      const uint32_t b = first_blk + kEncBlk * t;
      if (b >= in_lo && b + kEncBlk <= in_hi) {
#pragma unroll
        for (uint32_t d = 0; d < kEncBlk / 4; ++d) {
          HFZ_PAIR(w[d] & 0xff, (w[d] >> 8) & 0xff);
          HFZ_PAIR((w[d] >> 16) & 0xff, w[d] >> 24);
        }
      } else {
#pragma unroll
        for (uint32_t d = 0; d < kEncBlk / 4; ++d) {
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t p = b + 4 * d + j;
            if (p >= in_lo && p < in_hi) HFZ_ONE((w[d] >> (8 * j)) & 0xff);
          }
        }
      }
      // store the blocks that are complete (everything at or above wp is final)
      S &= 0xfffffu;
      wp -= (((S >> 5) - (s_seen >> 5)) & 0x7fffu) << 2;
      s_seen = S;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        if (nf_live && wp <= nf) {
          HFZ_STORE_BLOCK(nf);
          if (nf == nf_last) nf_live = false;
          nf -= kEncOutBlk;
        }
      }
    }
  }
#undef HFZ_LOAD_GROUP
  STAMP(3);  // main loop
  // end of stream: pad to a byte, then the 8 zero bytes of slop
  // (codec/huffman.cpp:770): the dword being filled and below it one zero dword,
  // or two if the padded dword is not empty
  *(uint32_t*)(ocol + ((S << 3) & kRowMask)) = wcur;
  *(uint32_t*)(ocol + (((S + 32) << 3) & kRowMask)) = 0;
  if (S & 31) *(uint32_t*)(ocol + (((S + 64) << 3) & kRowMask)) = 0;
  while (nf_live) {
    HFZ_STORE_BLOCK(nf);
    if (nf == nf_last) nf_live = false;
    nf -= kEncOutBlk;
  }
#undef HFZ_PUT
#undef HFZ_PAIR
#undef HFZ_ONE
#undef HFZ_STORE_BLOCK
  STAMP(4);  // tail
  STAMP_FLUSH(8);
}

// ---------------------------------------------------------------------------
// K4 + K5: decode.  The 4096-entry two-symbol table of Decoder2x
// (codec/huffman.cpp:644-681) is rebuilt in LDS from the frame header, then
// lane k decodes stream k (CodeReader + Decode, codec/huffman.cpp:502-586,
// 683-689, 931-953).  Entry layout (u32): byte 0 bits consumed, byte 1 eight
// times the symbol count, byte 2 first symbol, byte 3 second symbol, so that
// the bit position advances by adding the entry itself and appending its
// symbols is one shift by a running bit offset.
//
// Every lane moves its own stream between global memory and LDS in whole,
// aligned blocks: 64-byte input blocks (four 16-byte loads) into a two-block
// ring, 32-byte output blocks from a two-block ring.  Streams of one
// frame lie n/K bytes apart, so neighbouring lanes never share a cache line
// anyway; what matters is that each line is moved in few, aligned pieces.
//
// The wave works in super-rounds: every lane that has run out of staged input
// stages its next block (already loaded into registers during the previous
// super-round) and starts loading the one after it; then lanes run rounds of
// kDecRound lookups (at most 24 input bytes, at most 32 output bytes, one block
// store) until all of them have run out again.  All loads of a wave are thus
// issued together and have a whole super-round (~70 lookups) to arrive.
//
// The bit reader works on bit-reversed dwords.  A stream is read from its last
// byte downwards, most significant bit first; staging reverses the bits of every
// dword, so in the ring the stream is a sequence of dwords (descending
// addresses) that is consumed from bit 0 upwards.  The next code then sits at
// the BOTTOM of a 64-bit shift of two ring dwords, where one AND turns it into
// the byte offset of its table entry, and the table is indexed by the reversed
// code prefix.  The reader holds the two dwords under the bit position and the
// next one, fetched a pair of lookups ahead; after every pair it moves on by a
// whole dword or not at all (a pair consumes at most 24 bits), which is two
// bit-field selects and one aligned ring read.
//
// A lane does not stop exactly at the end of its slice: rounds always run all
// their lookups, and the up to 31 symbols decoded past the end (from whatever
// bits lie below the stream) stay in the output ring and are clipped away when
// the block is stored.  The decoded bytes of the slice are the same as with an
// exact stop (the last symbol may be the first half of a two-symbol entry,
// Decode1, codec/huffman.cpp:875-878: the second half is clipped).
constexpr uint32_t kDecInBlk = 64;   // bytes per staged input block
// Timing experiments only (-DHFZ_EMU_BITS=9|10, never in libhfz.so): the small-table
// decoder indexes with fewer bits and takes every longer code for one of that
// many bits, so the output is WRONG; what is measured is the speed a table of
// that size would allow.  -DHFZ_EMU_SEC adds the price of a second-level table
// read beside every lookup (a 16-bit read, a compare and a select).
#ifdef HFZ_EMU_BITS
constexpr uint32_t kSmallTabBits = HFZ_EMU_BITS;
#else
constexpr uint32_t kSmallTabBits = 11;
#endif
constexpr uint32_t kDecBlk = 32;     // bytes per stored output block
constexpr uint32_t kDecRound = 16;   // lookups per round (eight pairs)
constexpr uint32_t kDecInDw = 32;    // input ring per lane: 128 bytes
constexpr uint32_t kDecOutDw = 16;   // output ring per lane: 64 bytes
constexpr uint32_t kDecRoundDw = 6;  // a round consumes at most 16 * 12 bits = 6 dwords
// Both rings are stored transposed, dword d of lane l at [d * 64 + l]: whatever
// ring positions the lanes are at, an access puts lane l on bank l % 32, so the
// rings never cause bank conflicts and need no padding.  The input ring holds
// the dword at byte offset 4 * d of the stream in row 31 - (d & 31): rows grow
// as the reader walks down the stream.
//
// The decoder exists in two builds.  With kDecTabBits = 11 the lookup table is
// indexed by the next 11 bits, not 12: it is half the size
// (8 KiB), which is what bounds the number of resident waves per CU.  Codes of
// up to 11 bits, and pairs of codes of up to 11 bits together, resolve in one
// lookup.  12-bit codes come in sibling pairs that share an 11-bit prefix (the
// code is complete and they are the numerically last codes), so their entry
// holds both symbols (bytes 2 and 3) and counts one; the 12th bit picks the
// byte.  Plain one-symbol entries of this build carry their symbol in both
// bytes, so the same pick (one v_bfe_u32 whose offset is 16, or 24 if the
// entry counts one symbol and the 12th bit is set) serves every entry and
// nothing has to tell the sibling entries apart.

// Table-free build (kDecTabBits = 0): the reference's permute decoder
// (codec/huffman.cpp:1697-1756, 1892-1940) finds a code's length by comparing
// the window with the first code of every length and its symbol by indexing
// the canonical symbol list, and needs no lookup table.  The same here: 13
// words (first code and first symbol index per length) and the 256-byte symbol
// list in LDS instead of an 8 or 16 KiB table; one symbol per step, the length
// found by counting the lengths whose first code is not above the window's 12
// bits (the first codes ascend with the length).  It builds the same one-symbol
// entry a table lookup would return, so everything around it is shared.
__device__ __forceinline__ uint32_t table_free_entry(uint32_t c12, const uint32_t (&fend)[kMaxCodeLen],
                                                     const uint32_t* cls) {
  uint32_t L = 0;
#pragma unroll
  for (int l = 0; l < kMaxCodeLen; ++l) L += c12 >= fend[l] ? 1u : 0u;  // fend[l] = first code beyond length l
  const uint32_t cl = cls[L];                                           // first code | first symbol index << 16
  const uint32_t idx = (cl >> 16) + ((c12 - (cl & 0xffffu)) >> ((uint32_t)kMaxCodeLen - L));
  const uint32_t sym = ((const uint8_t*)(cls + 16))[idx];
  return L | (8u << 8) | (sym << 16);
}

// kWaves = 1: one frame per wave and workgroup, every frame builds its own table.
// kWaves > 1 (shared-code batches): kWaves frames per workgroup, one wave each;
// the table is built once, by all waves, from the header of the workgroup's
// first frame, and a frame whose code section differs from that one is refused
// (HFZ_E_CODE_MISMATCH) instead of being decoded with the wrong table.
// kF = 2 (K <= 32, kWaves = 1): lanes 0..31 decode frame 2b, lanes 32..63 frame
// 2b+1 of workgroup b.  The prologue (header, table) runs once per frame with
// all 64 lanes helping; the lanes of the frame's half keep its parameters.
// kStore: how finished 32-byte output blocks leave the ring.
//   0: the lane that decoded a block stores it (two 16-byte stores).
//   1: as 0, but a block waits in registers for its neighbour and the two leave
//      as one aligned 64-byte piece where they can (batches beyond the last-level
//      cache: fewer, larger pieces of a line reach memory; 7-8 % faster at 1 GiB
//      and beyond with one frame per wave, slower with two).
//   2: lanes 2i and 2i+1 store each other's blocks together, 16 bytes each side
//      by side, so that one store instruction writes 32 contiguous bytes of a
//      line instead of 16 (one frame per wave only).  A store instruction whose
//      64 lanes write 16 bytes of 64 different lines costs the memory pipeline
//      most; this halves the number of such pieces.  11-14 % faster with slices
//      of up to 2 KiB (128 KiB frames of 64 streams), 4-11 % slower with longer
//      ones, where a line reaches memory in four pieces that are rounds apart
//      instead of two and the lanes' lines crowd the same memory channels.
template <uint32_t kDecTabBits, uint32_t kWaves, uint32_t kF, uint32_t kStore, bool kParts, uint32_t kInBlk = kDecInBlk>
__global__ __launch_bounds__(kParts ? 512 : 64 * kWaves) void decode_kernel(
    uint32_t K, const uint8_t* __restrict__ comp, uint64_t comp_bytes,
    const uint64_t* __restrict__ comp_off, const uint32_t* __restrict__ comp_size,
    uint64_t n_frames, uint8_t* __restrict__ raw, uint64_t raw_cap, uint64_t frame_bytes,
    int32_t* __restrict__ status, uint32_t log2_parts) {
  static_assert(kF == 1 || (kF == 2 && kWaves == 1), "two frames per wave only without table sharing");
  static_assert(kStore <= 2 && (kStore != 2 || (kF == 1 && kWaves == 1)), "pair stores: one frame per wave or workgroup");
  constexpr bool kPair64 = kStore == 1;
  constexpr bool kTableFree = kDecTabBits == 0;
  // input ring per lane: two blocks of kInBlk bytes (64; 32 halves the ring at the
  // price of staging twice as often)
  static_assert(kInBlk == 32 || kInBlk == 64, "input blocks of 32 or 64 bytes");
  constexpr uint32_t kInDw = kInBlk / 2;
  constexpr uint32_t kBlkDw = kInBlk / 4;
  static_assert(!kTableFree || (kF == 1 && kWaves == 1), "table-free build: one frame per wave");
  static_assert(!kParts || (kF == 1 && kWaves == 1 && kStore != 1 && !kTableFree), "parts: one frame per workgroup");
  // table-free: words 0..12 first code | first symbol index << 16 per length, bytes 64.. the symbol list
  __shared__ uint32_t tab[kTableFree ? 16 + 64 : (kF << kDecTabBits)];
  uint32_t fend[kMaxCodeLen];  // table-free: first code beyond every length below 12 (wave-uniform)
#pragma unroll
  for (int l = 0; l < kMaxCodeLen; ++l) fend[l] = 0;
  // one pair of rings per wave; with parts the launch sizes them (dynamic LDS)
  __shared__ uint32_t ringmem_one[kParts ? 1 : kWaves * 64 * (kInDw + kDecOutDw)];
  extern __shared__ uint32_t ringmem_dyn[];
  uint32_t* const ringmem = kParts ? ringmem_dyn : ringmem_one;
  // parts: the length of the first code under every table index stays alive next to the table
  // (the three below only exist with parts: a one-wave workgroup's LDS is its table and rings)
  uint8_t* l1keep = nullptr;
  uint32_t* cls_keep = nullptr;
  int* wg_bad_p = nullptr;  // parts: some thread of the workgroup found its stream corrupt
  // parts: for every 12-bit (reversed) window the whole codes that lie in it: their bits in
  // byte 0, their number in byte 1.  Counting the symbols of a piece needs no symbols, so it
  // can take up to twelve codes per lookup where the decoding table gives two.
  uint16_t* cntab = nullptr;
  if constexpr (kParts) {
    __shared__ uint8_t l1keep_s[1u << kDecTabBits];
    __shared__ uint32_t cls_keep_s[16];
    __shared__ int wg_bad_s;
    __shared__ uint16_t cntab_s[1u << kMaxCodeLen];
    l1keep = l1keep_s;
    cls_keep = cls_keep_s;
    wg_bad_p = &wg_bad_s;
    cntab = cntab_s;
  }
#ifdef HFZ_EMU_SEC
  __shared__ uint16_t emu_sec[kF * (HFZ_EMU_BITS == 9 ? 1024 : 512)];
#endif
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = (kWaves == 1 && !kParts) ? 0 : tid >> 6;
  const uint32_t n_threads = kParts ? blockDim.x : 64 * kWaves;
  const uint32_t half = kF == 1 ? 0 : lane >> 5;
  // stream index within the frame (parts: thread t is part t % P of stream t / P)
  const uint32_t k = kParts ? tid >> log2_parts : (kF == 1 ? lane : lane & 31);
  uint32_t* const iring = ringmem + wave * 64 * (kInDw + kDecOutDw);
  uint32_t* const obuf_all = iring + 64 * kInDw;
  // scratch that is only live while a table is built (on the first wave's input
  // rings): the length of the first code under every (reversed) table index
  uint8_t* const l1tab = kParts ? l1keep : (uint8_t*)ringmem;
  static_assert(kParts || sizeof(uint32_t) * 64 * kInDw >= (1u << kDecTabBits), "l1tab fits in the input rings");
  static_assert(64 * kDecOutDw >= 192, "header staging fits in the output rows");
  STAMP_DECL();
#define HFZ_BARRIER()                           \
  do {                                          \
    if (kWaves == 1 && !kParts) wave_sync();    \
    else __syncthreads();                       \
  } while (0)

  // what a lane keeps of its frame
  uint64_t f = 0;
  bool present = false;
  int err = HFZ_OK;
  const uint8_t* in = comp;
  uint32_t raw_size = 0, hdr_total = 0, end = kSlop;
  uint32_t begin_k = 0;      // parts: where this thread's stream begins in the stream area
  bool multi_symbol = false;  // parts: the code has at least two symbols (every code has a bit)
  if (kParts && tid == 0) *wg_bad_p = 0;

#pragma unroll
  for (uint32_t pass = 0; pass < kF; ++pass) {
    const bool mine = kF == 1 || half == pass;
    const uint64_t pf = kParts ? (uint64_t)blockIdx.x : ((uint64_t)blockIdx.x * kWaves + wave) * kF + pass;
    // The frame's header: one round trip brings all of it (at most 8 + 13 + 256
    // + 4*63 bytes) into LDS; everything below parses it there.
    const bool p_present = pf < n_frames;
    uint64_t off = 0;
    uint32_t size = 0;
    if (p_present) {
      off = comp_off[pf];
      size = comp_size[pf];
    }
    int p_err = HFZ_OK;
    if (off > comp_bytes || size > comp_bytes - off || size < 8) {
      p_err = HFZ_E_CORRUPT;
      size = 0;
    }
    const uint8_t* p_in = comp + (p_err == HFZ_OK ? off : 0);
    uint32_t* const hdrw = obuf_all;  // 192 dwords, dead once the streams start
#pragma unroll
    for (uint32_t j = 0; j < 3; ++j) {
      const uint32_t idx = lane + 64 * j;
      uint32_t v = 0;
      if (4 * idx + 4 <= size) {
        v = load_u32u(p_in + 4 * idx);
      } else {
        for (uint32_t b = 0; b < 4; ++b)
          if (4 * idx + b < size) v |= (uint32_t)p_in[4 * idx + b] << (8 * b);
      }
      hdrw[idx] = v;
    }
    HFZ_BARRIER();

    // The code section (len_mask, per-length counts, canonical symbol order) is
    // read from the workgroup's first frame, which is this frame when kWaves = 1.
    // ParseCompressedHeader (codec/huffman.cpp:714-736) + canonical code starts
    // (ForallCodes, codec/huffman.cpp:260-284).
    const uint32_t* const lead = ringmem + 64 * kInDw;
    const uint8_t* const hb = (const uint8_t*)lead;
    const uint32_t len_mask = lead[1];
    const uint32_t nlen = __popc(len_mask);
    int lead_err = HFZ_OK;
    if ((len_mask >> (kMaxCodeLen + 1)) != 0) lead_err = HFZ_E_CORRUPT;
    uint32_t cnt[kMaxCodeLen + 1], fcode[kMaxCodeLen + 2], soff[kMaxCodeLen + 1];
    uint32_t cur = 0, nsym = 0, hp = 8;
#pragma unroll
    for (int L = 0; L <= kMaxCodeLen; ++L) {
      uint32_t c = 0;
      if ((len_mask >> L) & 1) {
        c = hb[hp++];
        if (c == 0 && nlen == 1) c = 256;  // 256 is stored as 0 (:724-728)
      }
      cnt[L] = c;
      fcode[L] = cur;
      soff[L] = nsym;
      cur += c << (kMaxCodeLen - L);
      nsym += c;
    }
    fcode[kMaxCodeLen + 1] = cur;
    const uint32_t lead_raw_size = lead[0];
    if (nsym > 256) lead_err = HFZ_E_CORRUPT;
    if (nsym == 0 ? lead_raw_size != 0 : cur != (1u << kMaxCodeLen)) lead_err = HFZ_E_CORRUPT;
    if (kF == 1 && lead_err != HFZ_OK) {  // uniform over the workgroup: nobody waits at a barrier below
      if ((kParts ? tid : lane) == 0 && status && p_present) status[pf] = p_err != HFZ_OK ? p_err : lead_err;
      return;
    }
    if (p_err == HFZ_OK) p_err = lead_err;
    const uint8_t* const syms = hb + 8 + nlen;  // canonical symbol order

    // this frame's own numbers
    const uint32_t p_raw_size = hdrw[0];
    const uint32_t p_hdr_total = 8 + nlen + nsym + 4 * (K - 1);
    if (p_err == HFZ_OK) {
      const uint64_t p_out = pf * frame_bytes;
      uint64_t room = p_out <= raw_cap ? raw_cap - p_out : 0;
      if (room > frame_bytes) room = frame_bytes;
      if (p_raw_size > room) p_err = HFZ_E_CAPACITY;
      if ((uint64_t)p_hdr_total + (uint64_t)kSlop * K > size) p_err = HFZ_E_CORRUPT;
      if (nsym == 0 && p_raw_size != 0) p_err = HFZ_E_CORRUPT;
      if (kWaves > 1) {
        // same code section as the first frame, byte for byte: dwords 1.. up to 8 + nlen + nsym
        const uint32_t code_end = 8 + nlen + nsym;
        bool differs = false;
        for (uint32_t i = 1 + lane; 4 * i < code_end; i += 64) {
          uint32_t m = 0xffffffffu;
          if (4 * i + 4 > code_end) m >>= 8 * (4 * i + 4 - code_end);
          differs |= ((hdrw[i] ^ lead[i]) & m) != 0;
        }
        if (__any(differs) && p_err == HFZ_OK) p_err = HFZ_E_CODE_MISMATCH;
      }
    }

    // stream boundaries (codec/huffman.cpp:897-901); lanes without a stream get an empty one
    const uint32_t area_size = p_err == HFZ_OK ? size - p_hdr_total : 0;
    uint32_t p_end = kSlop;
    uint32_t p_begin = 0;
    if (p_err == HFZ_OK) {
      const uint32_t eo = 8 + nlen + nsym;  // byte offset of the end-offset array
      // the stream between end offsets kk - 1 and kk; *bad_: it does not fit the area or its slop
#define HFZ_STREAM_BOUNDS(kk, begin_, end_, bad_)                                             \
  do {                                                                                        \
    if ((kk) < K - 1) {                                                                       \
      const uint32_t q_ = eo + 4 * (kk);                                                      \
      end_ = __builtin_amdgcn_alignbyte(hdrw[(q_ >> 2) + 1], hdrw[q_ >> 2], q_ & 3);          \
    } else {                                                                                  \
      end_ = area_size;                                                                       \
    }                                                                                         \
    if (kk) {                                                                                 \
      const uint32_t q_ = eo + 4 * ((kk) - 1);                                                \
      begin_ = __builtin_amdgcn_alignbyte(hdrw[(q_ >> 2) + 1], hdrw[q_ >> 2], q_ & 3);        \
    }                                                                                         \
    bad_ = end_ > area_size || begin_ > end_ || end_ - begin_ < (uint32_t)kSlop;              \
  } while (0)
      bool bad = false;
      if (kParts) {
        // every wave checks all streams (lane l: stream l), so that the verdict is the
        // same in every wave and the workgroup leaves, or stays, together
        if (lane < K) {
          uint32_t b0 = 0, e0 = kSlop;
          HFZ_STREAM_BOUNDS(lane, b0, e0, bad);
        }
        bad = __any(bad);
        if (!bad && k < K) {
          bool mine_bad;
          HFZ_STREAM_BOUNDS(k, p_begin, p_end, mine_bad);
          (void)mine_bad;
        }
      } else if (mine && k < K) {
        HFZ_STREAM_BOUNDS(k, p_begin, p_end, bad);
      }
#undef HFZ_STREAM_BOUNDS
      if (__any(bad)) p_err = HFZ_E_CORRUPT;
    }
    if (kWaves == 1 && kF == 1 && p_err != HFZ_OK) {
      if ((kParts ? tid : lane) == 0 && status && p_present) status[pf] = p_err;
      return;
    }
    if (mine) {
      f = pf;
      present = p_present;
      err = p_err;
      in = p_in;
      raw_size = p_raw_size;
      hdr_total = p_hdr_total;
      end = p_end;
      begin_k = p_begin;
      multi_symbol = nsym >= 2;
    }
    STAMP(0);  // header

    // Single-symbol entries.  The reader looks codes up bit-reversed, so entry r
    // of the table belongs to the code prefix whose reversed bits are r.  Every
    // thread fills the entries r = tid, tid + threads, ...: neighbouring lanes
    // write neighbouring words (filling class by class in code order instead puts
    // a wave's 64 stores on one or two banks, because codes that are neighbours in
    // code order differ in their LAST bits, the top bits of r).  The length of the
    // code under a prefix is the number of lengths whose codes end at or below it
    // (the first codes ascend with the length); first code and first symbol index
    // of every length sit in a 13-word list next to the scratch.
    uint32_t* const ptab = tab + (pass << kDecTabBits);
    constexpr uint32_t kTabShift = kTableFree ? 0 : (uint32_t)kMaxCodeLen - kDecTabBits;
#ifndef HFZ_EMU_BITS
    static_assert(kTabShift <= 1, "sibling entries cover one bit");
#endif
    // right behind l1tab, still inside the first wave's input ring (the header sits in its output ring)
    uint32_t* const cls = kParts ? cls_keep : ringmem + (kTableFree ? 0 : (1u << kDecTabBits) / 4);
    static_assert(kParts || kTableFree || sizeof(uint32_t) * 64 * kInDw >= (1u << kDecTabBits) + 64,
                  "l1tab and the class list fit in the input ring");
    if (kTableFree) {
#pragma unroll
      for (int l = 0; l < kMaxCodeLen; ++l) fend[l] = __builtin_amdgcn_readfirstlane(fcode[l + 1]);
#pragma unroll
      for (int L = 0; L <= kMaxCodeLen; ++L)
        if (lane == (uint32_t)L) tab[L] = fcode[L] | (soff[L] << 16);
      uint8_t* const list = (uint8_t*)(tab + 16);
      for (uint32_t i = lane; i < nsym; i += 64) list[i] = syms[i];
    } else {
#pragma unroll
      for (int L = 0; L <= kMaxCodeLen; ++L)
        if (tid == (uint32_t)L) cls[L] = fcode[L] | (soff[L] << 16);
    }
    HFZ_BARRIER();
    if (!kTableFree && (kF == 1 || p_err == HFZ_OK)) {
#pragma unroll 4
      for (uint32_t r = tid; r < (1u << kDecTabBits); r += n_threads) {
        const uint32_t c12 = (__builtin_bitreverse32(r) >> (32 - (kTableFree ? 1 : kDecTabBits))) << kTabShift;
        uint32_t L = 0;
#pragma unroll
        for (int l = 0; l < kMaxCodeLen; ++l) L += c12 >= fcode[l + 1] ? 1u : 0u;
#ifdef HFZ_EMU_BITS
        if (L > kDecTabBits) L = kDecTabBits;
#endif
        const uint32_t cl = cls[L];
        const uint32_t d = c12 - (cl & 0xffffu);
        // 11-bit table: a 12-bit code shares its entry with its sibling (their
        // number is even: the code is complete), both symbols in the entry; a
        // lone symbol sits in bytes 2 and 3 alike
        const bool sibling = kTabShift && L == (uint32_t)kMaxCodeLen;
        uint32_t j = (cl >> 16) + (sibling ? d : d >> ((uint32_t)kMaxCodeLen - L));
#ifdef HFZ_EMU_BITS
        j &= 255;
#endif
        const uint32_t s0 = syms[j];
        const uint32_t s1 = sibling ? (uint32_t)syms[j + 1] : (kTabShift ? s0 : 0u);
        ptab[r] = L | (8u << 8) | (s0 << 16) | (s1 << 24);
        l1tab[r] = (uint8_t)L;
      }
    }
    HFZ_BARRIER();
    // second symbol where both codes fit in the index bits: dropping the first
    // code from a prefix is a right shift of its reversed index.  Byte 2 of an
    // entry and l1tab never change in this pass, so reading another lane's entry
    // while it is being rewritten is safe.
    if (!kTableFree && (kF == 1 || p_err == HFZ_OK)) {
#pragma unroll 8
      for (uint32_t r = tid; r < (1u << kDecTabBits); r += n_threads) {
        const uint32_t L1 = l1tab[r];
        const uint32_t rr = r >> L1;
        const uint32_t L2 = l1tab[rr];
        if (L1 + L2 <= kDecTabBits)
          ptab[r] = (L1 + L2) | (16u << 8) | (ptab[r] & 0xff0000) | ((ptab[rr] & 0xff0000) << 8);
      }
    }
#ifdef HFZ_EMU_SEC
    for (uint32_t i = tid; i < (HFZ_EMU_BITS == 9 ? 1024u : 512u); i += n_threads)
      emu_sec[pass * (HFZ_EMU_BITS == 9 ? 1024u : 512u) + i] = (uint16_t)(syms[i & 255] | (i << 8));
#endif
    if (kParts) {
      // the first code's length is known for every prefix (l1tab): walk the window code by code
      for (uint32_t r = tid; r < (1u << kMaxCodeLen); r += n_threads) {
        uint32_t pos = 0, c = 0;
        for (int it = 0; it < kMaxCodeLen; ++it) {
          const uint32_t L = l1tab[(r >> pos) & ((1u << kDecTabBits) - 1)];
          if (L == 0 || pos + L > (uint32_t)kMaxCodeLen) break;
          pos += L;
          ++c;
        }
        cntab[r] = (uint16_t)(pos | (c << 8));
      }
    }
    HFZ_BARRIER();  // table complete; the scratch aliased on the rings is free again
    STAMP(1);  // table
  }
#undef HFZ_BARRIER
  if ((kWaves > 1 || kF > 1) && (err != HFZ_OK || !present)) {  // past the last barrier: lanes part ways
    if (k == 0 && status && present) status[f] = err;
    return;
  }
  const uint8_t* const tabb = (const uint8_t*)(tab + (half << kDecTabBits));
  // the lookup mask lives in a scalar register (K < 2^30: the OR adds nothing, but the
  // compiler cannot fold it): an AND with a 32-bit literal is an 8-byte instruction
  const uint32_t kTabMask = (kTableFree ? 0 : ((1u << kDecTabBits) - 1) << 2) | (K >> 30);

  // Input side: offsets are relative to cbase, a 64-byte aligned address 256
  // bytes below the frame, so that the block arithmetic below (up to three
  // blocks of look-ahead) never goes negative.  Blocks are 64-byte aligned in
  // absolute terms, so none straddles a page; blocks that start below the
  // 64-byte block holding the first byte of `comp` are never loaded (a valid
  // stream never needs them).
  const uint8_t* cbase = in - ((uintptr_t)in & 63) - 256;  // pointer arithmetic: stays global
  const int64_t min_off = (int64_t)(((uintptr_t)comp & ~uintptr_t(63)) - (uintptr_t)cbase);
  const uint32_t area_off = (uint32_t)((uintptr_t)in - (uintptr_t)cbase) + hdr_total;
  const uint32_t g_end = area_off + end;  // one past the stream's last byte
  // Output side: offsets relative to obase, the 32-byte block holding the
  // frame's first output byte.
  uint8_t* const frame_out = raw + f * frame_bytes;
  uint8_t* const obase = frame_out - ((uintptr_t)frame_out & 31);
  Slice s;
  s.begin = 0;
  s.size = 0;
  if (k < K) s = slice_of(raw_size, K, k);
  // the bytes this thread produces: its stream's slice, or with parts a piece of it (set below)
  uint32_t out_lo = (uint32_t)((uintptr_t)frame_out - (uintptr_t)obase) + s.begin;
  uint32_t out_hi = out_lo + s.size;
  int32_t left = (int32_t)s.size;  // symbols still to produce; a round may overshoot below zero
  bool stream_bad = false;

  uint32_t* const irow = iring + lane;      // row r of this lane's input ring: irow[64 * r]
  uint32_t* const orow32 = obuf_all + lane;  // dword d of its output ring: orow32[64 * d]

  // loads block `blk` (offset from cbase) of this lane's stream into registers
  uint4 pre[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) pre[q] = make_uint4(0, 0, 0, 0);
#define HFZ_LOAD_BLOCK(blk)                                       \
  do {                                                            \
    if ((int64_t)(int32_t)(blk) >= min_off) {                     \
      _Pragma("unroll") for (int q_ = 0; q_ < (int)(kInBlk / 16); ++q_) \
        pre[q_] = *(const uint4*)(cbase + (blk) + 16 * q_);       \
    }                                                             \
  } while (0)
  // writes the registers, every dword bit-reversed, into the ring rows of block
  // `blk`: dword j of a block of n dwords goes to row (blk & 4n ? 0 : n) + n - 1 - j
#define HFZ_STAGE_BLOCK(blk)                                      \
  do {                                                            \
    uint32_t* d_ = irow + 64 * (((blk) & kInBlk) ? 0 : kBlkDw);   \
    _Pragma("unroll") for (int q_ = 0; q_ < (int)(kInBlk / 16); ++q_) { \
      d_[64 * ((int)kBlkDw - 1 - 4 * q_)] = __builtin_bitreverse32(pre[q_].x); \
      d_[64 * ((int)kBlkDw - 2 - 4 * q_)] = __builtin_bitreverse32(pre[q_].y); \
      d_[64 * ((int)kBlkDw - 3 - 4 * q_)] = __builtin_bitreverse32(pre[q_].z); \
      d_[64 * ((int)kBlkDw - 4 - 4 * q_)] = __builtin_bitreverse32(pre[q_].w); \
    }                                                             \
  } while (0)

  // Bit reader (CodeReader, codec/huffman.cpp:502-586).  wlo and whi are the
  // (reversed) dwords under the bit position, wlo the one being consumed; wnx
  // is the dword below them, read from the ring a pair of lookups before it is
  // needed, so no ring read sits on the lookup chain.  sacc = (bits of wlo
  // consumed) - 2, kept in 0..31 between pairs: the 64-bit shift of whi:wlo by
  // sacc puts the next code at bits 2.., i.e. times four, which is the byte
  // offset of its table entry.  Within a pair sacc has whole entries added to
  // it: the shifts look at its low six bits only, and bit 5 after the pair
  // says that wlo is used up (a pair consumes at most 24 bits).
  // ra = 256 * (ring row of wnx), running on without wrapping; ra_low the same
  // for the lowest staged dword, so ra_low - ra = 256 * (staged dwords below
  // wnx); ra_area for the dword holding the first byte of the stream area.
  uint32_t low_blk = 0;  // lowest block staged so far
  uint32_t ra = 0, ra_low = 0, wlo = 0, whi = 0, wnx = 0, sacc = 0;
  const uint32_t ra_area = (31u - (area_off >> 2)) << 8;
  const uint8_t* const icol = (const uint8_t*)irow;
  constexpr uint32_t kInRowMask = (kInDw - 1) << 8;
  // acc_ += x_ << 3 as ONE instruction (left to itself the compiler keeps a second
  // running sum and rebuilds acc_ from it: two instructions per step)
#define HFZ_LSHL3_ADD(acc_, x_) asm("v_lshl_add_u32 %0, %1, 3, %0" : "+v"(acc_) : "v"(x_))
  // after a pair: wlo used up -> whi:wlo <- wnx:whi, and the next wnx is read
#define HFZ_REFILL()                                                         \
  do {                                                                       \
    const uint32_t z_ = sacc & 32;                                           \
    wlo = z_ ? whi : wlo;                                                    \
    whi = z_ ? wnx : whi;                                                    \
    HFZ_LSHL3_ADD(ra, z_);                                                   \
    sacc &= 31;                                                              \
    wnx = *(const uint32_t*)(icol + (ra & kInRowMask));                      \
  } while (0)
  // Puts the reader at bit bp_ of the stream (0 = its first bit, the top bit of
  // its last byte): primes the ring with the two blocks below that byte and starts
  // loading the third; then one dword up: the bytes of the dword above the
  // position, the whole dword above that one and the bp_ & 7 bits of the byte
  // itself count as consumed, and the first refill moves down.
#define HFZ_READER_AT(bp_)                                                   \
  do {                                                                       \
    const uint32_t ge_ = g_end - ((bp_) >> 3); /* one past the byte holding the bit */ \
    low_blk = (ge_ - 1) & ~(kInBlk - 1);                                     \
    HFZ_LOAD_BLOCK(low_blk);                                                 \
    HFZ_STAGE_BLOCK(low_blk);                                                \
    low_blk -= kInBlk;                                                       \
    HFZ_LOAD_BLOCK(low_blk);                                                 \
    HFZ_STAGE_BLOCK(low_blk);                                                \
    HFZ_LOAD_BLOCK(low_blk - kInBlk);                                        \
    const uint32_t d_last_ = (ge_ - 1) >> 2; /* dword holding that byte */   \
    ra = (31u - (d_last_ - 1)) << 8;                                         \
    ra_low = (31u - (low_blk >> 2)) << 8;                                    \
    wlo = *(const uint32_t*)(icol + ((ra - 512) & kInRowMask));              \
    whi = *(const uint32_t*)(icol + ((ra - 256) & kInRowMask));              \
    wnx = *(const uint32_t*)(icol + (ra & kInRowMask));                      \
    sacc = 32 + 8 * (3 - ((ge_ - 1) & 3)) - 2 + ((bp_) & 7);                 \
    HFZ_REFILL();                                                            \
  } while (0)
  // the entry for the code at the bottom of window t_ (bits 2..)
#ifdef HFZ_EMU_SEC
  const uint32_t emu_n1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((raw_size & 1) + 3));
  const uint8_t* const emu_secb = (const uint8_t*)(emu_sec + half * (HFZ_EMU_BITS == 9 ? 1024 : 512));
  constexpr uint32_t kEmuSecMask = ((HFZ_EMU_BITS == 9 ? 1024u : 512u) - 1) << 1;
#define HFZ_ENTRY(t_)                                                                           \
  ([&]() {                                                                                      \
    const uint32_t e1_ = *(const uint32_t*)(tabb + ((t_) & kTabMask));                          \
    const uint32_t e2_ = *(const uint16_t*)(emu_secb + (((t_) >> emu_n1) & kEmuSecMask));       \
    return (int32_t)e1_ < 0 ? ((e2_ << 16) | (0x800u + kDecTabBits)) : e1_;                     \
  }())
#else
#define HFZ_ENTRY(t_)                                                                           \
  (kTableFree ? table_free_entry(__builtin_bitreverse32((t_) >> 2) >> 20, fend, tab)            \
              : *(const uint32_t*)(tabb + ((t_) & kTabMask)))
#endif
  // A round moves the reader down by at most kDecRoundDw dwords and its last
  // refill reads the dword below: a lane can run a round when that many dwords
  // below wnx are staged.
  constexpr int32_t kReach = (int32_t)(kDecRoundDw << 8);

  uint32_t start_bit = 0;  // where this thread starts reading its stream
  if (kParts) {
    // Several threads per stream.  A stream is one chain of codes with no marks
    // in it, but a decoder that starts at a wrong bit falls into step with the
    // true code boundaries after a few codes (a Huffman code synchronises
    // itself), so: the stream's bits are cut into P pieces of equal length;
    // thread p starts kGuard bits before its piece, takes the first code boundary
    // it reaches at or beyond the piece's start as its start s_p, and decodes on
    // (counting symbols, writing nothing) to the first boundary e_p at or beyond
    // the next piece's start.  If e_(p-1) = s_p for every p, all starts are true
    // boundaries (thread 0's is, and each thread hands a boundary of its own chain
    // to the next); a thread whose start is not what its neighbour reached starts
    // again from there, until nothing changes (at most P times, once in practice).
    // A prefix sum of the counts places every piece in the output, and the
    // pieces are decoded like streams of their own.  Same bytes as one lane per
    // stream (codec/huffman.cpp:931-953).
    const uint32_t P = 1u << log2_parts;
    const uint32_t part = tid & (P - 1);
    // far enough for text-like codes to fall into step all but once in a few thousand
    // starts (skewed codes do so within 64 bits)
    constexpr uint32_t kGuard = 256;
    // bits of the stream, rounded up to whole bytes; a code without bits (a frame
    // of one symbol) gives nothing to cut: everything goes to the last part
    const uint32_t total_bits = (multi_symbol && k < K && s.size) ? 8 * (end - begin_k - (uint32_t)kSlop) : 0;
    const uint32_t piece = (total_bits + P - 1) >> log2_parts;
    const uint32_t b_lo = part * piece < total_bits ? part * piece : total_bits;
    const uint32_t b_hi = (part + 1) * piece < total_bits ? (part + 1) * piece : total_bits;
    const bool last_part = part == P - 1;
    uint32_t s_bit = 0, e_bit = 0, n_sym = 0;
    for (uint32_t iter = 0; iter <= P; ++iter) {
      // this pass: start at bit x; while in_pre run to tgt = b_lo without counting
      // (that boundary becomes s_bit), then count up to b_hi
      uint32_t x, tgt;
      bool in_pre = false, run;
      if (iter == 0) {
        run = true;
        in_pre = part != 0;
        x = in_pre ? (b_lo > kGuard ? b_lo - kGuard : 0) : 0;
        tgt = in_pre ? b_lo : b_hi;
      } else {
        const uint32_t prev_e = (uint32_t)__shfl_up((int)e_bit, 1, (int)P);
        run = part != 0 && prev_e != s_bit;
        if (!__any(run)) break;
        if (run) s_bit = prev_e;
        x = s_bit;
        tgt = last_part ? x : b_hi;  // the last part only needs its start
      }
      if (in_pre && x >= tgt) {  // nothing to skip: the piece starts the stream or lies beyond its end
        s_bit = x;
        in_pre = false;
        tgt = last_part ? x : b_hi;
      }
      uint32_t n = 0;
      bool active = run && x < tgt;
      if (__any(active)) {
        if (active) HFZ_READER_AT(x);
        for (;;) {
          if (!__any(active)) break;
          if (active && (int32_t)(ra_low - ra) < kReach) {  // out of staged input: next block
            if ((int32_t)(ra_low - ra_area) >= kReach) {    // see the decoding loop below
              stream_bad = true;
              active = false;
            } else {
              low_blk -= kInBlk;
              ra_low += kBlkDw << 8;
              HFZ_STAGE_BLOCK(low_blk);
              HFZ_LOAD_BLOCK(low_blk - kInBlk);
            }
          }
          for (;;) {
            const bool can = active && (int32_t)(ra_low - ra) >= kReach;
            if (!__any(can)) break;
            if (can) {
              if (tgt - x > kDecRound * (uint32_t)kMaxCodeLen) {
                // far from the target: a round of lookups as in the decoding loop, but in the
                // counting table (all whole codes of a 12-bit window at once) and without output
                const uint32_t ra_b = ra, sacc_b = sacc;
                uint32_t cnt = 0;
                const uint8_t* const cnb = (const uint8_t*)cntab;
                constexpr uint32_t kCntMask = ((1u << kMaxCodeLen) - 1) << 1;
#pragma unroll
                for (uint32_t j = 0; j < kDecRound / 2; ++j) {
                  const uint64_t win = ((uint64_t)whi << 32) | wlo;
                  const uint32_t t1 = (uint32_t)(win >> (sacc & 63));
                  const uint32_t e1 = *(const uint16_t*)(cnb + ((t1 >> 1) & kCntMask));
                  sacc += e1;  // bits of the codes; what byte 1 adds lies above the bits looked at
                  const uint32_t t2 = (uint32_t)(win >> (sacc & 63));
                  const uint32_t e2 = *(const uint16_t*)(cnb + ((t2 >> 1) & kCntMask));
                  sacc += e2;
                  cnt += e1 + e2;  // byte 0 stays below 256 over a round (16 * 12 bits), so byte 1.. is the count
                  HFZ_REFILL();
                }
                x += ((ra - ra_b) >> 3) + sacc - sacc_b;  // 256 per dword = 8 per bit
                n += cnt >> 8;
              } else {
                // near the target: an entry is taken whole while even its last bit lies
                // before the target (then so does the boundary between its two codes);
                // else one code at a time
#pragma unroll 4
                for (uint32_t j = 0; j < kDecRound; ++j) {
                  const uint64_t win = ((uint64_t)whi << 32) | wlo;
                  const uint32_t t = (uint32_t)(win >> (sacc & 63));
                  const uint32_t e = HFZ_ENTRY(t);
                  uint32_t bits = e & 0xffu, syms = (e >> 11) & 3u;
                  const bool go = x < tgt;
                  if (__any(go && x + bits > tgt)) {
                    const uint32_t l1 = l1tab[(t & kTabMask) >> 2];
                    if (x + bits > tgt) {
                      bits = l1;
                      syms = 1;
                    }
                  }
                  bits = go ? bits : 0;
                  syms = go ? syms : 0;
                  sacc += bits;
                  x += bits;
                  n += syms;
                  HFZ_REFILL();
                }
              }
              if (in_pre && x >= tgt) {  // reached the piece: this boundary is the start
                s_bit = x;
                n = 0;
                in_pre = false;
                tgt = last_part ? x : b_hi;
              }
              active = x < tgt;
            }
          }
        }
      }
      if (run) {
        e_bit = x;
        n_sym = n;
      }
    }
    // place the pieces: symbols before this one, clamped to the slice (a piece that
    // reads past the end of a stream counts padding bits as codes)
    uint32_t incl = last_part ? 0u : n_sym;
    for (uint32_t d = 1; d < P; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, (int)d, (int)P);
      if (part >= d) incl += o;
    }
    uint32_t before = incl - (last_part ? 0u : n_sym);
    before = before < s.size ? before : s.size;
    uint32_t mine_n = last_part ? s.size - before : n_sym;
    if (mine_n > s.size - before) mine_n = s.size - before;
    out_lo += before;
    out_hi = out_lo + mine_n;
    left = (int32_t)mine_n;
    start_bit = s_bit;
  }
  if (!kParts || left > 0) HFZ_READER_AT(start_bit);
  STAMP(2);  // priming

  // output: decoded bytes collect in ocur (sh = 8 * their number, < 32 between
  // pairs); wp is the output offset of the dword being filled.  That
  // dword is written to the ring after every pair and simply rewritten until it
  // is complete.  nfb = next 32-byte block to leave the ring.  A full block is
  // read into sb[] at the end of the round that completed it and stored to
  // memory at the end of the next one, when the LDS reads have long returned.
  uint32_t ocur = 0;
  uint32_t sh = 8 * (out_lo & 3);  // bytes below the slice start in the first dword
  uint32_t wp = out_lo & ~3u;
  // wa = 256 * (ring row of the dword at wp), running on without wrapping: one
  // row per finished dword.  wp itself is brought up to date once per round.
  uint8_t* const ocol = (uint8_t*)orow32;
  constexpr uint32_t kOutRowMask = (kDecOutDw - 1) << 8;
  uint32_t wa = ((wp >> 2) & (kDecOutDw - 1)) << 8;
  uint32_t nfb = out_lo & ~(kDecBlk - 1);
  uint32_t sb[8];   // the block at sb_at; with kPair64 the even block of a pair
  uint32_t sb_at = 0;
  bool sb_full = false;
  uint32_t sc[8];  // kPair64: the odd block of a pair, at sc_at
  uint32_t sc_at = 0;
  bool sc_full = false;

  // the symbols of an entry, w_ = 8 * their number (in bits 0..4; v_bfe_u32 looks no further).
  // 11-bit table: offset 24 instead of 16 iff the entry counts one symbol (bit 3
  // of w_) and the 12th bit of the code (bit 13 of t_) is set.
#ifdef HFZ_EMU_BITS
#define HFZ_EMU_PLAIN (kDecTabBits == kSmallTabBits)
#else
#define HFZ_EMU_PLAIN false
#endif
#define HFZ_SYMS(e_, t_, w_)                                                                    \
  ((!kTableFree && !HFZ_EMU_PLAIN && kDecTabBits < (uint32_t)kMaxCodeLen)                       \
       ? __builtin_amdgcn_ubfe((e_), ((((t_) >> 10) & (w_)) | 16), (w_))                        \
       : (e_) >> 16)
  // two lookups, their symbols appended to ocur, the finished dword written out,
  // the bit reader moved on
#define HFZ_PAIR()                                                           \
  do {                                                                       \
    const uint64_t win_ = ((uint64_t)whi << 32) | wlo;                       \
    const uint32_t t1_ = (uint32_t)(win_ >> (sacc & 63));                    \
    const uint32_t e1_ = HFZ_ENTRY(t1_);                                     \
    sacc += e1_;                                                             \
    const uint32_t t2_ = (uint32_t)(win_ >> (sacc & 63));                    \
    const uint32_t e2_ = HFZ_ENTRY(t2_);                                     \
    sacc += e2_;                                                             \
    const uint32_t w1_ = e1_ >> 8, w2_ = e2_ >> 8;                           \
    const uint32_t y1_ = HFZ_SYMS(e1_, t1_, w1_);                            \
    const uint32_t y2_ = HFZ_SYMS(e2_, t2_, w2_);                            \
    const uint64_t sp_ = (uint64_t)(y1_ | (y2_ << (w1_ & 31))) << sh;        \
    const uint32_t lo_ = ocur | (uint32_t)sp_;                               \
    sh += w1_ + w2_;                                                         \
    *(uint32_t*)(ocol + (wa & kOutRowMask)) = lo_;                           \
    const uint32_t f_ = sh & 32;                                             \
    ocur = (uint32_t)(((sp_ & 0xffffffff00000000ull) | lo_) >> f_);          \
    HFZ_LSHL3_ADD(wa, f_);                                                   \
    sh &= 31;                                                                \
    HFZ_REFILL();                                                            \
  } while (0)
  // moves the ring block holding output offsets [nfb, nfb+32) into sb[]
#define HFZ_TAKE_BLOCK()                                                     \
  do {                                                                       \
    const uint32_t* s_ = orow32 + 64 * ((nfb & kDecBlk) >> 2);               \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) sb[q_] = s_[64 * q_];   \
    sb_at = nfb;                                                             \
    nfb += kDecBlk;                                                          \
  } while (0)
#define HFZ_TAKE_ODD_BLOCK()                                                 \
  do {                                                                       \
    const uint32_t* s_ = orow32 + 64 * ((nfb & kDecBlk) >> 2);               \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) sc[q_] = s_[64 * q_]; \
    sc_at = nfb;                                                             \
    nfb += kDecBlk;                                                          \
  } while (0)
#define HFZ_STORE_ODD_BLOCK()                                                \
  do {                                                                       \
    if (sc_at >= out_lo && sc_at + kDecBlk <= out_hi) {                      \
      *(uint4*)(obase + sc_at) = make_uint4(sc[0], sc[1], sc[2], sc[3]); \
      *(uint4*)(obase + sc_at + 16) = make_uint4(sc[4], sc[5], sc[6], sc[7]); \
    } else {                                                                 \
      store_block_clipped(obase, sc_at, sc, out_lo, out_hi);                 \
    }                                                                        \
  } while (0)
  // stores sb[]; only bytes of this stream's slice [out_lo, out_hi) are written
#define HFZ_STORE_BLOCK()                                                    \
  do {                                                                       \
    if (sb_at >= out_lo && sb_at + kDecBlk <= out_hi) {                      \
      *(uint4*)(obase + sb_at) = make_uint4(sb[0], sb[1], sb[2], sb[3]);     \
      *(uint4*)(obase + sb_at + 16) = make_uint4(sb[4], sb[5], sb[6], sb[7]); \
    } else {                                                                 \
      store_block_clipped(obase, sb_at, sb, out_lo, out_hi);                 \
    }                                                                        \
  } while (0)

  for (;;) {
    if (!__any(left > 0)) break;
    if (left > 0 && (int32_t)(ra_low - ra) < kReach) {  // out of staged input: next block
      // A reader that still owes symbols is above the stream area, so it only
      // asks for a block when fewer than kDecRoundDw dwords between the lowest
      // staged one and the area are left; anything else is a corrupt stream.
      if ((int32_t)(ra_low - ra_area) >= kReach) {
        stream_bad = true;
        left = 0;
      } else {
        low_blk -= kInBlk;
        ra_low += kBlkDw << 8;
        HFZ_STAGE_BLOCK(low_blk);
        HFZ_LOAD_BLOCK(low_blk - kInBlk);
      }
    }
    STAMP(3);  // staging
    for (;;) {
      const bool can = left > 0 && (int32_t)(ra_low - ra) >= kReach;
      if (!__any(can)) break;
      bool full = false;  // this round completed the block at nfb
      if (can) {
        const uint32_t wa_b = wa, sh_b = sh;
#pragma unroll
        for (uint32_t j = 0; j < kDecRound / 2; ++j) HFZ_PAIR();
        const uint32_t adv = wa - wa_b;  // 256 per dword = 64 per byte
        left -= (int32_t)(((adv >> 3) + sh - sh_b) >> 3);
        wp += adv >> 6;
        full = wp >= nfb + kDecBlk;
        // last round's block goes to memory, the one this round filled is read
        if (kStore == 2) {
          // handled after the round, by all lanes (below)
        } else if (!kPair64) {
          if (sb_full) HFZ_STORE_BLOCK();
          sb_full = full;
          if (full) HFZ_TAKE_BLOCK();
        } else {
          // an even block waits for the odd one above it; blocks are taken in
          // address order, so when both are held they are neighbours
          if (sc_full) {
            if (sb_full && sb_at >= out_lo && sc_at + kDecBlk <= out_hi) {
              *(uint4*)(obase + sb_at) = make_uint4(sb[0], sb[1], sb[2], sb[3]);
              *(uint4*)(obase + sb_at + 16) = make_uint4(sb[4], sb[5], sb[6], sb[7]);
              *(uint4*)(obase + sb_at + 32) = make_uint4(sc[0], sc[1], sc[2], sc[3]);
              *(uint4*)(obase + sb_at + 48) = make_uint4(sc[4], sc[5], sc[6], sc[7]);
            } else {
              if (sb_full) HFZ_STORE_BLOCK();
              HFZ_STORE_ODD_BLOCK();
            }
            sb_full = sc_full = false;
          }
          if (full) {
            if (nfb & kDecBlk) {
              HFZ_TAKE_ODD_BLOCK();
              sc_full = true;
            } else {
              HFZ_TAKE_BLOCK();
              sb_full = true;
            }
          }
        }
      }
      STAMP(5);  // lookups
      if (kStore == 2) {
        // A round adds at most 32 bytes, so at most one more block is complete.
        // A lane that completed one tells its neighbour where it goes; the two
        // lanes fetch one 16-byte half each from the owner's ring rows and store
        // them side by side.  First and last block of a slice (they may hold a
        // neighbour's bytes) are stored by their owner alone, byte-clipped.
        if (__ballot(full)) {
          wave_sync();  // ring rows written in the round are read by the neighbour lane
          const bool whole = nfb >= out_lo && nfb + kDecBlk <= out_hi;
          if (full && !whole) {
            HFZ_TAKE_BLOCK();  // advances nfb
            HFZ_STORE_BLOCK();
          }
          // (nfb | 1) if this lane hands a block over, else 0 (nfb is a multiple of 32)
          const uint32_t give = full && whole ? nfb | 1u : 0u;
          if (full && whole) nfb += kDecBlk;
          // what the even and the odd lane of the pair hand over: quad_perm [0,0,2,2] / [1,1,3,3]
          const uint32_t g_even = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0xA0, 0xf, 0xf, false);
          const uint32_t g_odd = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0xF5, 0xf, 0xf, false);
          const uint32_t half16 = lane & 1;  // which 16 bytes of the 32 this lane moves
          const uint32_t* const pair_rows = obuf_all + (lane & ~1u) + 64 * 4 * half16;
          uint4 v_even, v_odd;
          if (g_even) {
            const uint32_t* src = pair_rows + 64 * ((g_even & kDecBlk) >> 2);
            v_even = make_uint4(src[0], src[64], src[128], src[192]);
          }
          if (g_odd) {
            const uint32_t* src = pair_rows + 1 + 64 * ((g_odd & kDecBlk) >> 2);
            v_odd = make_uint4(src[0], src[64], src[128], src[192]);
          }
          if (g_even) *(uint4*)(obase + (g_even & ~1u) + 16 * half16) = v_even;
          if (g_odd) *(uint4*)(obase + (g_odd & ~1u) + 16 * half16) = v_odd;
        }
      }
      STAMP(6);  // pair stores
    }
    STAMP(4);  // leaving the rounds
  }
  if (kStore != 2 && sb_full) HFZ_STORE_BLOCK();
  if (kPair64 && sc_full) HFZ_STORE_ODD_BLOCK();
  // the last partial dword, then what is left in the ring: up to two blocks
  *(uint32_t*)(ocol + (wa & kOutRowMask)) = ocur;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (nfb < out_hi) {
      HFZ_TAKE_BLOCK();
      HFZ_STORE_BLOCK();
    }
  }
#undef HFZ_REFILL
#undef HFZ_LSHL3_ADD
#undef HFZ_ENTRY
#undef HFZ_SYMS
#undef HFZ_PAIR
#undef HFZ_TAKE_BLOCK
#undef HFZ_STORE_BLOCK
#undef HFZ_TAKE_ODD_BLOCK
#undef HFZ_STORE_ODD_BLOCK
#undef HFZ_LOAD_BLOCK
#undef HFZ_STAGE_BLOCK
  STAMP(7);  // tail
  STAMP_FLUSH(0);
  if (kParts) {  // the verdict of all waves of the frame
    if (__ballot(stream_bad) != 0 && lane == 0) atomicOr(wg_bad_p, 1);
    __syncthreads();
    if (tid == 0 && status) status[f] = *wg_bad_p ? HFZ_E_CORRUPT : HFZ_OK;
    return;
  }
  if (status) {
    const uint64_t bad_lanes = __ballot(stream_bad);
    const bool any_bad = kF == 1 ? bad_lanes != 0 : ((bad_lanes >> (32 * half)) & 0xffffffffull) != 0;
    if (k == 0) status[f] = any_bad ? HFZ_E_CORRUPT : HFZ_OK;
  }
}

// ---------------------------------------------------------------------------
// Synthetic source: inverse-CDF sampling of a 256-symbol distribution from a
// counter-based splitmix64 stream; integer arithmetic only.
__device__ __forceinline__ uint32_t sample_byte(const uint32_t* cdf, uint64_t seed, uint64_t idx) {
  uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const uint32_t u = (uint32_t)(z >> 32);
  uint32_t lo = 0, hi = 255;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const uint32_t mid = (lo + hi) >> 1;
    const bool right = lo < hi && cdf[mid] <= u;
    const bool left = lo < hi && !right;
    lo = right ? mid + 1 : lo;
    hi = left ? mid : hi;
  }
  return lo;
}

__global__ __launch_bounds__(256) void generate_kernel(const uint32_t* __restrict__ cdf_g,
                                                       uint64_t seed, uint64_t first_index,
                                                       uint8_t* __restrict__ out, uint64_t n) {
  __shared__ uint32_t cdf[256];
  cdf[threadIdx.x] = cdf_g[threadIdx.x];
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * 256 * 16;
  for (uint64_t base = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16; base < n; base += stride) {
    const uint64_t left = n - base;
    if (left >= 16 && (((uintptr_t)(out + base)) & 15) == 0) {
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          x |= sample_byte(cdf, seed, first_index + base + 4 * j + k) << (8 * k);
        w[j] = x;
      }
      *(uint4*)(out + base) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      const uint32_t m = left < 16 ? (uint32_t)left : 16;
      for (uint32_t j = 0; j < m; ++j)
        out[base + j] = (uint8_t)sample_byte(cdf, seed, first_index + base + j);
    }
  }
}

inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

}  // namespace

std::atomic<int> g_decoder_table_bits{0};
std::atomic<int> g_frames_per_wave{0};
std::atomic<int> g_encoder_block{0};
std::atomic<int> g_encoder_parts{0};
std::atomic<int> g_decoder_store{0};
std::atomic<int> g_decoder_parts{0};
std::atomic<int> g_decoder_input_block{0};

// ---------------------------------------------------------------------------
// launchers

// A launch whose kernel is timed hands the span's events to the runtime with the
// dispatch (they receive the kernel's own begin and end timestamps).
#define HFZ_LAUNCH(span, kernel, grid, block, lds, stream, ...)                                    \
  do {                                                                                             \
    const KernelSpan* sp_ = (span);                                                                \
    if (sp_ && (sp_->begin || sp_->end))                                                           \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, sp_->begin, sp_->end, 0, __VA_ARGS__); \
    else                                                                                           \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                           \
  } while (0)

// accumulate: the counts are added to what d_hist holds (the caller cleared it).
int launch_histogram(const uint8_t* d_raw, uint64_t total, uint64_t frame_bytes,
                     uint64_t n_frames, uint32_t* d_hist, void* stream, bool accumulate,
                     const KernelSpan* span) {
  if (n_frames == 0) return hipSuccess;
  if (frame_bytes > 0xffffffffull) return hipErrorInvalidValue;  // frame_len() is 32 bits wide
  uint64_t bpf = (frame_bytes + (1u << 16) - 1) >> 16;  // ~64 KiB of input per block
  if (bpf < 1) bpf = 1;
  if (bpf > 65536) bpf = 65536;
  if (n_frames * bpf > 0x7fffffffull) return hipErrorInvalidValue;
  if (bpf > 1 && !accumulate) {
    hipError_t e = hipMemsetAsync(d_hist, 0, n_frames * 256 * sizeof(uint32_t), as_stream(stream));
    if (e != hipSuccess) return e;
  }
  HFZ_LAUNCH(span, hist_kernel, dim3((uint32_t)(n_frames * bpf)), dim3(256), 0,
                     as_stream(stream), d_raw, total, frame_bytes, (uint32_t)bpf,
                     accumulate ? 1u : 0u, d_hist);
  return hipGetLastError();
}

int launch_build_codes(const uint32_t* d_hist, uint64_t n_frames, FrameCode* d_code, void* stream) {
  if (n_frames == 0) return hipSuccess;
  if (n_frames > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(code_kernel, dim3((uint32_t)n_frames), dim3(64), 0, as_stream(stream), d_hist,
                     d_code);
  return hipGetLastError();
}

// d_hist != nullptr: builds d_code[f] from d_hist[f]; otherwise d_code is given,
// one record per frame or (shared_code) one for all.
int launch_stream_sizes(int K, const uint8_t* d_raw, uint64_t total, uint64_t frame_bytes,
                        uint64_t n_frames, const uint32_t* d_hist, FrameCode* d_code,
                        bool shared_code, uint32_t* d_stream_end, uint32_t* d_csize, void* stream,
                        const KernelSpan* span) {
  if (n_frames == 0) return hipSuccess;
  if (n_frames > 0x7fffffffull) return hipErrorInvalidValue;
  if (d_hist && shared_code) return hipErrorInvalidValue;
  HFZ_LAUNCH(span, sizes_kernel, dim3((uint32_t)n_frames), dim3(64), 0, as_stream(stream),
                     (uint32_t)K, d_raw, total, frame_bytes, d_hist, d_code,
                     (uint32_t)(d_hist != nullptr), shared_code ? 0u : 1u, d_stream_end, d_csize);
  return hipGetLastError();
}

int launch_count(int K, const uint8_t* d_raw, uint64_t total, uint64_t frame_bytes,
                 uint64_t n_frames, const FrameCode* d_code, bool shared_code,
                 uint32_t* d_counts_out, uint32_t* d_hist_out, uint32_t* d_stream_end,
                 uint32_t* d_csize, void* stream, const KernelSpan* span, uint32_t split_log2) {
  if (n_frames == 0) return hipSuccess;
  if (n_frames > 0x7fffffffull || (n_frames << split_log2) > 0x7fffffffull) return hipErrorInvalidValue;
  if (!count_kernel_fits(K, frame_bytes)) return hipErrorInvalidValue;
  if (d_counts_out ? d_hist_out == nullptr : d_code == nullptr) return hipErrorInvalidValue;
  if (split_log2 && !d_counts_out) return hipErrorInvalidValue;  // pieces only on the way to code_sizes_split
  HFZ_LAUNCH(span, count_kernel, dim3((uint32_t)(n_frames << split_log2)), dim3(kCntThreads), 0,
                     as_stream(stream), (uint32_t)K, d_raw, total, frame_bytes, d_code,
                     shared_code ? 0u : 1u, d_counts_out, d_hist_out, d_stream_end, d_csize,
                     split_log2);
  return hipGetLastError();
}

int launch_code_sizes_split(int K, const uint32_t* d_hist_u, const uint32_t* d_counts_u,
                            uint64_t n_frames, uint32_t split_log2, FrameCode* d_code,
                            uint32_t* d_part_first, uint32_t* d_stream_end, uint32_t* d_csize,
                            void* stream, const KernelSpan* span) {
  if (n_frames == 0) return hipSuccess;
  if (n_frames > 0x7fffffffull || split_log2 == 0 || split_log2 > kMaxSplitLog2) return hipErrorInvalidValue;
  HFZ_LAUNCH(span, code_sizes_split_kernel, dim3((uint32_t)n_frames), dim3(64u << split_log2), 0, as_stream(stream),
                     (uint32_t)K, d_hist_u, d_counts_u, d_code, split_log2, d_part_first,
                     d_stream_end, d_csize);
  return hipGetLastError();
}

int device_cus();
int launch_code_sizes(int K, const uint32_t* d_hist, const uint32_t* d_counts, uint64_t n_frames,
                      FrameCode* d_code, const FrameCode* d_given, bool shared_code,
                      uint32_t* d_stream_end, uint32_t* d_csize, void* stream,
                      const KernelSpan* span) {
  if (n_frames == 0) return hipSuccess;
  if (n_frames > 0x7fffffffull) return hipErrorInvalidValue;
  if (!d_given && n_frames <= 8 * (uint64_t)device_cus())
    HFZ_LAUNCH(span, code_sizes_kernel_wide, dim3((uint32_t)n_frames), dim3(64), 0, as_stream(stream),
                       (uint32_t)K, d_hist, d_counts, d_code, d_given, shared_code ? 0u : 1u,
                       d_stream_end, d_csize);
  else
    HFZ_LAUNCH(span, code_sizes_kernel, dim3((uint32_t)n_frames), dim3(64), 0, as_stream(stream),
                       (uint32_t)K, d_hist, d_counts, d_code, d_given, shared_code ? 0u : 1u,
                       d_stream_end, d_csize);
  return hipGetLastError();
}

int launch_hist_total(const uint32_t* d_hist, uint64_t n_frames, uint64_t* d_total, void* stream,
                      const KernelSpan* span) {
  hipError_t e = hipMemsetAsync(d_total, 0, 256 * sizeof(uint64_t), as_stream(stream));
  if (e != hipSuccess || n_frames == 0) return e;
  const uint32_t blocks = n_frames < 1024 ? (uint32_t)n_frames : 1024u;
  HFZ_LAUNCH(span, hist_total_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), d_hist,
                     n_frames, (unsigned long long*)d_total);
  return hipGetLastError();
}

int launch_frame_offsets(const uint32_t* d_csize, uint64_t n_frames, uint64_t comp_stride,
                         uint64_t comp_cap, uint64_t* d_comp_off, uint32_t* d_comp_size,
                         int32_t* d_status, void* stream, const KernelSpan* span) {
  if (n_frames == 0) return hipSuccess;
  HFZ_LAUNCH(span, offsets_kernel, dim3(1), dim3(1024), 0, as_stream(stream), d_csize, n_frames,
                     comp_stride, comp_cap, d_comp_off, d_comp_size, d_status);
  return hipGetLastError();
}

int device_cus();
// two frames per wave once the pairs still give every SIMD a wave
bool pair_frames(int K, uint64_t n_frames) {
  const int fpw = g_frames_per_wave.load(std::memory_order_relaxed);
  if (K > 32 || fpw == 1) return false;
  return fpw == 2 || n_frames > 4 * (uint64_t)device_cus();
}

int device_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  return cus;
}

// LDS a workgroup may use on the current device (160 KiB on gfx950, the only
// target of this code object; asked for all the same, so that a launch is sized
// to what the device says rather than to a constant).
size_t device_lds_per_block() {
  int dev = 0, bytes = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&bytes, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess ||
      bytes <= 0)
    bytes = 64 << 10;
  return (size_t)bytes;
}

// Parts per stream (as a power of two) for a batch: none while one wave per
// frame fills the chip four waves per CU, else as many as it takes to get there,
// within 512 threads per frame and no less than 512 bytes per part.
uint32_t encoder_log2_parts(int K, uint64_t n_frames, uint64_t frame_bytes) {
  const int forced = g_encoder_parts.load(std::memory_order_relaxed);
  uint32_t max_log2 = 0;
  while (max_log2 < 6 && ((uint64_t)K << (max_log2 + 1)) <= 512) ++max_log2;
  if (forced > 0) {
    uint32_t l = 0;
    while ((2u << l) <= (uint32_t)forced) ++l;
    return l < max_log2 ? l : max_log2;
  }
  const uint64_t want_threads = 4ull * 64 * (uint64_t)device_cus();
  uint32_t l = 0;
  while (l < max_log2 && (n_frames * (uint64_t)K << l) < want_threads &&
         frame_bytes / ((uint64_t)K << (l + 1)) >= 512)
    ++l;
  return l;
}

// Threads per stream (as a power of two) in the decoder: none once one wave per
// frame fills the chip, else as many as it takes to get there, within 512
// threads per frame, the parts of a stream inside one wave, and no less than 256
// bytes of output per part.
uint32_t decoder_log2_parts(int K, uint64_t n_frames, uint64_t frame_bytes) {
  const int forced = g_decoder_parts.load(std::memory_order_relaxed);
  uint32_t max_log2 = 0;
  while (max_log2 < 6 && ((uint64_t)K << (max_log2 + 1)) <= 512) ++max_log2;
  if (forced > 0) {
    uint32_t l = 0;
    while ((2u << l) <= (uint32_t)forced) ++l;
    return l < max_log2 ? l : max_log2;
  }
  // Finding the pieces costs a pass of its own (about half a decoding pass), so two
  // threads per stream never pay: parts only where at least four still leave the
  // chip with no more than its eight waves per CU (a CU's LDS pipe is busy with
  // eight; more waves on it do not decode faster).
  const uint64_t want_threads = 8ull * 64 * (uint64_t)device_cus();
  uint32_t l = 0;
  while (l < max_log2 && (n_frames * (uint64_t)K << (l + 1)) <= want_threads &&
         frame_bytes / ((uint64_t)K << (l + 1)) >= 256)
    ++l;
  return l >= 2 ? l : 0;
}

// Threads per stream (as a power of two) that the encoder will use on this batch:
// none with two frames per wave, else encoder_log2_parts() cut down to what the
// device's LDS holds (the rings of a parts launch: 8 KiB per wave of dynamic LDS,
// beyond the 64 KiB a launch may ask for by default when all eight waves are used,
// next to 2 KiB of static LDS).
uint32_t encoder_parts_plan(int K, uint64_t n_frames, uint64_t frame_bytes) {
  if (pair_frames(K, n_frames)) return 0;
  uint32_t log2_parts = encoder_log2_parts(K, n_frames, frame_bytes);
  for (;; --log2_parts) {
    const uint32_t threads = log2_parts ? (((uint32_t)K << log2_parts) + 63) & ~63u : 64u;
    const size_t ring_bytes = log2_parts ? (threads / 64) * 64 * kEncRingDw * sizeof(uint32_t) : 0;
    if (log2_parts == 0 || ring_bytes + 2048 <= device_lds_per_block()) break;
  }
  return log2_parts;
}

// The most pieces per stream (as a power of two) that a batch of n_frames frames
// can be counted in, whatever its frame size: what hfz_workspace_bytes provides for.
uint32_t max_split_log2(int K, uint64_t n_frames) {
  if (K > 32 ? false : g_frames_per_wave.load(std::memory_order_relaxed) == 2) return 0;
  uint32_t max_log2 = 0;
  while (max_log2 < kMaxSplitLog2 && ((uint64_t)K << (max_log2 + 1)) <= 512) ++max_log2;
  const uint64_t want_threads = 4ull * 64 * (uint64_t)device_cus();
  uint32_t l = 0;
  while (l < max_log2 && (n_frames * (uint64_t)K << l) < want_threads) ++l;
  return l;
}

// plan_log2_parts: what encoder_parts_plan() said when the batch was counted (the
// caller asked once, so that counting and encoding agree); d_part_first: first bit
// of every part from code_sizes_split_kernel, or null (the parts add up their
// code lengths themselves).
int launch_encode(int K, const uint8_t* d_raw, uint64_t total, uint64_t frame_bytes,
                  uint64_t n_frames, const FrameCode* d_code, bool shared_code,
                  const uint32_t* d_stream_end, const uint64_t* d_comp_off,
                  const uint32_t* d_comp_size, uint8_t* d_comp, uint64_t comp_cap, void* stream,
                  const KernelSpan* span, int plan_log2_parts, const uint32_t* d_part_first) {
  if (n_frames == 0) return hipSuccess;
  if (n_frames > 0x7fffffffull) return hipErrorInvalidValue;
  const int enc_blk = g_encoder_block.load(std::memory_order_relaxed);
  const bool wide = enc_blk ? enc_blk == 64
                            : n_frames > 4 * (uint64_t)device_cus();  // > one wave per SIMD
  const uint32_t log2_parts = plan_log2_parts >= 0 ? (uint32_t)plan_log2_parts
                                                   : encoder_parts_plan(K, n_frames, frame_bytes);
  const bool pair = log2_parts == 0 && pair_frames(K, n_frames);
  if (log2_parts == 0) d_part_first = nullptr;
  const dim3 grid((uint32_t)(pair ? (n_frames + 1) / 2 : n_frames));
  const uint32_t threads = log2_parts ? (((uint32_t)K << log2_parts) + 63) & ~63u : 64u;
  const size_t ring_bytes = log2_parts ? (threads / 64) * 64 * kEncRingDw * sizeof(uint32_t) : 0;
#define HFZ_LAUNCH_ENCODE(F, B, P)                                                               \
  do {                                                                                           \
    if (ring_bytes > (48u << 10)) {                                                              \
      hipError_t e_ = hipFuncSetAttribute((const void*)&encode_kernel<F, B, P>,                  \
                                          hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                          (int)ring_bytes);                                      \
      if (e_ != hipSuccess) return e_;                                                           \
    }                                                                                            \
    HFZ_LAUNCH(span, (encode_kernel<F, B, P>), grid, dim3(threads), ring_bytes,                  \
                       as_stream(stream), (uint32_t)K, d_raw, total, frame_bytes, n_frames,      \
                       d_code, shared_code ? 0u : 1u, d_stream_end, d_comp_off, d_comp_size,     \
                       d_comp, comp_cap, log2_parts, d_part_first);                              \
  } while (0)
  if (log2_parts && wide) HFZ_LAUNCH_ENCODE(1, 64, true);
  else if (log2_parts) HFZ_LAUNCH_ENCODE(1, 32, true);
  else if (pair && wide) HFZ_LAUNCH_ENCODE(2, 64, false);
  else if (pair) HFZ_LAUNCH_ENCODE(2, 32, false);
  else if (wide) HFZ_LAUNCH_ENCODE(1, 64, false);
  else HFZ_LAUNCH_ENCODE(1, 32, false);
#undef HFZ_LAUNCH_ENCODE
  return hipGetLastError();
}

constexpr uint32_t kSharedWaves = 3;  // frames per workgroup when they share one code
// Lane pairs store together while the streams of a frame lie at most this far
// apart (measured with 64 streams: 128 KiB frames 11-14 % faster that way from
// 128 MiB to 4 GiB, 256 KiB frames and larger 4-11 % slower).
constexpr uint64_t kPairStoreMaxSlice = 2048;
constexpr uint64_t kSharedMaxFrame = 128u << 10;


int launch_decode(int K, const uint8_t* d_comp, uint64_t comp_bytes, const uint64_t* d_comp_off,
                  const uint32_t* d_comp_size, uint64_t n_frames, bool shared_code, uint8_t* d_raw,
                  uint64_t raw_cap, uint64_t frame_bytes, int32_t* d_status, void* stream,
                  const KernelSpan* span) {
  if (n_frames == 0) return hipSuccess;
  if (n_frames > 0x7fffffffull) return hipErrorInvalidValue;
  // 12-bit table (28 KiB of LDS, 5 frames per CU) while that holds every frame
  // at once; beyond that the 11-bit one (20 KiB, 8 per CU), whose 12th-bit pick
  // costs less than the extra resident waves return.
  const int cus = device_cus();
  const int tab_bits = g_decoder_table_bits.load(std::memory_order_relaxed);
  // Diagnostic only: HFZ_DECODER_LDS_PAD=<bytes> in the environment makes every
  // one-wave decoder workgroup ask for that much unused LDS on top of its own, so
  // that fewer of them share a CU (occupancy experiments); read once.
  static const size_t lds_pad = [] {
    const char* v = getenv("HFZ_DECODER_LDS_PAD");
    const long n = v ? atol(v) : 0;
    return (size_t)(n > 0 && n <= (140 << 10) ? n : 0);
  }();
#define HFZ_LAUNCH_DECODE_BLK(BITS, WAVES, F, STORE, BLK)                                         \
  HFZ_LAUNCH(span, (decode_kernel<BITS, WAVES, F, STORE, false, BLK>),                            \
             dim3((uint32_t)((n_frames + F - 1) / F)), dim3(64 * WAVES), lds_pad, as_stream(stream), \
             (uint32_t)K, d_comp, comp_bytes, d_comp_off, d_comp_size, n_frames, d_raw, raw_cap,  \
             frame_bytes, d_status, 0u)
#define HFZ_LAUNCH_DECODE(BITS, WAVES, F, STORE) HFZ_LAUNCH_DECODE_BLK(BITS, WAVES, F, STORE, 64)
  // input blocks of 32 bytes (a 64-byte input ring per lane: 16 KiB of LDS per
  // frame with the 11-bit table, ten frames per CU instead of eight) only on request
  const bool small_blocks = g_decoder_input_block.load(std::memory_order_relaxed) == 32;
  if (tab_bits == HFZ_DECODER_TABLE_FREE) {
    // only on request (tests, comparison): one symbol per step, no lookup table
    HFZ_LAUNCH_DECODE(0, 1, 1, 0);
    return hipGetLastError();
  }
  if (shared_code && frame_bytes <= kSharedMaxFrame) {
    // one 12-bit table per workgroup of kSharedWaves frames: 16 + 3 * 12 KiB of
    // LDS, three workgroups (nine frames) per CU and no sibling entries.  Pays
    // off while building a table is a visible part of a frame's work; larger
    // frames go through the per-frame kernels below, which are faster there
    // (one-wave workgroups pack the CUs better) and decode any frame.
    const uint64_t groups = (n_frames + kSharedWaves - 1) / kSharedWaves;
    HFZ_LAUNCH(span, (decode_kernel<12, kSharedWaves, 1, 0, false>), dim3((uint32_t)groups),
                       dim3(64 * kSharedWaves), 0, as_stream(stream), (uint32_t)K, d_comp,
                       comp_bytes, d_comp_off, d_comp_size, n_frames, d_raw, raw_cap, frame_bytes,
                       d_status, 0u);
    return hipGetLastError();
  }
  // few frames: several threads per stream, one workgroup per frame
  uint32_t log2_parts = decoder_log2_parts(K, n_frames, frame_bytes);
  uint32_t threads = 64;
  size_t ring_bytes = 0;
  for (;; --log2_parts) {
    threads = log2_parts ? (((uint32_t)K << log2_parts) + 63) & ~63u : 64u;
    ring_bytes = (size_t)(threads / 64) * 64 * (kDecInDw + kDecOutDw) * sizeof(uint32_t);
    if (log2_parts == 0 || ring_bytes + (32u << 10) <= device_lds_per_block()) break;  // 28.2 KiB of static LDS
  }
  if (log2_parts) {
#define HFZ_LAUNCH_DECODE_PARTS(BITS, STORE)                                                      \
  do {                                                                                            \
    if (ring_bytes > (32u << 10)) { /* beyond 64 KiB together with the static LDS */              \
      hipError_t e_ = hipFuncSetAttribute((const void*)&decode_kernel<BITS, 1, 1, STORE, true>,   \
                                          hipFuncAttributeMaxDynamicSharedMemorySize,             \
                                          (int)ring_bytes);                                       \
      if (e_ != hipSuccess) return e_;                                                            \
    }                                                                                             \
    HFZ_LAUNCH(span, (decode_kernel<BITS, 1, 1, STORE, true>), dim3((uint32_t)n_frames),          \
               dim3(threads), ring_bytes, as_stream(stream), (uint32_t)K, d_comp, comp_bytes,     \
               d_comp_off, d_comp_size, n_frames, d_raw, raw_cap, frame_bytes, d_status,          \
               log2_parts);                                                                       \
  } while (0)
    // one table for all waves of the frame: the 12-bit one unless the other is asked
    // for; lane pairs (neighbouring pieces of a stream) store together while a
    // piece is short, as the lanes of short streams do
    const int store = g_decoder_store.load(std::memory_order_relaxed);
    const bool pair_store = store ? store == 2
                                  : frame_bytes <= kPairStoreMaxSlice * ((uint64_t)K << log2_parts);
    if (tab_bits == 11 && pair_store) HFZ_LAUNCH_DECODE_PARTS(11, 2);
    else if (tab_bits == 11) HFZ_LAUNCH_DECODE_PARTS(11, 0);
    else if (pair_store) HFZ_LAUNCH_DECODE_PARTS(12, 2);
    else HFZ_LAUNCH_DECODE_PARTS(12, 0);
#undef HFZ_LAUNCH_DECODE_PARTS
    return hipGetLastError();
  }
  if (tab_bits != 12 && pair_frames(K, n_frames)) {
    // two frames per wave, 11-bit tables: 28 KiB of LDS, 5 waves = 10 frames per CU
    if (small_blocks) HFZ_LAUNCH_DECODE_BLK(kSmallTabBits, 1, 2, 0, 32);
    else HFZ_LAUNCH_DECODE(kSmallTabBits, 1, 2, 0);
    return hipGetLastError();
  }
  const bool wide = tab_bits ? tab_bits == 12 : n_frames <= (uint64_t)cus * 5;
  const bool pair_store = g_decoder_store.load(std::memory_order_relaxed)
                              ? g_decoder_store.load(std::memory_order_relaxed) == 2
                              : frame_bytes <= kPairStoreMaxSlice * (uint64_t)K;
  if (wide)
    HFZ_LAUNCH_DECODE(12, 1, 1, 0);
  else if (pair_store && small_blocks)
    HFZ_LAUNCH_DECODE_BLK(kSmallTabBits, 1, 1, 2, 32);
  else if (pair_store)
    HFZ_LAUNCH_DECODE(kSmallTabBits, 1, 1, 2);
  else if (small_blocks)
    HFZ_LAUNCH_DECODE_BLK(kSmallTabBits, 1, 1, 1, 32);
  else
    HFZ_LAUNCH_DECODE(kSmallTabBits, 1, 1, 1);
#undef HFZ_LAUNCH_DECODE
#undef HFZ_LAUNCH_DECODE_BLK
  return hipGetLastError();
}

int launch_generate(const uint32_t* d_cdf, uint64_t seed, uint64_t first_index, uint8_t* d_out,
                    uint64_t n, void* stream) {
  if (n == 0) return hipSuccess;
  uint64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(generate_kernel, dim3((uint32_t)blocks), dim3(256), 0, as_stream(stream),
                     d_cdf, seed, first_index, d_out, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The batch encoder without a caller-provided workspace and without any
// allocation: for callers that must not allocate (code under HIP graph capture,
// tight serving loops).  The workspace (histograms, codes, stream ends, frame
// sizes, per-stream counters) lives in a device array of this code object,
// sized for kStaticMaxFrames frames; every GPU the library runs on holds its own
// copy, at its own address, until the process ends.
constexpr size_t kStaticMaxFrames = HFZ_STATIC_MAX_FRAMES;
constexpr size_t kStaticBytes =
    ((kStaticMaxFrames * (256 * 4 + sizeof(FrameCode) + 64 * 4 + 4 + 256 * 32 * 4)) + 5 * 256 + 255) &
    ~size_t(255);
static __device__ __align__(256) uint8_t scratch[kStaticBytes];

namespace {
// `scratch` in host code is only a handle for the runtime, not a device
// address; the address on each device comes from hipGetSymbolAddress, once per
// device, and is what the kernels get.
constexpr int kMaxDevices = 64;
std::atomic<uint8_t*> g_scratch_ptr[kMaxDevices];

int static_workspace(uint8_t** out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_failure((int)e);
  if (dev < 0 || dev >= kMaxDevices) return HFZ_E_ARG;
  uint8_t* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    void* sym = nullptr;
    e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));  // the current device's copy
    if (e != hipSuccess) return hip_failure((int)e);
    if (!sym || ((uintptr_t)sym & 255) != 0) return HFZ_E_HIP;
    p = (uint8_t*)sym;
    g_scratch_ptr[dev].store(p, std::memory_order_release);  // racing threads store the same value
  }
  *out = p;
  return HFZ_OK;
}
}  // namespace

extern "C" int hfz_static_workspace_init(void) {
  uint8_t* p = nullptr;
  return static_workspace(&p);
}

extern "C" size_t hfz_static_max_frames(void) { return kStaticMaxFrames; }

extern "C" int hfz_compress_batch_static_dev(int K, const void* d_raw, size_t total,
                                             size_t frame_bytes, void* d_comp, size_t comp_cap,
                                             size_t comp_stride, uint64_t* d_comp_off,
                                             uint32_t* d_comp_size, int32_t* d_status,
                                             void* stream) {
  if (frame_bytes == 0) return HFZ_E_ARG;
  const size_t nf = total == 0 ? 1 : (total + frame_bytes - 1) / frame_bytes;
  if (nf > kStaticMaxFrames) return HFZ_E_TOO_LARGE;  // more frames than the static workspace holds
  static_assert(kStaticBytes >= 256, "workspace");
  if (workspace_layout(nf).total > kStaticBytes) return HFZ_E_TOO_LARGE;
  uint8_t* ws = nullptr;
  const int rc = static_workspace(&ws);
  if (rc != HFZ_OK) return rc;
  // Every workspace word a kernel reads in this call is written earlier in the
  // same call (the separate-pass histogram clears its own counters with a
  // memset on the stream), so nothing left by the previous call is looked at.
  return compress_batch(K, d_raw, total, frame_bytes, d_comp, comp_cap, comp_stride, d_comp_off,
                        d_comp_size, d_status, ws, kStaticBytes, nullptr, stream);
}

#ifdef HFZ_STAMP
extern "C" int hfz_debug_stamps(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32);
  if (e == hipSuccess && reset) {
    unsigned long long z[32] = {};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
  }
  return (int)e;
}
#endif

}  // namespace hfz
