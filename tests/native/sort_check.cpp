// Checks the oracle's restated libstdc++ introsort against std::sort itself,
// on the call shape of codec/huffman.cpp:342-354 (symbols ascending, ordered
// by count descending).  Exit status 0 = all equal.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../oracle/huffman_oracle.h"

static int check(const uint32_t* count, const char* what, long iter) {
  uint8_t a[256], b[256];
  int n = 0;
  for (int c = 0; c < 256; ++c)
    if (count[c]) a[n++] = uint8_t(c);
  memcpy(b, a, sizeof(a));
  std::sort(a, a + n, [&](uint8_t x, uint8_t y) { return count[x] > count[y]; });
  ho_sort_syms_desc(b, n, count);
  if (memcmp(a, b, size_t(n)) != 0) {
    printf("MISMATCH %s iter %ld n=%d\n", what, iter, n);
    return 1;
  }
  return 0;
}

// Median-of-3 killer (Musser): drives introsort into its heapsort fallback.
static void killer(uint32_t* count, int n) {
  // values for ascending-order sort; we sort descending by count, so invert.
  std::vector<uint32_t> v(n);
  int k = n / 2;
  for (int i = 1; i <= k; ++i) {
    if (i % 2 == 1) {
      v[i - 1] = i;
      v[i] = k + i;
    }
    v[k + i - 1] = 2 * i;
  }
  for (int i = 0; i < n; ++i) count[i] = 100000u - v[i];
}

int main() {
  std::mt19937_64 rng(12345);
  uint32_t count[256];
  int bad = 0;
  long heap_like = 0;
  for (long it = 0; it < 200000 && !bad; ++it) {
    int mode = it % 8;
    int nsym = 1 + int(rng() % 256);
    memset(count, 0, sizeof(count));
    for (int i = 0; i < nsym; ++i) {
      int c = int(rng() % 256);
      uint32_t v;
      switch (mode) {
        case 0: v = 1 + uint32_t(rng() % 3); break;            // very tie-heavy
        case 1: v = 1 + uint32_t(rng() % 16); break;
        case 2: v = 1 + uint32_t(rng() % 1000); break;
        case 3: v = 1u << (rng() % 20); break;                 // powers of two
        case 4: v = 1 + uint32_t(rng() >> 40); break;          // mostly distinct
        case 5: v = 1 + uint32_t(c / 8); break;                // ascending runs
        case 6: v = 300 - uint32_t(c); break;                  // already sorted
        default: v = 7; break;                                 // all equal
      }
      count[c] = v;
    }
    bad |= check(count, "random", it);
  }
  for (int n = 17; n <= 256 && !bad; ++n) {
    memset(count, 0, sizeof(count));
    killer(count, n & ~1);
    bad |= check(count, "killer", n);
    ++heap_like;
    // organ pipe
    for (int i = 0; i < n; ++i) count[i] = 1000u + uint32_t(i < n / 2 ? i : n - i);
    bad |= check(count, "organ", n);
    for (int i = 0; i < n; ++i) count[i] = 1000u + uint32_t(n - i);
    bad |= check(count, "desc", n);
    for (int i = 0; i < n; ++i) count[i] = 1000u + uint32_t(i);
    bad |= check(count, "asc", n);
  }
  printf(bad ? "FAIL\n" : "OK\n");
  return bad;
}
