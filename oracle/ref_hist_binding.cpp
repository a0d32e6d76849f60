// TEST INFRASTRUCTURE ONLY.  C entry points over the reference's own byte
// histogram functions, so that tests can call them through ctypes.  This file
// is a caller of the reference, nothing else: the functions themselves are
// compiled from the unmodified /root/reference/codec/histogram.cpp by
// oracle/Makefile (target `ref`), into oracle/_ref/libref_hist.so.
//
// variant: 0 MakeHistogram (dispatcher, codec/histogram.cpp:193-201),
//          1 MakeHistogramSimple (:184-191), 2 MakeHistogramMulti (:144-182),
//          3 MakeHistogramVectorized (:14-92),
//          4 MakeHistogramGatherScatter (:95-142; AVX-512, check the CPU first).
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string_view>

#include "codec/histogram.h"

extern "C" int ref_histogram(int variant, const void* src, size_t n, uint32_t out[256]) {
  const std::string_view text(static_cast<const char*>(src), n);
  huffman::ByteHistogram h;
  switch (variant) {
    case 0: h = huffman::MakeHistogram(text); break;
    case 1: h = huffman::MakeHistogramSimple(text); break;
    case 2: h = huffman::MakeHistogramMulti(text); break;
    case 3: h = huffman::MakeHistogramVectorized(text); break;
    case 4: h = huffman::MakeHistogramGatherScatter(text); break;
    default: return -1;
  }
  std::memcpy(out, h.data(), sizeof(uint32_t) * 256);
  return 0;
}
