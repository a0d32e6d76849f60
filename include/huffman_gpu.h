// C++ mirror of the reference's compressor interface (codec/huffman.h) on top
// of the hfz C ABI.  Same names, argument meaning and ownership as
//   template <int K> std::string CompressMulti(std::string_view raw);      (codec/huffman.h:9-10)
//   template <int K> std::string DecompressMulti(std::string_view comp);   (codec/huffman.h:11-12)
//   template <int K> class HuffmanCompressorMulti { Compress, Decompress, name }  (codec/huffman.h:42-52)
// so the reference's typed test suite (codec/huffman_test.cpp:47-54) and
// benchmark templates (codec/huffman_benchmark.cpp:61-81) accept
// HuffmanCompressorGpu<K> unchanged.  Error behaviour: the reference aborts on
// failed assertions; these throw std::runtime_error carrying the hfz code.
#pragma once

#include <stdexcept>
#include <string>
#include <string_view>

#include "hfz.h"

namespace huffman {

namespace detail {
inline void check(int rc, const char* where) {
  if (rc != HFZ_OK)
    throw std::runtime_error(std::string(where) + ": " + hfz_strerror(rc) + " (" +
                             std::to_string(rc) + ") " + hfz_last_hip_error());
}
}  // namespace detail

template <int K>
std::string CompressMultiGpu(std::string_view raw) {
  static_assert(K >= 1 && K <= HFZ_MAX_STREAMS, "K must be in 1..64");
  std::string out(hfz_compress_bound(raw.size(), K), '\0');
  size_t n = 0;
  detail::check(hfz_compress(K, raw.data(), raw.size(), out.data(), out.size(), &n),
                "hfz_compress");
  out.resize(n);
  return out;
}

template <int K>
std::string DecompressMultiGpu(std::string_view compressed) {
  static_assert(K >= 1 && K <= HFZ_MAX_STREAMS, "K must be in 1..64");
  size_t raw_size = 0;
  detail::check(hfz_decompressed_size(compressed.data(), compressed.size(), &raw_size),
                "hfz_decompressed_size");
  std::string out(raw_size, '\0');
  size_t n = 0;
  detail::check(hfz_decompress(K, compressed.data(), compressed.size(), out.data(), out.size(), &n),
                "hfz_decompress");
  out.resize(n);
  return out;
}

template <int K>
class HuffmanCompressorGpu {
 public:
  static std::string Compress(std::string_view raw) { return CompressMultiGpu<K>(raw); }
  static std::string Decompress(std::string_view compressed) {
    return DecompressMultiGpu<K>(compressed);
  }
  static std::string name() { return "HuffmanGpu<" + std::to_string(K) + ">"; }
};

}  // namespace huffman
