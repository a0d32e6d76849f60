// The reference's typed compressor test suite (codec/huffman_test.cpp:47-184)
// re-created without gtest, instantiated over huffman::HuffmanCompressorGpu<K>.
// Every assertion is round-trip equality, as in the reference.  Inputs are the
// committed fixtures (argv[1] = tests/golden/inputs).  Exit status 0 = all pass.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/huffman_gpu.h"

static std::string slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f.good()) {
    fprintf(stderr, "cannot open %s\n", path.c_str());
    exit(2);
  }
  return std::string(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

static int g_failed = 0, g_run = 0;

template <typename T>
static void round_trip(const char* test, const std::string& raw) {
  ++g_run;
  std::string compressed = T::Compress(raw);
  std::string decompressed = T::Decompress(compressed);
  if (decompressed != raw) {
    ++g_failed;
    printf("[  FAILED  ] %s/%s (%zu bytes)\n", T::name().c_str(), test, raw.size());
  }
}

template <typename T>
static void run_suite(const std::string& dir) {
  round_trip<T>("Hello", "Hello World");
  round_trip<T>("LongerText", slurp(dir + "/lorem.bin"));
  round_trip<T>("EqualCounts", slurp(dir + "/equal_counts.bin"));
  round_trip<T>("LongRandom", slurp(dir + "/long_random.bin"));
  round_trip<T>("SingleSymbolOnly", "AAA");
  round_trip<T>("SingleSymbolOnly", std::string(1000, 'a'));
  round_trip<T>("LongCodes", slurp(dir + "/long_codes.bin"));
  round_trip<T>("EmptyString", "");
  const std::string pack = slurp(dir + "/many_random.pack");
  size_t pos = 0;
  while (pos + 4 <= pack.size()) {
    uint32_t n;
    memcpy(&n, pack.data() + pos, 4);
    pos += 4;
    round_trip<T>("ManyRandom", pack.substr(pos, n));
    pos += n;
  }
  printf("[       OK ] %s\n", T::name().c_str());
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "tests/golden/inputs";
  if (hfz_device_count() <= 0) {
    printf("no GPU\n");
    return 3;
  }
  run_suite<huffman::HuffmanCompressorGpu<1>>(dir);
  run_suite<huffman::HuffmanCompressorGpu<4>>(dir);
  run_suite<huffman::HuffmanCompressorGpu<8>>(dir);
  run_suite<huffman::HuffmanCompressorGpu<32>>(dir);
  run_suite<huffman::HuffmanCompressorGpu<64>>(dir);
  // codec/huffman_test.cpp:186-197 (MultiTest.Compress2 prints the K=2 frame)
  const std::string c2 = huffman::CompressMultiGpu<2>("Hello World");
  for (unsigned char ch : c2) printf("%02x ", ch);
  printf("\n%d round trips, %d failed\n", g_run, g_failed);
  return g_failed ? 1 : 0;
}
