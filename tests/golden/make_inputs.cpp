// Regenerates the input fixtures under tests/golden/inputs/.
//
// Each input re-creates one case of the reference's own test and benchmark
// suites (codec/huffman_test.cpp, codec/histogram_test.cpp,
// codec/huffman_benchmark.cpp).  Several of them depend on glibc rand() and on
// libstdc++'s std::shuffle / uniform_real_distribution, which are
// implementation-defined, so the bytes are generated once with this program
// and committed; tests read the committed files and never run this.
//
//   g++ -std=c++17 -O2 -o /tmp/make_inputs tests/golden/make_inputs.cpp
//   /tmp/make_inputs tests/golden/inputs
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

static const char* const kLorem = R"(
Lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod
tempor incididunt ut labore et dolore magna aliqua. Ut enim ad minim
veniam, quis nostrud exercitation ullamco laboris nisi ut aliquip ex ea
commodo consequat. Duis aute irure dolor in reprehenderit in voluptate
velit esse cillum dolore eu fugiat nulla pariatur. Excepteur sint
occaecat cupidatat non proident, sunt in culpa qui officia deserunt
mollit anim id est laborum.
    )";

static void put(const std::string& dir, const char* name, const std::string& s) {
  std::string path = dir + "/" + name;
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) {
    perror(path.c_str());
    exit(1);
  }
  fwrite(s.data(), 1, s.size(), f);
  fclose(f);
  uint64_t h = 0xcbf29ce484222325ull;
  for (unsigned char c : s) {
    h ^= c;
    h *= 0x100000001b3ull;
  }
  printf("%-28s %8zu  fnv1a64=%016llx\n", name, s.size(), (unsigned long long)h);
}

static void put_u32(std::string& s, uint32_t x) {
  for (int i = 0; i < 4; ++i) s.push_back(char((x >> (8 * i)) & 0xff));
}

// codec/huffman_benchmark.cpp:27-36
static std::string proba(double p, size_t len) {
  std::string data(len, 0);
  std::mt19937_64 mt;
  std::uniform_real_distribution<> dist(0.0, 1.0);
  double logp = log(1 - p);
  for (char& c : data) c = int(log(dist(mt)) / logp) % 256;
  return data;
}

int main(int argc, char** argv) {
  std::string dir = argc > 1 ? argv[1] : ".";
  const size_t LEN = 100 << 10;

  put(dir, "hello.bin", "Hello World");   // huffman_test.cpp:56-71
  put(dir, "lorem.bin", kLorem);          // :73-98
  {                                       // :100-113 EqualCounts
    std::string raw;
    for (int i = 0; i < 4; ++i)
      for (int c = 0; c < 256; ++c) raw.push_back(char(c));
    put(dir, "bytes_x4.bin", raw);  // unshuffled variant (tie-order pin)
    std::shuffle(raw.begin(), raw.end(), std::mt19937());
    put(dir, "equal_counts.bin", raw);
  }
  {  // :115-131 LongRandom
    std::string raw;
    srand(0);
    for (int i = 0; i < 100000; ++i) raw.push_back(char((rand() & rand() & rand()) & 0xff));
    put(dir, "long_random.bin", raw);
  }
  put(dir, "aaa.bin", "AAA");                     // :133-142
  put(dir, "a1000.bin", std::string(1000, 'a'));  //
  {                                               // :144-156 LongCodes
    std::string text;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < (1 << i); ++j) text.push_back(char('A' + i));
    put(dir, "long_codes.bin", text);
  }
  put(dir, "empty.bin", "");  // :158-162
  {                           // :164-184 ManyRandom, packed as (u32 len, bytes)*
    std::mt19937 mt;
    std::string pack;
    for (int k = 0; k < 100; ++k) {
      std::string raw;
      int len = 1 + mt() % 1000;
      for (int i = 0; i < len; ++i) {
        uint8_t ch = 0;
        do {
          ch = (mt() & mt()) & 0xff;
          ch ^= 'A';
        } while (!std::isprint(ch));
        raw.push_back(char(ch));
      }
      put_u32(pack, uint32_t(raw.size()));
      pack += raw;
    }
    put(dir, "many_random.pack", pack);
  }
  {  // histogram_test.cpp:28-42 (rand() with its default seed)
    srand(1);
    std::string str;
    for (int i = 0; i < 2007; ++i) str.push_back(char((rand() & rand() & rand()) & 0xff));
    put(dir, "hist_long.bin", str);
  }
  put(dir, "foobar.bin", "foobar");  // histogram_test.cpp:18-26
  // huffman_benchmark.cpp generators
  {
    std::string biased = proba(0.2, LEN);  // :62-68
    put(dir, "biased_100k.bin", biased);
    std::sort(biased.begin(), biased.end());  // :84-91
    put(dir, "sorted_100k.bin", biased);
  }
  {  // :108-122 Uniform
    srand(0);
    std::string raw;
    for (size_t i = 0; i < LEN; ++i) raw.push_back(char(uint8_t(rand())));
    put(dir, "uniform_100k.bin", raw);
  }
  {  // :142-156 Short
    srand(0);
    std::string raw;
    for (int i = 0; i < 100; ++i) raw.push_back(char(uint8_t(rand() & rand() & rand())));
    put(dir, "short_100.bin", raw);
  }
  {  // :187-199 Lorem
    std::string raw;
    while (raw.size() < LEN) raw.append(kLorem);
    raw.resize(LEN);
    put(dir, "lorem_100k.bin", raw);
  }
  return 0;
}
