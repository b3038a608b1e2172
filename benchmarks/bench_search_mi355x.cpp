// SPDX-License-Identifier: GPL-3.0-or-later
// bench_search_mi355x.cpp -- the reference's benchmark cases (benchmarks/bench_search.cpp:
// BM_Search/Relative{,/Wildcard/{Front,Middle,Back}}/{8,16}-Bit, 128 KiB .. 16 MiB x4,
// mt19937(42) data, bytes per second) run through the SAME public API -- MonkeyMoore<T>
// constructed once, search(data, size) timed -- against the MI355X facade.
//
// google-benchmark is not in the image, so this is a small stand-alone timer with the same
// case names.  Every search() call includes the host->HBM upload of the borrowed buffer:
// these are the PCIe-inclusive numbers of the drop-in path, not the HBM-resident metric of
// bench.py.
//
//   g++ -std=c++17 -O2 -Iinclude benchmarks/bench_search_mi355x.cpp -Lmonkey-moore_amd/lib \
//       -lmonkey-core -lmmoore_hip -Wl,-rpath,$PWD/monkey-moore_amd/lib -o bench_search_mi355x
#include <chrono>
#include <cstdio>
#include <limits>
#include <random>
#include <string>
#include <vector>

#include "mmoore/monkey_moore.hpp"

template <typename DataType>
static std::vector<DataType> generate_data(size_t size_in_bytes)
{
   std::vector<DataType> data(size_in_bytes / sizeof(DataType));
   std::mt19937 rng(42);
   std::uniform_int_distribution<unsigned int> dist(0, std::numeric_limits<DataType>::max());
   for (auto &v : data) {
      v = static_cast<DataType>(dist(rng));
   }
   return data;
}

template <typename DataType>
static void run_case(const std::string &name, const std::vector<CharType> &keyword, CharType wildcard)
{
   // RangeMultiplier(4)->Range(128<<10, 16<<20) of bench_search.cpp:69-70: 128 KiB, 512 KiB, 2, 8 and 16 MiB
   for (size_t bytes : {size_t(128) << 10, size_t(512) << 10, size_t(2) << 20, size_t(8) << 20, size_t(16) << 20}) {
      auto data = generate_data<DataType>(bytes);
      MonkeyMoore<DataType> searcher(keyword, wildcard, {});
      size_t hits = 0;
      for (int i = 0; i < 3; i++) {
         hits = searcher.search(data.data(), data.size()).size();       // warm-up (context, clocks)
      }
      int iterations = 0;
      const auto t0 = std::chrono::steady_clock::now();
      double elapsed = 0;
      while (elapsed < 0.25 || iterations < 5) {
         hits = searcher.search(data.data(), data.size()).size();
         iterations++;
         elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      }
      const double per_iter = elapsed / iterations;
      std::printf("%-44s/%-9zu %10.1f us  %8.3f GB/s  iterations %-6d hits %zu\n", name.c_str(), bytes, per_iter * 1e6,
                  bytes / per_iter / 1e9, iterations, hits);
   }
}

int main()
{
   const std::vector<CharType> abcde = {'a', 'b', 'c', 'd', 'e'};
   run_case<uint8_t>("BM_Search/Relative/8-Bit", abcde, 0);
   run_case<uint16_t>("BM_Search/Relative/16-Bit", abcde, 0);
   run_case<uint8_t>("BM_Search/Relative/Wildcard/Front/8-Bit", {'*', 'b', 'c', 'd', 'e'}, '*');
   run_case<uint8_t>("BM_Search/Relative/Wildcard/Middle/8-Bit", {'a', 'b', '*', 'd', 'e'}, '*');
   run_case<uint8_t>("BM_Search/Relative/Wildcard/Back/8-Bit", {'a', 'b', 'c', 'd', '*'}, '*');
   run_case<uint16_t>("BM_Search/Relative/Wildcard/Front/16-Bit", {'*', 'b', 'c', 'd', 'e'}, '*');
   run_case<uint16_t>("BM_Search/Relative/Wildcard/Middle/16-Bit", {'a', 'b', '*', 'd', 'e'}, '*');
   run_case<uint16_t>("BM_Search/Relative/Wildcard/Back/16-Bit", {'a', 'b', 'c', 'd', '*'}, '*');
   return 0;
}
