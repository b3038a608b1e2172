// SPDX-License-Identifier: GPL-3.0-or-later
// bench_engine_file.cpp -- end-to-end rate of SearchEngine<T>::run on a FILE (the reference's
// GUI / test entry point, src/core/search_engine.cpp:23-216): file in the page cache (tmpfs)
// -> parallel readers -> pinned staging -> PCIe -> HBM -> scan -> equivalency maps.  This is
// the PCIe-inclusive number of the drop-in path; bench.py's metric is the HBM-resident scan.
//
//   bench_engine_file [size_MiB = 4096] [dir = /dev/shm] [repeats = 3]
//
// Linked against libmonkey-core.so of this repository; the same source builds against the
// reference's monkey-core for a CPU figure on the same file.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <unistd.h>

#include "mmoore/search_engine.hpp"

static uint64_t splitmix(uint64_t &x)
{
   uint64_t z = (x += 0x9E3779B97F4A7C15ull);
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
   return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
   const uint64_t mib = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 4096;
   const std::string dir = argc > 2 ? argv[2] : "/dev/shm";
   const int repeats = argc > 3 ? std::atoi(argv[3]) : 3;
   const uint64_t size = mib << 20;
   const std::string path = dir + "/mmoore_bench_" + std::to_string(getpid()) + ".rom";
   const std::string keyword = "relativesrch";

   {
      // random bytes with the keyword's shape planted once per MiB (shifted alphabets)
      std::ofstream out(path, std::ios::binary);
      std::vector<uint64_t> piece((1u << 20) / 8);
      uint64_t state = 42;
      for (uint64_t m = 0; m < mib; m++) {
         for (auto &w : piece) {
            w = splitmix(state);
         }
         uint8_t *bytes = reinterpret_cast<uint8_t *>(piece.data());
         const size_t at = 1000 + (m * 7919) % 900000;
         const int shift = static_cast<int>(m % 100) - 50;
         for (size_t k = 0; k < keyword.size(); k++) {
            bytes[at + k] = static_cast<uint8_t>(keyword[k] + shift);
         }
         out.write(reinterpret_cast<const char *>(bytes), 1u << 20);
      }
   }

   mmoore::SearchConfig cfg;
   cfg.file_path = path;
   cfg.is_relative_search = true;
   cfg.keyword.assign(keyword.begin(), keyword.end());
   cfg.wildcard = '*';
   cfg.endianness = mmoore::Endianness::Little;
   cfg.preferred_num_threads = 0;
   cfg.preferred_search_block_size = 524288;
   cfg.preferred_preview_width = 50;
   if (const char *t = std::getenv("BENCH_THREADS")) {
      cfg.preferred_num_threads = std::atoi(t);
   }
   if (cfg.preferred_num_threads <= 0) {
      cfg.preferred_num_threads = static_cast<int>(sysconf(_SC_NPROCESSORS_ONLN));
   }

   int rc = 0;
   try {
      double best = 1e30;
      size_t hits = 0;
      for (int r = 0; r < repeats + 1; r++) {
         mmoore::SearchEngine<uint8_t> engine(cfg);
         std::atomic<bool> abort{false};
         const auto t0 = std::chrono::steady_clock::now();
         auto results = engine.run([](int, const mmoore::SearchStep) {}, abort, false);
         const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
         hits = results.size();
         std::printf("run %d%s: %.1f ms  %.2f GB/s  %zu matches\n", r, r == 0 ? " (warm-up)" : "", s * 1e3, size / s / 1e9, hits);
         if (r > 0 && s < best) {
            best = s;
         }
      }
      std::printf("{\"bench\": \"SearchEngine<uint8_t>::run on a %llu MiB tmpfs file\", \"best_ms\": %.2f, \"GB_per_s\": %.2f, "
                  "\"matches\": %zu, \"threads\": %d}\n",
                  (unsigned long long)mib, best * 1e3, size / best / 1e9, hits, cfg.preferred_num_threads);
   }
   catch (const std::exception &e) {
      std::fprintf(stderr, "failed: %s\n", e.what());
      rc = 1;
   }
   unlink(path.c_str());
   return rc;
}
