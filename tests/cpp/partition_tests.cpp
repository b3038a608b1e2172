// SPDX-License-Identifier: GPL-3.0-or-later
// partition_tests.cpp -- the multi-GPU partition rule of the C ABI (mmh_partition), host only.
// Replaces what the reference's compute_search_blocks + dispatcher guarantee together
// (src/core/search_engine.cpp:66-188, :218-253): every block belongs to exactly one worker and
// every worker sees the (L-1)*S bytes behind its last block.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "mmoore_hip.h"

static int failures = 0, checks = 0;
#define CHECK(cond, ...) do { checks++; if (!(cond)) { failures++; std::printf("FAIL %s:%d: %s -- ", __FILE__, __LINE__, #cond); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

int main()
{
   const uint64_t totals[] = {0, 1, 11, 12, 4096, 4097, 524288, (64ull << 20) + 12345, 8192ull * 524288, (8ull << 30) * 8 + 5, (1ull << 44) + 3};
   const uint64_t blocks[] = {5, 4096, 524288, 8388608};
   for (uint64_t total : totals) {
      for (uint64_t block : blocks) {
         if (total / block > (1ull << 24)) {
            continue;                                   // keep the per-block loop below short
         }
         for (uint32_t S = 1; S <= 2; S++) {
            for (uint32_t L : {2u, 8u, 12u, 32u}) {
               for (int world : {1, 2, 3, 8, 64}) {
                  const uint64_t overlap = (uint64_t)(L - 1) * S;
                  const uint64_t nblocks = (total + block - 1) / block;
                  std::vector<uint64_t> first(world + 1), bytes(world);
                  for (int r = 0; r < world; r++) {
                     CHECK(mmh_partition(total, block, L, S, r, world, &first[r], &bytes[r]) == MMH_OK, "rc");
                  }
                  first[world] = total;
                  uint64_t blocks_seen = 0;
                  for (int r = 0; r < world; r++) {
                     CHECK(first[r] % block == 0, "partition %d of %d starts inside a block", r, world);
                     CHECK(r == 0 ? first[r] == 0 : first[r] >= first[r - 1], "partitions out of order");
                     const uint64_t next = r + 1 < world ? first[r + 1] : nblocks * block;
                     const uint64_t own_blocks = (next - first[r]) / block;
                     blocks_seen += own_blocks;
                     // balanced to within one block
                     CHECK(own_blocks + 1 >= nblocks / world && own_blocks <= nblocks / world + 1, "unbalanced: %llu of %llu blocks",
                           (unsigned long long)own_blocks, (unsigned long long)nblocks);
                     // the partition holds its blocks plus the overlap, clipped to the file
                     const uint64_t want_end = own_blocks ? (next + overlap < total ? next + overlap : total) : first[r];
                     CHECK(first[r] + bytes[r] == want_end, "partition %d: bytes %llu", r, (unsigned long long)bytes[r]);
                     // attribution: a match starting at any offset o in [first, next) with all L elements in the
                     // file lies inside [first, first + bytes)
                     if (own_blocks) {
                        const uint64_t last_start = next < total ? next - 1 : (total >= (uint64_t)L * S ? total - (uint64_t)L * S : 0);
                        if (last_start >= first[r] && last_start + (uint64_t)L * S <= total) {
                           CHECK(last_start + (uint64_t)L * S <= first[r] + bytes[r] || S == 2,
                                 "a match starting in partition %d sticks out of it", r);
                        }
                     }
                  }
                  CHECK(blocks_seen == nblocks, "blocks dealt out: %llu of %llu", (unsigned long long)blocks_seen, (unsigned long long)nblocks);
               }
            }
         }
      }
   }
   uint64_t a = 0, b = 0;
   CHECK(mmh_partition(100, 0, 12, 1, 0, 1, &a, &b) == MMH_E_ARG, "block size 0 must be rejected");
   CHECK(mmh_partition(100, 16, 12, 1, 2, 2, &a, &b) == MMH_E_ARG, "rank out of range must be rejected");
   CHECK(mmh_partition(100, 16, 12, 3, 0, 1, &a, &b) == MMH_E_ARG, "element size 3 must be rejected");
   CHECK(mmh_partition(100, 16, 12, 1, 0, 1, nullptr, &b) == MMH_E_ARG, "null output must be rejected");
   std::printf("%d checks, %d failures\n", checks, failures);
   return failures ? 1 : 0;
}
