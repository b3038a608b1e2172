// SPDX-License-Identifier: GPL-3.0-or-later
// The plan builder (csrc/mm_plan.cpp: the restatement of MonkeyMoore's constructors and
// preprocess* as a POD plan) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU --
// GPU sanitizers are not available on the pool, and this is the part of the library that turns
// caller-controlled keywords into table indices.  Random keywords of every mode and length
// 1 .. 140 (beyond MMH_MAX_KEYWORD on purpose), wildcards anywhere, mixed case, custom
// sequences with symbols missing from them, value scans with extreme values.  The checks here
// are structural; the tables' CONTENTS are pinned against the reference in test_plan_and_model.py.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "mmoore_hip.h"

static std::string g_error;
extern "C" void mmh_set_error(const char *fmt, ...)
{
   char buf[512];
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(buf, sizeof buf, fmt, ap);
   va_end(ap);
   g_error = buf;
}
extern "C" const char *mmh_last_error(void) { return g_error.c_str(); }

static int fail(const char *what, int line)
{
   std::fprintf(stderr, "plan_sanitize: %s (line %d): %s\n", what, line, g_error.c_str());
   return 1;
}
#define CHECK(x) do { if (!(x)) return fail(#x, __LINE__); } while (0)

int main()
{
   std::mt19937_64 rng(20261003);
   uint64_t built = 0, refused = 0;
   for (int trial = 0; trial < 60000; trial++) {
      const uint32_t elem = (rng() & 1) ? 1 : 2;
      const uint32_t kind = (uint32_t)(rng() % 5);       // 0 simple, 1 wildcard, 2 mixed case, 3 custom sequence, 4 value scan
      uint32_t L = (uint32_t)(rng() % 24);
      if (rng() % 8 == 0) {
         L = (uint32_t)(rng() % 141);                     // long ones, also beyond the limit
      }
      mmh_plan_desc plan;
      std::memset(&plan, 0xA5, sizeof plan);
      int rc;
      if (kind == 4) {
         std::vector<int16_t> v(L);
         for (auto &x : v) {
            x = (rng() % 16 == 0) ? (int16_t)((rng() & 1) ? 32767 : -32768) : (int16_t)(rng() % 512 - 256);
         }
         rc = mmh_plan_value_scan(elem, v.data(), L, &plan);
      }
      else {
         std::vector<uint32_t> kw(L), seq;
         const uint32_t alphabet = 2 + (uint32_t)(rng() % 24);
         for (auto &c : kw) {
            c = 'a' + (uint32_t)(rng() % alphabet);
            if (kind == 2 && (rng() & 1)) {
               c = c - 'a' + 'A';
            }
            if (rng() % 64 == 0) {
               c = (uint32_t)(rng() % 0x11000);           // anything, also beyond the element type
            }
         }
         uint32_t wildcard = 0;
         if (kind == 1 || rng() % 6 == 0) {
            wildcard = '*';
            for (auto &c : kw) {
               if (rng() % 3 == 0) {
                  c = '*';
               }
            }
         }
         if (kind == 3) {
            const uint32_t n = (uint32_t)(rng() % 40);
            for (uint32_t i = 0; i < n; i++) {
               seq.push_back('a' + (uint32_t)(rng() % 30));   // duplicates and gaps on purpose
            }
         }
         rc = mmh_plan_relative(elem, kw.data(), L, wildcard, seq.empty() ? nullptr : seq.data(), (uint32_t)seq.size(), &plan);
      }
      if (rc != MMH_OK) {
         refused++;
         CHECK(!g_error.empty());
         continue;
      }
      built++;
      CHECK(plan.L >= 2 && plan.L <= MMH_MAX_KEYWORD);
      CHECK(plan.elem_bytes == elem);
      CHECK(plan.match_jump >= 1 && plan.match_jump < plan.L);
      CHECK(plan.n_skip <= MMH_MAX_KEYWORD);
   }
   // the argument checks
   mmh_plan_desc plan;
   uint32_t kw[3] = {'a', 'b', 'c'};
   CHECK(mmh_plan_relative(3, kw, 3, 0, nullptr, 0, &plan) != MMH_OK);
   CHECK(mmh_plan_relative(1, nullptr, 3, 0, nullptr, 0, &plan) != MMH_OK);
   CHECK(mmh_plan_relative(1, kw, 3, 0, nullptr, 0, nullptr) != MMH_OK);
   CHECK(mmh_plan_relative(1, kw, 0, 0, nullptr, 0, &plan) != MMH_OK);
   CHECK(mmh_plan_value_scan(1, nullptr, 3, &plan) != MMH_OK);
   std::printf("plan_sanitize: %llu plans built, %llu keywords refused, no sanitizer report\n", (unsigned long long)built,
               (unsigned long long)refused);
   return built > 10000 && refused > 1000 ? 0 : fail("too few cases of one kind", __LINE__);
}
