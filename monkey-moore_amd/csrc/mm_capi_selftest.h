// SPDX-License-Identifier: GPL-3.0-or-later
// mm_capi_selftest.h -- the first-use known-answer self-test of mmh_create.
// A section of mm_capi.hip (included there once, in this place: one translation unit, the helpers keep internal
// linkage).  Round 6 cut the 2 900-line file along its seams: workspace, validation, pipeline, engines, lanes, split,
// self-test; mm_capi.hip itself keeps the context, the ROM entry points, the synchronous scan and the small queries.

namespace {

// The known answer: 4133 bytes of splitmix64 noise with the keyword's shape ("abcde": four deltas of +1) planted at
// the start, across block boundaries (1020, 2044, 3068, 4092), behind a constant run, in the last bytes, twice back to
// back, off the reference's skip chain and once modulo 256 (which the reference's signed compare does not report);
// blocks of 1024 bytes.  What the reference reports is the constant below (tests/test_oracle.py holds it against the
// oracle and the compiled reference).
constexpr uint64_t kKatBytes = 4133, kKatBlock = 1024;
// (15 plants, 12 reported: 1505 and 2050 are not on the reference's skip chain, 3000 wraps around 0xFF)
const uint64_t kKatExpected[] = {0, 16, 600, 1020, 1028, 1500, 2044, 2596, 3068, 3500, 4092, 4128};

void kat_rom(uint8_t *rom)
{
   uint64_t x = 0x6d6d6f6f72653432ull;
   for (uint64_t i = 0; i < kKatBytes; i += 8) {
      x += 0x9E3779B97F4A7C15ull;
      uint64_t z = x;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      for (uint64_t k = 0; k < 8 && i + k < kKatBytes; k++) {
         rom[i + k] = (uint8_t)(z >> (8 * k));
      }
   }
   std::memset(rom + 2500, 0x41, 96);                        // a constant run: the chain crosses it in default skips
   const struct { uint32_t at; uint8_t base; } plants[] = {
      {0, 0x30}, {16, 0x61}, {600, 0x11}, {1020, 0x10}, {1028, 0xF0}, {1500, 0x00}, {1505, 0x20}, {2044, 0x77}, {2050, 0x22},
      {2596, 0x41}, {3000, 0xFD}, {3068, 0x05}, {3500, 0x90}, {4092, 0x80}, {4128, 0x33}};
   for (const auto &p : plants) {
      for (uint32_t k = 0; k < 5; k++) {
         rom[p.at + k] = (uint8_t)(p.base + k);             // (0xFD: wraps -- matches modulo 256 only)
      }
   }
}

mmh_plan_desc kat_plan()
{
   const uint32_t kw[5] = {'a', 'b', 'c', 'd', 'e'};
   mmh_plan_desc plan;
   std::memset(&plan, 0, sizeof plan);
   (void)mmh_plan_relative(1, kw, 5, 0, nullptr, 0, &plan);
   return plan;
}

// one scan of the KAT on context t with `mask` routes off; through mmh_scan or through the submit lanes
bool kat_scan(mmh_ctx *t, const uint8_t *rom, const mmh_plan_desc &plan, uint32_t mask, int engine, bool lanes, std::string *why)
{
   constexpr uint64_t n_expected = sizeof(kKatExpected) / sizeof(kKatExpected[0]);
   uint64_t got[64] = {0}, n = 0;
   t->route_off = mask;
   t->engine = engine;
   const uint64_t fallbacks = t->health.fallbacks;
   int rc = mmh_rom_upload(t, rom, kKatBytes);
   if (rc == MMH_OK) {
      if (lanes) {
         int ticket = 0;
         rc = mmh_scan_submit(t, &plan, kKatBlock, 0, 0, &ticket);
         if (rc == MMH_OK) {
            rc = mmh_scan_collect(t, ticket, got, 64, &n);
         }
      }
      else {
         rc = mmh_scan(t, &plan, kKatBlock, 0, 0, got, 64, &n);
      }
   }
   t->engine = 0;
   char buf[256];
   if (rc != MMH_OK) {
      snprintf(buf, sizeof buf, "routes off 0x%x, engine %d%s: error %d (%s)", mask, engine, lanes ? ", lanes" : "", rc, mmh_last_error());
      *why = buf;
      return false;
   }
   if (t->health.fallbacks != fallbacks) {
      snprintf(buf, sizeof buf, "routes off 0x%x%s: the published block failed validation (reason %llu)", mask, lanes ? ", lanes" : "",
               (unsigned long long)t->health.last_reason);
      *why = buf;
      return false;
   }
   if (n != n_expected || std::memcmp(got, kKatExpected, n * sizeof(uint64_t)) != 0) {
      uint64_t first = 0;
      while (first < n && first < n_expected && got[first] == kKatExpected[first]) {
         first++;
      }
      snprintf(buf, sizeof buf, "routes off 0x%x, engine %d%s: %llu offsets instead of %llu, first difference at entry %llu (%llu)", mask, engine,
               lanes ? ", lanes" : "", (unsigned long long)n, (unsigned long long)n_expected, (unsigned long long)first,
               (unsigned long long)(first < n ? got[first] : 0));
      *why = buf;
      return false;
   }
   return true;
}

} // namespace

extern "C" int mmh_selftest_kat(uint8_t *rom, uint64_t rom_cap, uint64_t *rom_bytes, uint64_t *expected, uint64_t expected_cap,
                                uint64_t *expected_count)
{
   constexpr uint64_t n_expected = sizeof(kKatExpected) / sizeof(kKatExpected[0]);
   if (!rom || !rom_bytes || !expected_count || (!expected && expected_cap) || rom_cap < kKatBytes) {
      mmh_set_error("mmh_selftest_kat: bad argument (the ROM takes %llu bytes)", (unsigned long long)kKatBytes);
      return MMH_E_ARG;
   }
   kat_rom(rom);
   *rom_bytes = kKatBytes;
   *expected_count = n_expected;
   if (expected_cap < n_expected) {
      return MMH_E_CAPACITY;
   }
   std::memcpy(expected, kKatExpected, sizeof(kKatExpected));
   return MMH_OK;
}

extern "C" int mmh_selftest_run(int device, uint32_t *routes_off_out)
{
   if (!routes_off_out) {
      mmh_set_error("mmh_selftest_run: null argument");
      return MMH_E_ARG;
   }
   *routes_off_out = 0;
   mmh_ctx *t = nullptr;
   int rc = create_context(device, &t);
   if (rc != MMH_OK) {
      return rc;
   }
   std::vector<uint8_t> rom(kKatBytes);
   kat_rom(rom.data());
   const mmh_plan_desc plan = kat_plan();
   std::string why;
   // the plain kernels and the sequential chain kernel must know the answer: everything else is measured against them
   if (!kat_scan(t, rom.data(), plan, 15, 0, false, &why) || !kat_scan(t, rom.data(), plan, 15, 1, false, &why)) {
      mmh_destroy(t);
      mmh_set_error("self-test: %s", why.c_str());
      return MMH_E_DEVICE;
   }
   // the fast routes, all on first; then with more and more of them switched off until the answer is right
   static const bool trace = mm_trace("selftest");
   const uint32_t masks[] = {0, MMH_ROUTE_NO_ZERO_COPY, MMH_ROUTE_NO_SINGLE_LAUNCH, MMH_ROUTE_NO_ZERO_COPY | MMH_ROUTE_NO_SINGLE_LAUNCH,
                             MMH_ROUTE_NO_ZERO_COPY | MMH_ROUTE_NO_SINGLE_LAUNCH | MMH_ROUTE_NO_BUCKETS, 15};
   uint32_t settled = 15;
   for (uint32_t mask : masks) {
      std::string w1;
      if (kat_scan(t, rom.data(), plan, mask, 0, false, &w1) && kat_scan(t, rom.data(), plan, mask, 0, true, &w1)) {
         settled = mask;
         break;
      }
      fprintf(stderr, "libmmoore_hip: self-test on device %d: %s\n", device, w1.c_str());
   }
   if (trace) {
      fprintf(stderr, "libmmoore_hip: self-test on device %d: routes off 0x%x\n", device, settled);
   }
   mmh_destroy(t);
   *routes_off_out = settled;
   return MMH_OK;
}
