// scan_probe.cpp -- drives the C ABI like bench.py does and prints stage timings (dev tool).
//   g++ -O2 -Iinclude tools/scan_probe.cpp -Lmonkey-moore_amd/lib -lmmoore_hip -Wl,-rpath,$PWD/monkey-moore_amd/lib -o tools/scan_probe.bin
#include "mmoore_hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
int main(int argc, char **argv)
{
   mmh_ctx *c; if (mmh_create(0, &c)) { printf("%s\n", mmh_last_error()); return 1; }
   uint64_t n = argc > 3 ? (uint64_t)atoll(argv[3]) << 20 : 4ull << 30;      // [reps] [plant] [MiB]
   if (argc > 4) { mmh_set_timing(c, 0); }                                    // [reps] [plant] [MiB] [anything: no timing events]
   mmh_rom_alloc(c, n); mmh_rom_synth(c, 42, 0);
   mmh_plan_desc pl; uint32_t kw[12]; const char *k = "relativesrch"; for (int i = 0; i < 12; i++) kw[i] = k[i];
   mmh_plan_relative(1, kw, 12, 0, nullptr, 0, &pl);
   std::vector<uint64_t> out(1 << 16); uint64_t cnt;
   int reps = argc > 1 ? atoi(argv[1]) : 12;
   if (argc > 2 && argv[2][0] == '1') {   // plant matches: 1 per MiB
      for (uint64_t m = 0; m < (n >> 20); m++) {
         uint8_t v[12]; for (int i = 0; i < 12; i++) v[i] = (uint8_t)(k[i] - 40);
         mmh_rom_poke(c, (m << 20) + 1000 + (m * 7919) % 900000, v, 12);
      }
   }
   double wall = 0; float dev = 0; int cnt_w = 0;
   for (int i = 0; i < reps; i++) {
      auto t0 = std::chrono::steady_clock::now();
      int rc = mmh_scan(c, &pl, 524288, 0, 0, out.data(), out.size(), &cnt);
      double w = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      float t[4]; mmh_last_timings(c, t);
      if (i >= reps / 2) { wall += w; dev += t[3]; cnt_w++; }
      if (reps <= 12 || i % 20 == 0 || i == reps - 1)
         printf("scan %3d rc %d matches %llu filter %.3f resolve %.3f order %.3f total %.3f\n", i, rc, (unsigned long long)cnt, t[0], t[1], t[2], t[3]);
   }
   printf("host wall per scan %.1f us, device total %.1f us, host overhead %.1f us\n", wall / cnt_w * 1e6, dev / cnt_w * 1e3, wall / cnt_w * 1e6 - dev / cnt_w * 1e3);
   mmh_destroy(c);
   return 0;
}
