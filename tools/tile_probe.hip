// tile_probe.hip -- single-wave latency of mm_tile_map (dev tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imonkey-moore_amd/csrc tools/tile_probe.hip \
//         monkey-moore_amd/csrc/mm_plan.cpp -o tools/tile_probe.bin
#include "../monkey-moore_amd/csrc/mm_kernels.hip"
#include <cstdio>
#include <vector>

extern "C" void mmh_set_error(const char *, ...) {}

__global__ __launch_bounds__(256) void probe(MmTileArgs a, int ntiles, long long *cycles, uint64_t *maps)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLds Wv[4];
   mm_plan_to_lds(P, a.plan);
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   const int D = a.plan.L - 1;
   for (int t = wave; t < ntiles; t += 4) {
      long long t0 = wall_clock64();
      long long c0 = clock64();
      MmPhaseMap<4> M;
      int64_t lo = (int64_t)t * MM_TILE;
      mm_tile_map<4>(a, P, Wv[wave], 0, lo, MM_TILE, (uint32_t)(lo % D), lane, M);
      long long c1 = clock64();
      long long t1 = wall_clock64();
      if (lane == 0) {
         cycles[2 * (blockIdx.x * ntiles + t)] = c1 - c0;
         cycles[2 * (blockIdx.x * ntiles + t) + 1] = t1 - t0;
         maps[blockIdx.x * ntiles + t] = M.w[0];
      }
   }
}

int main(int argc, char **argv)
{
   int blocks = argc > 1 ? atoi(argv[1]) : 1;
   int ntiles = 16;
   size_t n = (size_t)64 << 20;
   uint8_t *rom;
   hipMalloc(&rom, n);
   mm::launch_synth(0, rom, n, 42, 0);
   mmh_plan_desc pl;
   uint32_t kw[12]; const char *k = "relativesrch";
   for (int i = 0; i < 12; i++) kw[i] = k[i];
   mmh_plan_relative(1, kw, 12, 0, nullptr, 0, &pl);
   MmGeom g; g.rom = rom; g.nbytes = n; g.block_bytes = 0; g.nblocks = 1; g.S = 1; g.L = 12; g.big_endian = 0; g.whole = 1;
   MmTileArgs a = mm::tile_args(g, pl);
   long long *cyc; uint64_t *maps;
   hipMalloc(&cyc, sizeof(long long) * 2 * ntiles * blocks); hipMalloc(&maps, 8 * ntiles * blocks);
   for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, a, ntiles, cyc, maps);
      hipDeviceSynchronize();
   }
   std::vector<long long> h(2 * ntiles * blocks);
   hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
   for (int t = 0; t < ntiles; t++) printf("tile %2d: %8lld shader cycles  %6.2f us (100MHz wall ticks %lld)\n", t, h[2 * t], h[2 * t + 1] / 100.0, h[2 * t + 1]);
   return 0;
}
