// filter_probe.hip -- timing of the streaming filter and experimental variants (dev tool).
#include "../monkey-moore_amd/csrc/mm_kernels.hip"
#include <cstdio>
#include <vector>
#include <functional>
#include <chrono>

extern "C" void mmh_set_error(const char *, ...) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__device__ __forceinline__ uint4 ld(const uint4 *p)
{
   if (MODE == 3) {
      u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
      return make_uint4(v.x, v.y, v.z, v.w);
   }
   return *p;
}
// variant: MODE 0 = full compute, 1 = loads + trivial xor (no SWAR), 2 = SWAR but no dpp/readlane carry
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void filt_var(MmFilterArgs a, uint32_t *sink)
{
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = __builtin_amdgcn_readfirstlane((uint32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
   const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
   const uint64_t gps = a.groups_per_span;
   const uint4 *rom4 = reinterpret_cast<const uint4 *>(a.g.rom);
   uint32_t acc = 0;
   for (uint64_t g0 = wave * gps; g0 < a.ngroups; g0 += nwaves * gps) {
      const uint64_t g1 = g0 + gps < a.ngroups ? g0 + gps : a.ngroups;
      uint32_t carry = 0;
      uint4 w[DEPTH + 1][4];
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
         const uint64_t gg = g0 + d < g1 ? g0 + d : g1 - 1;
         const uint4 *p = rom4 + gg * 256 + lane;
         w[d][0] = ld<MODE>(p); w[d][1] = ld<MODE>(p + 64); w[d][2] = ld<MODE>(p + 128); w[d][3] = ld<MODE>(p + 192);
      }
      for (uint64_t g = g0; g < g1; g += DEPTH + 1) {
#pragma unroll
         for (int s = 0; s <= DEPTH; s++) {
            // slot s holds group g+s; refill the slot DEPTH ahead
            const int slot_new = (s + DEPTH) % (DEPTH + 1);
            const uint64_t gn = g + s + DEPTH < g1 ? g + s + DEPTH : g1 - 1;
            const uint4 *pn = rom4 + gn * 256 + lane;
            w[slot_new][0] = ld<MODE>(pn); w[slot_new][1] = ld<MODE>(pn + 64); w[slot_new][2] = ld<MODE>(pn + 128); w[slot_new][3] = ld<MODE>(pn + 192);
            if (g + s < g1) {
               if (MODE == 1) {
#pragma unroll
                  for (int u = 0; u < 4; u++) acc ^= w[s][u].x ^ w[s][u].y ^ w[s][u].z ^ w[s][u].w;
               }
               else {
                  uint32_t h[4][4];
                  uint32_t any = 0;
                  constexpr int CM = MODE == 3 ? 0 : MODE;
#pragma unroll
                  for (int u = 0; u < 4; u++) {
                     uint32_t c = CM == 2 ? 0u : (u == 0 ? carry : __builtin_amdgcn_readlane(w[s][u - 1].w, 63));
                     if (CM == 2) {
                        uint32_t back = w[s][u].w << 8;
                        uint32_t dbprev = mm_bytesub(back, back << 8);
                        h[u][0] = mm_f8_hits<2>(w[s][u].x, back, dbprev, a.pat);
                        h[u][1] = mm_f8_hits<2>(w[s][u].y, w[s][u].x, dbprev, a.pat);
                        h[u][2] = mm_f8_hits<2>(w[s][u].z, w[s][u].y, dbprev, a.pat);
                        h[u][3] = mm_f8_hits<2>(w[s][u].w, w[s][u].z, dbprev, a.pat);
                        any |= h[u][0] | h[u][1] | h[u][2] | h[u][3];
                     }
                     else {
                        uint32_t back = __builtin_amdgcn_update_dpp(c, w[s][u].w, 0x138, 0xf, 0xf, false);
                        any |= mm_f8_chunk<4>(w[s][u], back, a.pat, h[u]);
                     }
                  }
                  carry = __builtin_amdgcn_readlane(w[s][3].w, 63);
                  if (__ballot(any != 0) != 0) {
                     acc += any;
                  }
               }
            }
         }
      }
   }
   if (acc == 0x12345678) sink[0] = acc;
}

static void timeit(const char *name, uint64_t bytes, std::function<void()> launch)
{
   hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
   for (int i = 0; i < 3; i++) launch();
   CK(hipDeviceSynchronize());
   float best = 1e9, sum = 0; int n = 10;
   for (int i = 0; i < n; i++) {
      CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; sum += ms;
   }
   printf("%-52s avg %.3f ms %5.0f GB/s   best %.3f ms %5.0f GB/s\n", name, sum / n, bytes / (sum / n) / 1e6, best, bytes / best / 1e6);
}

int main(int argc, char **argv)
{
   uint64_t n = 4ull << 30;
   uint8_t *rom; CK(hipMalloc(&rom, n + 64));
   mm::launch_synth(0, rom, n, 42, 0);
   uint32_t *sink; CK(hipMalloc(&sink, 64));
   uint64_t *cand; unsigned long long *cnt; CK(hipMalloc(&cand, 8 << 20)); CK(hipMalloc(&cnt, 65536));
   mmh_plan_desc pl; uint32_t kw[12]; const char *k = "relativesrch";
   for (int i = 0; i < 12; i++) kw[i] = k[i];
   mmh_plan_relative(1, kw, 12, 0, nullptr, 0, &pl);
   MmGeom g; g.rom = rom; g.nbytes = n; g.block_bytes = 524288; g.nblocks = n / 524288; g.S = 1; g.L = 12; g.big_endian = 0; g.whole = 0;
   mm::FilterChoice fc; mm::choose_filter(pl, &fc);
   MmFilterArgs a; a.g = g; a.plan = pl; for (int q = 0; q < 4; q++) a.pat[q] = fc.pat[q]; a.iA = fc.iA; a.ncond = fc.ncond;
   a.cand = cand; a.list_count = cnt; a.list_cap = (1 << 20) / MM_CAND_LISTS; a.verify = 0; a.ngroups = n / 4096; a.edge_first = a.ngroups * 256;
   CK(hipDeviceSynchronize());
   char nm[128];
   for (int gps : {4, 16, 64}) for (int grid : {1024, 2048, 4096}) {
      a.groups_per_span = gps;
      snprintf(nm, sizeof nm, "product mm_filter_u8<4> gps %d grid %d", gps, grid);
      timeit(nm, n, [&] { hipMemsetAsync(cnt, 0, 8, 0); hipLaunchKernelGGL(mm_filter_u8<4>, dim3(grid), dim3(256), 0, 0, a); });
   }
   {
      hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float sum = 0, best = 1e9;
      for (int i = 0; i < 14; i++) {
         CK(hipMemsetAsync(cnt, 0, 8, st));
         CK(hipEventRecord(e0, st)); hipLaunchKernelGGL(mm_filter_u8<4>, dim3(2048), dim3(256), 0, st, a); CK(hipEventRecord(e1, st));
         CK(hipStreamSynchronize(st));
         float ms; CK(hipEventElapsedTime(&ms, e0, e1));
         if (i >= 4) { sum += ms; best = ms < best ? ms : best; }
      }
      printf("launch pattern: %-28s avg %.3f ms  best %.3f ms\n", "non-blocking stream", sum / 10, best);
      // cold device allocation like Engine.alloc (padding + memset of the tail)
      uint8_t *rom2; CK(hipMalloc(&rom2, n + 32));
      mm::launch_synth(st, rom2, n, 42, 0);
      MmFilterArgs a2 = a; a2.g.rom = rom2;
      sum = 0; best = 1e9;
      for (int i = 0; i < 14; i++) {
         CK(hipMemsetAsync(cnt, 0, 8, st));
         CK(hipEventRecord(e0, st)); hipLaunchKernelGGL(mm_filter_u8<4>, dim3(2048), dim3(256), 0, st, a2); CK(hipEventRecord(e1, st));
         CK(hipStreamSynchronize(st));
         float ms; CK(hipEventElapsedTime(&ms, e0, e1));
         if (i >= 4) { sum += ms; best = ms < best ? ms : best; }
      }
      printf("launch pattern: %-28s avg %.3f ms  best %.3f ms\n", "second 4 GiB buffer", sum / 10, best);
   }
   // how the launch pattern changes the kernel time (same kernel, same data)
   a.groups_per_span = 16;
   {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int mode = 0; mode < 4; mode++) {
         float sum = 0, best = 1e9;
         for (int i = 0; i < 14; i++) {
            if (mode >= 1) CK(hipDeviceSynchronize());
            if (mode == 2) { auto t0 = std::chrono::steady_clock::now(); while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 200e-6) {} }
            if (mode == 3) { auto t0 = std::chrono::steady_clock::now(); while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 5e-3) {} }
            CK(hipMemsetAsync(cnt, 0, 8, 0));
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(mm_filter_u8<4>, dim3(2048), dim3(256), 0, 0, a); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (i >= 4) { sum += ms; best = ms < best ? ms : best; }
         }
         const char *names[] = {"event-sync only", "device sync before each", "sync + 200 us host gap", "sync + 5 ms host gap"};
         printf("launch pattern: %-28s avg %.3f ms  best %.3f ms\n", names[mode], sum / 10, best);
      }
   }
   a.groups_per_span = 16;
   for (int grid : {2048, 4096}) {
      snprintf(nm, sizeof nm, "variant full depth1 grid %d", grid);
      timeit(nm, n, [&] { hipLaunchKernelGGL((filt_var<0, 1>), dim3(grid), dim3(256), 0, 0, a, sink); });
      snprintf(nm, sizeof nm, "variant full depth2 grid %d", grid);
      timeit(nm, n, [&] { hipLaunchKernelGGL((filt_var<0, 2>), dim3(grid), dim3(256), 0, 0, a, sink); });
      snprintf(nm, sizeof nm, "variant xor-only depth1 grid %d", grid);
      timeit(nm, n, [&] { hipLaunchKernelGGL((filt_var<1, 1>), dim3(grid), dim3(256), 0, 0, a, sink); });
      snprintf(nm, sizeof nm, "variant xor-only depth2 grid %d", grid);
      timeit(nm, n, [&] { hipLaunchKernelGGL((filt_var<1, 2>), dim3(grid), dim3(256), 0, 0, a, sink); });
      snprintf(nm, sizeof nm, "variant swar-no-carry depth1 grid %d", grid);
      timeit(nm, n, [&] { hipLaunchKernelGGL((filt_var<2, 1>), dim3(grid), dim3(256), 0, 0, a, sink); });
   }
   a.groups_per_span = 16;
   for (int grid : {2048}) {
      snprintf(nm, sizeof nm, "variant full nt-loads depth2 grid %d", grid);
      timeit(nm, n, [&] { hipLaunchKernelGGL((filt_var<3, 2>), dim3(grid), dim3(256), 0, 0, a, sink); });
      snprintf(nm, sizeof nm, "variant full nt-loads depth3 grid %d", grid);
      timeit(nm, n, [&] { hipLaunchKernelGGL((filt_var<3, 3>), dim3(grid), dim3(256), 0, 0, a, sink); });
      snprintf(nm, sizeof nm, "variant full depth3 grid %d", grid);
      timeit(nm, n, [&] { hipLaunchKernelGGL((filt_var<0, 3>), dim3(grid), dim3(256), 0, 0, a, sink); });
   }
   a.groups_per_span = 48;
   for (int grid : {2048}) {
      snprintf(nm, sizeof nm, "variant full depth2 gps48 grid %d", grid);
      timeit(nm, n, [&] { hipLaunchKernelGGL((filt_var<0, 2>), dim3(grid), dim3(256), 0, 0, a, sink); });
   }
   return 0;
}
