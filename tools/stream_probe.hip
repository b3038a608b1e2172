// stream_probe.hip -- read-bandwidth ceilings on the box for the access patterns
// the filter kernel can use.  Dev tool (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// pattern A: grid-stride, one 16B chunk per lane per load, UNROLL loads stride apart
template <int UNROLL>
__global__ __launch_bounds__(256) void read_gridstride(const uint4 *p, uint64_t nchunks, uint32_t *sink)
{
   uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
   uint32_t acc = 0;
   for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c + stride * (UNROLL - 1) < nchunks; c += stride * UNROLL) {
      uint4 w[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) w[u] = p[c + stride * u];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) acc ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w;
   }
   if (acc == 0x12345678) sink[0] = acc;
}

// pattern B: each wave streams a contiguous span, UNROLL consecutive 1 KiB pieces per iteration
template <int UNROLL>
__global__ __launch_bounds__(256) void read_wavespan(const uint4 *p, uint64_t nchunks, uint64_t span_chunks, uint32_t *sink)
{
   uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   uint32_t lane = threadIdx.x & 63;
   uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
   uint32_t acc = 0;
   for (uint64_t s = wave * span_chunks; s < nchunks; s += nwaves * span_chunks) {
      uint64_t e = s + span_chunks < nchunks ? s + span_chunks : nchunks;
      for (uint64_t c = s + lane; c + 64 * (UNROLL - 1) < e; c += 64 * UNROLL) {
         uint4 w[UNROLL];
#pragma unroll
         for (int u = 0; u < UNROLL; u++) w[u] = p[c + 64 * u];
#pragma unroll
         for (int u = 0; u < UNROLL; u++) acc ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w;
      }
   }
   if (acc == 0x12345678) sink[0] = acc;
}

// pattern A with the filter's amount of VALU work per dword (14 dependent-ish ops), software pipelined:
// the load of iteration i+1 is issued before iteration i is processed (DEPTH loads in flight per lane)
__device__ __forceinline__ uint32_t busy14(uint32_t w, uint32_t prev, uint32_t k)
{
   uint32_t d = (w | 0x80808080u) - (__builtin_amdgcn_alignbit(w, prev, 24) & 0x7F7F7F7Fu);
   d ^= ~(w ^ prev) & 0x80808080u;
   uint32_t z = d ^ k;
   z |= __builtin_amdgcn_alignbit(d, prev, 24) ^ (k * 3u);
   return (z - 0x01010101u) & ~z;
}
template <int DEPTH>
__global__ __launch_bounds__(256) void read_gridstride_work(const uint4 *p, uint64_t nchunks, uint32_t *sink, uint32_t k)
{
   const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
   uint32_t acc = 0;
   uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
   uint4 w[DEPTH];
#pragma unroll
   for (int u = 0; u < DEPTH; u++) w[u] = p[(c + stride * u) < nchunks ? c + stride * u : c];
   for (; c < nchunks; c += stride * DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; u++) {
         const uint4 cur = w[u];
         const uint64_t nx = c + stride * (u + DEPTH);
         w[u] = p[nx < nchunks ? nx : c];
         const uint32_t back = __builtin_amdgcn_update_dpp(k, cur.w, 0x138, 0xf, 0xf, false);
         acc |= busy14(cur.x, back, k) | busy14(cur.y, cur.x, k) | busy14(cur.z, cur.y, k) | busy14(cur.w, cur.z, k);
      }
   }
   if (acc == k * 7u) sink[0] = acc;
}

// pattern C ("sweep", round 5): the whole grid sweeps the ROM front to back -- in round k wave v takes the LOADS adjacent
// 1 KiB pieces at piece (k nwaves + v) LOADS, so that what is in flight at any time is DEPTH contiguous windows of
// nwaves x LOADS KiB instead of one 28 KiB span per wave all over the ROM.  With the filter's VALU work per dword when WORK.
template <int LOADS, int DEPTH, bool WORK>
__global__ __launch_bounds__(256) void read_sweep(const uint4 *p, uint64_t nchunks, uint32_t *sink, uint32_t k)
{
   const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
   const uint64_t ngroups = nchunks / (64 * LOADS);
   uint32_t acc = 0;
   uint4 w[DEPTH][LOADS];
   uint64_t g = wave;
#pragma unroll
   for (int d = 0; d < DEPTH; d++) {
      const uint64_t gg = g + d * nwaves < ngroups ? g + d * nwaves : wave;
#pragma unroll
      for (int u = 0; u < LOADS; u++) w[d][u] = p[(gg * LOADS + u) * 64 + lane];
   }
   for (; g < ngroups; g += nwaves * DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
         uint4 cur[LOADS];
#pragma unroll
         for (int u = 0; u < LOADS; u++) cur[u] = w[d][u];
         const uint64_t gn = g + (uint64_t)(d + DEPTH) * nwaves;
         const uint64_t gg = gn < ngroups ? gn : wave;
#pragma unroll
         for (int u = 0; u < LOADS; u++) w[d][u] = p[(gg * LOADS + u) * 64 + lane];
         if (g + (uint64_t)d * nwaves < ngroups) {
#pragma unroll
            for (int u = 0; u < LOADS; u++) {
               if (WORK) {
                  const uint32_t back = __builtin_amdgcn_update_dpp(k, cur[u].w, 0x138, 0xf, 0xf, false);
                  acc |= busy14(cur[u].x, back, k) | busy14(cur[u].y, cur[u].x, k) | busy14(cur[u].z, cur[u].y, k) | busy14(cur[u].w, cur[u].z, k);
               }
               else {
                  acc ^= cur[u].x ^ cur[u].y ^ cur[u].z ^ cur[u].w;
               }
            }
         }
      }
   }
   if (acc == k * 7u) sink[0] = acc;
}

// pattern B with the filter's work: a wave streams spans of SPAN_GROUPS 4 KiB groups, DEPTH groups ahead in flight
template <int DEPTH>
__global__ __launch_bounds__(256) void read_wavespan_work(const uint4 *p, uint64_t nchunks, uint64_t span_groups, uint32_t *sink, uint32_t k)
{
   const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
   const uint64_t ngroups = nchunks / 256;
   uint32_t acc = 0;
   for (uint64_t g0 = wave * span_groups; g0 < ngroups; g0 += nwaves * span_groups) {
      const uint64_t g1 = g0 + span_groups < ngroups ? g0 + span_groups : ngroups;
      uint4 w[DEPTH + 1][4];
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
         const uint64_t gg = g0 + d < g1 ? g0 + d : g1 - 1;
#pragma unroll
         for (int u = 0; u < 4; u++) w[d][u] = p[(gg * 4 + u) * 64 + lane];
      }
      for (uint64_t g = g0; g < g1; g += DEPTH + 1) {
#pragma unroll
         for (int s = 0; s <= DEPTH; s++) {
            const int slot_new = (s + DEPTH) % (DEPTH + 1);
            const uint64_t gn = g + s + DEPTH < g1 ? g + s + DEPTH : g1 - 1;
#pragma unroll
            for (int u = 0; u < 4; u++) w[slot_new][u] = p[(gn * 4 + u) * 64 + lane];
            if (g + s < g1) {
#pragma unroll
               for (int u = 0; u < 4; u++) {
                  const uint4 cur = w[s][u];
                  const uint32_t back = __builtin_amdgcn_update_dpp(k, cur.w, 0x138, 0xf, 0xf, false);
                  acc |= busy14(cur.x, back, k) | busy14(cur.y, cur.x, k) | busy14(cur.z, cur.y, k) | busy14(cur.w, cur.z, k);
               }
            }
         }
      }
   }
   if (acc == k * 7u) sink[0] = acc;
}

// pattern A plus the 4-byte look-back load the v1 filter does
template <int UNROLL>
__global__ __launch_bounds__(256) void read_gridstride_back(const uint4 *p, uint64_t nchunks, uint32_t *sink)
{
   uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
   uint32_t acc = 0;
   const uint32_t *p32 = reinterpret_cast<const uint32_t *>(p);
   for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; c + stride * (UNROLL - 1) < nchunks; c += stride * UNROLL) {
      uint4 w[UNROLL]; uint32_t b[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) { w[u] = p[c + stride * u]; b[u] = p32[(c + stride * u) * 4 - 1]; }
#pragma unroll
      for (int u = 0; u < UNROLL; u++) acc ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w ^ b[u];
   }
   if (acc == 0x12345678) sink[0] = acc;
}

template <class F>
void timeit(const char *name, uint64_t bytes, F launch)
{
   hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
   for (int i = 0; i < 3; i++) launch();
   CK(hipDeviceSynchronize());
   float best = 1e9, sum = 0; int n = 10;
   for (int i = 0; i < n; i++) {
      CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; sum += ms;
   }
   printf("%-44s avg %.3f ms  %.0f GB/s   best %.3f ms  %.0f GB/s\n", name, sum / n, bytes / (sum / n) / 1e6, best, bytes / best / 1e6);
}

__global__ void fill(uint4 *p, uint64_t n) {
   uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
   for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      uint32_t x = (uint32_t)(i * 2654435761u);
      p[i] = make_uint4(x, x ^ 0x9e3779b9u, x * 31u, x + 7u);
   }
}

int main(int argc, char **argv)
{
   double gib = argc > 1 ? atof(argv[1]) : 4.0;
   uint64_t bytes = (uint64_t)(gib * (1ull << 30));
   uint64_t nchunks = bytes / 16;
   uint4 *p; uint32_t *sink;
   CK(hipMalloc(&p, bytes)); CK(hipMalloc(&sink, 64));
   hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, p, nchunks);
   CK(hipDeviceSynchronize());
   const uint32_t kk = (uint32_t)argc * 0x01020304u;
   for (int grid : {1536, 1792}) {
      char nm[128];
#define SWEEP(L, D, W)                                                                                                              \
      snprintf(nm, sizeof nm, "sweep %d KiB per wave and round, %d rounds in flight%s, grid %d", L, D, W ? " + 14 VALU/dword" : "", grid); \
      timeit(nm, bytes, [&] { hipLaunchKernelGGL((read_sweep<L, D, W>), dim3(grid), dim3(256), 0, 0, p, nchunks, sink, kk); });
      SWEEP(1, 1, false) SWEEP(1, 2, false) SWEEP(1, 4, false) SWEEP(4, 1, false) SWEEP(4, 2, false) SWEEP(4, 3, false) SWEEP(2, 2, false)
      SWEEP(1, 2, true) SWEEP(1, 4, true) SWEEP(1, 8, true) SWEEP(4, 1, true) SWEEP(4, 2, true) SWEEP(4, 3, true) SWEEP(2, 2, true) SWEEP(2, 4, true)
      snprintf(nm, sizeof nm, "wavespan 28K + 14 VALU/dword, 2 groups ahead, grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_wavespan_work<2>, dim3(grid), dim3(256), 0, 0, p, nchunks, (uint64_t)7, sink, kk); });
      snprintf(nm, sizeof nm, "wavespan 28K + 14 VALU/dword, 1 group ahead, grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_wavespan_work<1>, dim3(grid), dim3(256), 0, 0, p, nchunks, (uint64_t)7, sink, kk); });
      snprintf(nm, sizeof nm, "wavespan 28K + 14 VALU/dword, 3 groups ahead, grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_wavespan_work<3>, dim3(grid), dim3(256), 0, 0, p, nchunks, (uint64_t)7, sink, kk); });
      snprintf(nm, sizeof nm, "wavespan 16K + 14 VALU/dword, 2 groups ahead, grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_wavespan_work<2>, dim3(grid), dim3(256), 0, 0, p, nchunks, (uint64_t)4, sink, kk); });
      for (uint64_t sg : {3, 5, 6, 9, 11, 13, 15}) {
         snprintf(nm, sizeof nm, "wavespan %lluK + 14 VALU/dword, 2 groups ahead, grid %d", (unsigned long long)(4 * sg), grid);
         timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_wavespan_work<2>, dim3(grid), dim3(256), 0, 0, p, nchunks, sg, sink, kk); });
      }
      snprintf(nm, sizeof nm, "wavespan 64K + 14 VALU/dword, 2 groups ahead, grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_wavespan_work<2>, dim3(grid), dim3(256), 0, 0, p, nchunks, (uint64_t)16, sink, kk); });
      snprintf(nm, sizeof nm, "gridstride u1 grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_gridstride<1>, dim3(grid), dim3(256), 0, 0, p, nchunks, sink); });
      snprintf(nm, sizeof nm, "gridstride u4 grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_gridstride<4>, dim3(grid), dim3(256), 0, 0, p, nchunks, sink); });
      snprintf(nm, sizeof nm, "gridstride u8 grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_gridstride<8>, dim3(grid), dim3(256), 0, 0, p, nchunks, sink); });
      snprintf(nm, sizeof nm, "gridstride u2 grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_gridstride<2>, dim3(grid), dim3(256), 0, 0, p, nchunks, sink); });
      snprintf(nm, sizeof nm, "gridstride + 14 VALU/dword, 1 load ahead, grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_gridstride_work<1>, dim3(grid), dim3(256), 0, 0, p, nchunks, sink, (uint32_t)argc * 0x01020304u); });
      snprintf(nm, sizeof nm, "gridstride + 14 VALU/dword, 2 loads ahead, grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_gridstride_work<2>, dim3(grid), dim3(256), 0, 0, p, nchunks, sink, (uint32_t)argc * 0x01020304u); });
      snprintf(nm, sizeof nm, "gridstride+back u4 grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_gridstride_back<4>, dim3(grid), dim3(256), 0, 0, p, nchunks, sink); });
      for (uint64_t span_kb : {16, 64, 512}) {
         snprintf(nm, sizeof nm, "wavespan u4 span %lluK grid %d", (unsigned long long)span_kb, grid);
         timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_wavespan<4>, dim3(grid), dim3(256), 0, 0, p, nchunks, span_kb * 64, sink); });
      }
      snprintf(nm, sizeof nm, "wavespan u8 span 64K grid %d", grid);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(read_wavespan<8>, dim3(grid), dim3(256), 0, 0, p, nchunks, 64 * 64, sink); });
   }
   return 0;
}
