// mm_dense.h -- the candidate-free forward engine: exact emulation of every chain of the
// ROM, in parallel.  Included by mm_kernels.hip after mm_tiles.h.
//
// Used when the per-candidate resolvers do not apply: candidate sets too dense (constant
// data with a constant-delta keyword matches every L-1 positions), patterns the
// SWAR filter cannot key on (no two adjacent literals), prefixes too long for
// mm_hard_resolve.  Cost is linear in the ROM, independent of the data.
//
//   mm_dense_maps    every tile of every domain -> its phase map (mm_tile_map), 32 B each
//   mm_dense_super   every super-tile (256 tiles) -> the composition of its maps
//   mm_dense_entries per domain: walk the super-tile maps from phase 0 (the chain starts
//                    at the domain's first position, monkey_moore.cpp:329) -> entry phase
//                    of each super-tile
//   mm_dense_tiles   per super-tile: walk its tile maps from that phase -> entry phase of
//                    every tile
//   mm_dense_emit    every tile again, now with its TRUE entry phase: jumps and match flags
//                    of all positions in parallel (mm_tile_jumps), the phase maps of the
//                    tile's 32 groups of 64 positions in parallel (mm_group_maps), one lane
//                    threads the entry phase through them, then lane g walks group g (a
//                    handful of jumps) and notes the visited positions where the compare loop
//                    matched.  One atomic per tile reserves the output range.
#ifndef MM_DENSE_H
#define MM_DENSE_H

constexpr int MM_SUPER = 256;                 // tiles per super-tile

struct MmDenseArgs {
   MmTileArgs t;
   uint64_t ndom;           // domains worked on: nblocks * S in engine mode, 1 in whole-buffer mode, or the length of dom_list
   const uint32_t *dom_list;   // nullptr: every domain; else the domains to work on (buffers are indexed by list position)
   uint32_t tpd;            // tiles per domain (ceil(max positions / MM_TILE))
   uint32_t nsup;           // super-tiles per domain
   uint8_t *maps;           // [ndom * tpd][MM_MAXD]
   uint8_t *supmaps;        // [ndom * nsup][MM_MAXD]
   uint8_t *supentry;       // [ndom * nsup]
   uint8_t *entry;          // [ndom * tpd]
   uint64_t *out;           // MM_CAND_LISTS output lists of list_cap values
   unsigned long long *list_count;   // their counters, MM_LIST_STRIDE words apart
   uint64_t list_cap;
   uint64_t base_offset;
};

__device__ __forceinline__ void mm_dense_domain(const MmDenseArgs &a, uint64_t dom, uint64_t *start, int64_t *nv)
{
   if (a.dom_list) {
      dom = a.dom_list[dom];
   }
   const uint64_t b = a.t.g.whole ? 0 : dom / a.t.g.S;
   const uint32_t p = a.t.g.whole ? 0 : (uint32_t)(dom % a.t.g.S);
   *start = mm_domain_start(a.t.g, b, p);
   *nv = mm_domain_nv(a.t.g, b, p);
}

__device__ __forceinline__ void mm_dense_maps_body(const MmDenseArgs &a, const MmPlanLds &P, MmWaveLds &W)
{
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   const uint64_t ntiles = a.ndom * a.tpd;
   const uint64_t nwaves = (uint64_t)gridDim.x * MM_WAVES;
   for (uint64_t item = (uint64_t)blockIdx.x * MM_WAVES + wave; item < ntiles; item += nwaves) {
      const uint64_t dom = item / a.tpd;
      const int64_t lo = (int64_t)(item % a.tpd) * MM_TILE;
      uint64_t start; int64_t nv;
      mm_dense_domain(a, dom, &start, &nv);
      uint32_t map = (uint32_t)lane;                      // tiles past the domain's end: identity
      if (lo < nv) {
         const int npos = (int)(nv - lo < MM_TILE ? nv - lo : MM_TILE);
         map = mm_tile_map(a.t, P, W, start, lo, npos, mm_modd64(a.t, (uint64_t)lo), lane);
      }
      if (lane < MM_MAXD) {
         a.maps[item * MM_MAXD + lane] = (uint8_t)map;
      }
   }
}

__global__ __launch_bounds__(64 * MM_WAVES) void mm_dense_maps(MmDenseArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLds Wv[MM_WAVES];
   mm_plan_to_lds(P, a.t.plan);
   mm_dense_maps_body(a, P, Wv[threadIdx.x >> 6]);
}

// one workgroup (one wave) per super-tile.  MODE 0: compose the tile maps -> supmaps.
// MODE 1: walk them from the super-tile's entry phase -> entry[] of every tile.
template <int MODE>
__global__ __launch_bounds__(64) void mm_dense_super(MmDenseArgs a)
{
   __shared__ uint32_t chunk[MM_SUPER * MM_MAXD / 4];
   const uint8_t *cm = reinterpret_cast<const uint8_t *>(chunk);
   const int lane = threadIdx.x;
   const uint64_t nitems = a.ndom * a.nsup;
   for (uint64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
      const uint64_t dom = item / a.nsup;
      const uint32_t sup = (uint32_t)(item % a.nsup);
      const uint32_t t0 = sup * MM_SUPER;
      const uint32_t n = a.tpd - t0 < MM_SUPER ? a.tpd - t0 : MM_SUPER;
      const uint32_t *src = reinterpret_cast<const uint32_t *>(a.maps + (dom * a.tpd + t0) * MM_MAXD);
      __syncthreads();
      for (uint32_t k = lane; k < n * (MM_MAXD / 4); k += 64) {
         chunk[k] = src[k];
      }
      __syncthreads();
      if (MODE == 0) {
         // lane e follows entry phase e through the n maps
         int v = lane & (MM_MAXD - 1);
         for (uint32_t k = 0; k < n; k++) {
            v = cm[k * MM_MAXD + v];
         }
         if (lane < MM_MAXD) {
            a.supmaps[item * MM_MAXD + lane] = (uint8_t)v;
         }
      }
      else if (lane == 0) {
         int v = a.supentry[item];
         for (uint32_t k = 0; k < n; k++) {
            a.entry[dom * a.tpd + t0 + k] = (uint8_t)v;
            v = cm[k * MM_MAXD + v];
         }
      }
   }
}

// one wave per domain: entry phase of each super-tile
__global__ __launch_bounds__(64) void mm_dense_entries(MmDenseArgs a)
{
   __shared__ uint32_t chunk[MM_SUPER * MM_MAXD / 4];
   const uint8_t *cm = reinterpret_cast<const uint8_t *>(chunk);
   const int lane = threadIdx.x;
   for (uint64_t dom = blockIdx.x; dom < a.ndom; dom += gridDim.x) {
      int v = 0;                                          // the chain of a domain starts at its first position
      for (uint32_t s0 = 0; s0 < a.nsup; s0 += MM_SUPER) {
         const uint32_t n = a.nsup - s0 < MM_SUPER ? a.nsup - s0 : MM_SUPER;
         const uint32_t *src = reinterpret_cast<const uint32_t *>(a.supmaps + (dom * a.nsup + s0) * MM_MAXD);
         __syncthreads();
         for (uint32_t k = lane; k < n * (MM_MAXD / 4); k += 64) {
            chunk[k] = src[k];
         }
         __syncthreads();
         if (lane == 0) {
            for (uint32_t k = 0; k < n; k++) {
               a.supentry[dom * a.nsup + s0 + k] = (uint8_t)v;
               v = cm[k * MM_MAXD + v];
            }
         }
         v = __builtin_amdgcn_readfirstlane(v);
      }
   }
}

__global__ __launch_bounds__(64 * MM_WAVES) void mm_dense_emit(MmDenseArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLds Wv[MM_WAVES];
   mm_plan_to_lds(P, a.t.plan);

   const uint32_t D = a.t.plan.L - 1;
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   MmWaveLds &W = Wv[wave];
   // tile positions of the matches on the chain, ascending: they overwrite the staged bytes,
   // which nobody needs once the jumps are known
   static_assert(sizeof(W.tile) >= MM_TILE * sizeof(uint16_t), "found[] must fit the tile buffer");
   uint16_t *found = reinterpret_cast<uint16_t *>(W.tile);
   const uint64_t ntiles = a.ndom * a.tpd;
   const uint64_t nwaves = (uint64_t)gridDim.x * MM_WAVES;
   const uint32_t list = blockIdx.x & (MM_CAND_LISTS - 1);

   for (uint64_t item = (uint64_t)blockIdx.x * MM_WAVES + wave; item < ntiles; item += nwaves) {
      const uint64_t dom = item / a.tpd;
      const int64_t lo = (int64_t)(item % a.tpd) * MM_TILE;
      uint64_t start; int64_t nv;
      mm_dense_domain(a, dom, &start, &nv);
      if (lo >= nv) {
         continue;
      }
      const int npos = (int)(nv - lo < MM_TILE ? nv - lo : MM_TILE);
      mm_tile_jumps(a.t, P, W, mm_uniform64(start), lo, npos, lane);

      // The one real chain enters the tile in phase entry[item].  Its phase on entering every
      // group of 64 positions comes from the group maps; then lane g walks group g and notes
      // the visited positions where the compare loop matched.
      const uint32_t lo_mod = mm_modd64(a.t, (uint64_t)lo);
      const int ngroups = (npos + 63) >> 6;
      mm_group_maps(a.t, W, npos, lo_mod, lane);
      if (lane == 0) {
         uint32_t ph = a.entry[item];
         for (int g = 0; g < ngroups; g++) {
            W.gentry[g] = (uint8_t)ph;
            ph = W.gmap[g][ph];
         }
      }
      mm_wave_sync();
      int nfound = 0;
      if (lane < ngroups) {
         const uint32_t first = 64u * (uint32_t)lane;
         const uint32_t end = first + 64 < (uint32_t)npos ? first + 64 : (uint32_t)npos;
         uint32_t off = (uint32_t)W.gentry[lane] + D - mm_modd(a.t, lo_mod + first);
         off = off >= D ? off - D : off;
         uint32_t p = first + off;
         while (p < end) {
            const uint32_t j = W.jump[p];
            if (j & MM_JUMP_MATCH) {
               found[first + nfound++] = (uint16_t)p;      // group g's finds live in found[64 g ...]
            }
            p += j & (MM_JUMP_MATCH - 1);
         }
      }
      // one atomic per tile reserves the output range; lanes copy their finds in group order
      int incl = nfound;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
         const int v = __shfl_up(incl, d);
         incl += lane >= d ? v : 0;
      }
      const int total = __shfl(incl, 63);
      if (total) {
         unsigned long long base = 0;
         if (lane == 0) {
            base = atomicAdd(a.list_count + list * MM_LIST_STRIDE, (unsigned long long)total);
         }
         base = __shfl(base, 0) + (unsigned long long)(incl - nfound);
         for (int k = 0; k < nfound; k++) {
            const unsigned long long slot = base + (unsigned long long)k;
            if (slot < a.list_cap) {
               const uint64_t j = (uint64_t)lo + found[64 * lane + k];
               a.out[(uint64_t)list * a.list_cap + slot] = a.t.g.whole ? j : start + j * a.t.g.S + a.base_offset;
            }
         }
      }
      mm_wave_sync();
   }
}

#endif
