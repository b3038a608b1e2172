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
//   mm_dense_emit    every tile again, now with its TRUE entry phase: the scalar unit
//                    tracks the one live phase through the exceptional positions and hands
//                    each group of 64 positions its phase; lane t then walks group t
//                    (a handful of jumps) and collects the visited positions where the
//                    compare loop matched.  One atomic per tile reserves the output range.
#ifndef MM_DENSE_H
#define MM_DENSE_H

constexpr int MM_SUPER = 256;                 // tiles per super-tile

struct MmDenseArgs {
   MmTileArgs t;
   uint64_t ndom;           // domains: nblocks * S in engine mode, 1 in whole-buffer mode
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
   const uint64_t b = a.t.g.whole ? 0 : dom / a.t.g.S;
   const uint32_t p = a.t.g.whole ? 0 : (uint32_t)(dom % a.t.g.S);
   *start = mm_domain_start(a.t.g, b, p);
   *nv = mm_domain_nv(a.t.g, b, p);
}

template <int BITS>
__device__ __forceinline__ void mm_dense_maps_body(const MmDenseArgs &a, const MmPlanLds &P, MmWaveLds &W)
{
   const int D = (int)a.t.plan.L - 1;
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   const uint64_t ntiles = a.ndom * a.tpd;
   const uint64_t nwaves = (uint64_t)gridDim.x * MM_WAVES;
   for (uint64_t item = (uint64_t)blockIdx.x * MM_WAVES + wave; item < ntiles; item += nwaves) {
      const uint64_t dom = item / a.tpd;
      const int64_t lo = (int64_t)(item % a.tpd) * MM_TILE;
      uint64_t start; int64_t nv;
      mm_dense_domain(a, dom, &start, &nv);
      MmPhaseMap<BITS> M;
      M.identity();
      if (lo < nv) {
         const int npos = (int)(nv - lo < MM_TILE ? nv - lo : MM_TILE);
         mm_tile_map<BITS>(a.t, P, W, start, lo, npos, (uint32_t)(lo % D), lane, M);
      }
      if (lane < MM_MAXD) {
         a.maps[item * MM_MAXD + lane] = (uint8_t)M.get(lane);
      }
   }
}

__global__ __launch_bounds__(64 * MM_WAVES) void mm_dense_maps(MmDenseArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLds Wv[MM_WAVES];
   mm_plan_to_lds(P, a.t.plan);
   if (a.t.plan.L - 1 <= 16) {
      mm_dense_maps_body<4>(a, P, Wv[threadIdx.x >> 6]);
   }
   else {
      mm_dense_maps_body<8>(a, P, Wv[threadIdx.x >> 6]);
   }
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

struct MmEmitLds {
   uint8_t jbuf[MM_TILE];                     // J of exceptional positions
   uint16_t found[64][64];                    // per lane: tile positions of its group's matches
};

__global__ __launch_bounds__(64 * MM_WAVES) void mm_dense_emit(MmDenseArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLds Wv[MM_WAVES];
   __shared__ MmEmitLds Ev[MM_WAVES];
   mm_plan_to_lds(P, a.t.plan);

   const int L = (int)a.t.plan.L, S = (int)a.t.g.S;
   const uint32_t D = (uint32_t)(L - 1);
   const bool be = a.t.g.big_endian != 0;
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   MmWaveLds &W = Wv[wave];
   MmEmitLds &E = Ev[wave];
   const int e1 = a.t.plan.expected[L - 1], b1 = a.t.plan.bridge[L - 1], w1 = a.t.plan.wst[L - 1];
   const uint32_t m1 = a.t.plan.cmp_mask[L - 1];
   const int match_jump = (int)a.t.plan.match_jump;
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
      const int mis = mm_stage_tile(a.t, W, start, lo, npos, lane);
      mm_wave_sync();
      const uint8_t *tile = reinterpret_cast<const uint8_t *>(W.tile) + mis;

      // jumps + match flags of every position; the live phase through the exceptions
      uint32_t cur = a.entry[item];                       // phase of the next visited position
      uint32_t ph0 = mm_modd(a.t, (uint32_t)(lo % D));    // phase of position 64t
      uint32_t my_phase = 0;                              // lane t: live phase at the start of group t
      unsigned long long my_exc = 0, my_match = 0;
      const int nballots = (npos + 63) >> 6;
      for (int t = 0; t < nballots; t++) {
         const int q = 64 * t + lane;
         int J = (int)D;
         bool mt = false;
         if (q < npos) {
            const int c = mm_tile_elem(tile, q + L - 1, S, be);
            const int pv = mm_tile_elem(tile, q + L - 1 + b1, S, be);
            const int d = c - pv;
            if (((uint32_t)(d ^ e1) & m1) != 0) {
               const int s = mm_tile_skip(a.t, P, d);
               J = s < w1 ? s : w1;
            }
            else {
               J = match_jump;
               mt = true;
               for (int i = L - 2; i >= 0; --i) {
                  const int ci = mm_tile_elem(tile, q + i, S, be);
                  const int pi = mm_tile_elem(tile, q + i + P.bridge[i], S, be);
                  const int di = ci - pi;
                  if (((uint32_t)(di ^ P.expected[i]) & P.cmp_mask[i]) != 0) {
                     const int s = mm_tile_skip(a.t, P, di);
                     const int w = P.wst[i];
                     J = s < w ? s : w;
                     mt = false;
                     break;
                  }
               }
            }
         }
         const bool exc = J != (int)D;
         if (exc) {
            E.jbuf[q] = (uint8_t)J;
         }
         const unsigned long long em = __ballot(exc);
         const unsigned long long mm = __ballot(mt);
         if (lane == t) {
            my_exc = em;
            my_match = mm;
            my_phase = cur;
         }
         // only the exceptions the live chain actually stands on move it
         unsigned long long mask = em;
         while (mask) {
            const int bit = __builtin_ctzll(mask);
            mask &= mask - 1;
            const uint32_t r = mm_modd(a.t, ph0 + (uint32_t)bit);
            if (r == cur) {
               const uint32_t Jb = (uint32_t)__builtin_amdgcn_readlane(J, bit);
               uint32_t r2 = r + Jb;
               cur = r2 >= D ? r2 - D : r2;
            }
         }
         ph0 = mm_modd(a.t, ph0 + 64u);
      }
      mm_wave_sync();

      // lane t walks group t from the first position in phase my_phase
      int nfound = 0;
      if (lane < nballots) {
         const uint32_t g_phase = mm_modd(a.t, mm_modd(a.t, (uint32_t)lane) * mm_modd(a.t, 64u) + (uint32_t)(lo % D));
         uint32_t off = my_phase + D - g_phase;           // (my_phase - g_phase) mod D
         off = off >= D ? off - D : off;
         int pos = 64 * lane + (int)off;
         const int end = 64 * lane + 64 < npos ? 64 * lane + 64 : npos;
         while (pos < end) {
            const int bit = pos & 63;
            if ((my_match >> bit) & 1) {
               E.found[lane][nfound++] = (uint16_t)pos;
            }
            pos += ((my_exc >> bit) & 1) ? E.jbuf[pos] : (int)D;
         }
      }
      // one atomic per tile reserves the output range; lanes write their finds in order
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
            const unsigned long long slot = base + k;
            if (slot < a.list_cap) {
               const uint64_t j = (uint64_t)lo + E.found[lane][k];
               a.out[(uint64_t)list * a.list_cap + slot] =
                  a.t.g.whole ? j : start + j * a.t.g.S + a.base_offset;
            }
         }
      }
      mm_wave_sync();
   }
}

#endif
