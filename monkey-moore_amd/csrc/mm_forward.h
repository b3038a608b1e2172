// mm_forward.h -- the forward engine, second generation: exact emulation of every chain of the
// ROM in ONE pass over it.  Included by mm_kernels.hip after mm_tiles.h (device code only).
//
// What it is for (unchanged): inputs the per-candidate path does not suit -- patterns without a
// SWAR key, candidate floods, prefixes too long for mm_hard_resolve, whole domains flagged by the
// resolvers.  Cost linear in the ROM, independent of the data.
//
// What changed against mm_dense.h (round 1: 4.2 ms per GiB, two passes over the ROM, five
// launches; profiles/r02_dense_before_*.txt showed the kernels bound by instruction issue --
// 40 VALU + 41 SALU + 18 branch instructions per 64 positions, most of them in the per-position
// jump loop -- not by LDS latency):
//   * the jump of a position comes from ONE LDS table lookup on its last delta (jump1: skip
//     table, wildcard cap and "compare on" flag folded into 511 bytes), four positions per lane
//     and iteration, straight from the staged dwords; the 1/256 positions whose first compare
//     holds take a second table (jump2), only what passes both runs the compare loop;
//   * one pass: a wave owns a BATCH of MM_FWD_BATCH consecutive tiles of one domain, maps them
//     (phase maps as before: Z_D -> Z_D per tile, composed per batch) and publishes the batch's
//     map; the phase in which the chain ENTERS the batch comes from decoupled look-back over the
//     published maps (Merrill & Garland's single-pass scan, with function composition in place of
//     addition).  A batch whose map is constant -- on ordinary data the chains of a 32 KiB batch
//     have long merged -- publishes its exit phase at once and nobody ever waits for it;
//   * tiles are only walked for matches when one of their positions passed the whole compare
//     loop (rare): then the tile is staged again with the now known entry phase.
// Batches are handed out through one ticket per workgroup and four batches, so a waiting wave
// only ever waits for batches that running waves own: no residency assumption, no deadlock.
#ifndef MM_FORWARD_H
#define MM_FORWARD_H

constexpr int MM_FWD_BATCH = 16;               // tiles per batch (one wave): 32 Ki positions
constexpr unsigned long long MM_FWD_AGGREGATE = 1, MM_FWD_INCLUSIVE = 2;

struct MmForwardArgs {
   MmTileArgs t;
   uint64_t ndom;            // domains worked on: nblocks * S in engine mode, 1 in whole-buffer mode, or the length of dom_list
   const uint32_t *dom_list; // nullptr: every domain; else the domains to work on
   uint32_t tpd;             // tiles per domain
   uint32_t bpd;             // batches per domain
   uint8_t *agg;             // [ndom * bpd][MM_MAXD] published batch maps
   unsigned long long *status;   // [ndom * bpd] look-back words (zeroed before the launch): state | exit phase << 8
   unsigned long long *ticket;   // next batch to hand out (zeroed before the launch)
   uint64_t *out;            // MM_CAND_LISTS output lists of list_cap values
   unsigned long long *list_count;
   uint64_t list_cap;
   uint64_t base_offset;
   // fast jump path (8-bit elements, first compare against an element at most 4 to the left)
   uint32_t fast;
   uint32_t i1, g1;          // keyword position of the first compare and the distance to its partner
   uint32_t has2, i2, g2;    // the same for the next compare down, when it qualifies
};

struct MmFwdTables {
   uint8_t jump1[512];       // [d + 255]: jump of a mismatch at i1 with delta d; 0x80: the compare holds
   uint8_t jump2[512];       // the same at i2
};

// block-cooperative; needs P; ends with a __syncthreads()
__device__ __forceinline__ void mm_fwd_tables(MmFwdTables &T, const MmPlanLds &P, const MmForwardArgs &a)
{
   if (a.fast) {
      for (int idx = threadIdx.x; idx < 1024; idx += blockDim.x) {
         const bool second = idx >= 512;
         if (second && !a.has2) {
            continue;
         }
         const int i = (int)(second ? a.i2 : a.i1);
         const int d = (idx & 511) - 255;
         uint8_t v = 1;
         if ((idx & 511) < 511) {
            if (((uint32_t)(d ^ P.expected[i]) & P.cmp_mask[i]) == 0) {
               v = MM_JUMP_MATCH;                 // here: "this compare holds, look further"
            }
            else {
               const int s = P.skip8[d + 255];
               const int w = P.wst[i];
               v = (uint8_t)(s < w ? s : w);
            }
         }
         (second ? T.jump2 : T.jump1)[idx & 511] = v;
      }
   }
   __syncthreads();
}

// the reference's compare loop at position q of the staged tile, from keyword position `from`
// downwards (everything above already holds): the jump, | MM_JUMP_MATCH when the loop reports a match
__device__ __forceinline__ int mm_deep_jump(const MmTileArgs &a, const MmPlanLds &P, const uint8_t *tile, int q, int from)
{
   const int S = (int)a.g.S;
   const bool be = a.g.big_endian != 0;
   for (int i = from; i >= 0; --i) {
      const int ci = mm_tile_elem(tile, q + i, S, be);
      const int pi = mm_tile_elem(tile, q + i + P.bridge[i], S, be);
      const int di = ci - pi;
      if (((uint32_t)(di ^ P.expected[i]) & P.cmp_mask[i]) != 0) {
         const int s = mm_tile_skip(a, P, di);
         const int w = P.wst[i];
         const int J = s < w ? s : w;
         return J < 1 ? 1 : J;
      }
   }
   return (int)a.plan.match_jump | MM_JUMP_MATCH;
}

// Stage positions [lo, lo + npos) of the domain at byte `start` and leave the jump of every
// position in LDS.  Returns the jump array (J[p], p in [0, npos): W.jump shifted by 0..3 bytes so
// that four positions' jumps are stored as one dword) and, in *tile_out, the staged tile's first
// byte; *any_match: some position passed the whole compare loop.
__device__ __forceinline__ const uint8_t *mm_fwd_jumps(const MmForwardArgs &a, const MmPlanLds &P, const MmFwdTables &T, MmWaveLds &W,
                                                       uint64_t start, int64_t lo, int npos, int lane, bool *any_match,
                                                       const uint8_t **tile_out)
{
   if (!a.fast) {
      const uint8_t *tile = mm_tile_jumps(a.t, P, W, start, lo, npos, lane);
      bool m = false;
      for (int q = lane; q < npos; q += 64) {
         m = m || (W.jump[q] & MM_JUMP_MATCH) != 0;
      }
      *any_match = __ballot(m) != 0;
      *tile_out = tile;
      return W.jump;
   }
   const int mis = mm_stage_tile(a.t, W, start, lo, npos, lane);
   mm_wave_sync();
   const uint8_t *tile = reinterpret_cast<const uint8_t *>(W.tile) + mis;
   // position p compares LDS byte p + base with the byte g1 in front of it; LDS dword b4 + u holds
   // the compared bytes of positions 4u - r + k, k = 0..3, whose jumps form dword u of J - r
   const int base = mis + (int)a.i1;
   const int r = base & 3, b4 = base >> 2;
   const int ndw = (npos + r + 3) >> 2;
   uint32_t *jump32 = reinterpret_cast<uint32_t *>(W.jump);
   const uint32_t sh1 = 32u - 8u * a.g1;
   bool m = false;
   for (int u = lane; u < ndw; u += 64) {
      const int mdw = b4 + u;
      const uint32_t w = W.tile[mdw];
      const uint32_t wp = mdw > 0 ? W.tile[mdw - 1] : 0u;
      const uint32_t wq = mm_alignbit(w, wp, sh1);            // byte k: the partner of w's byte k
      uint32_t jj = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
         const int d = (int)((w >> (8 * k)) & 0xFF) - (int)((wq >> (8 * k)) & 0xFF);
         jj |= (uint32_t)T.jump1[d + 255] << (8 * k);
      }
      // positions outside [0, npos) (in front of the tile in dword 0, behind it in the last one)
      const int p0 = 4 * u - r;
      uint32_t valid = 0xFFFFFFFFu;
      if (p0 < 0) {
         valid <<= 8 * (-p0);
      }
      if (p0 + 4 > npos) {
         valid &= p0 >= npos ? 0u : 0xFFFFFFFFu >> (8 * (p0 + 4 - npos));
      }
      uint32_t on = jj & valid & 0x80808080u;                 // first compare holds: 1/256 of the positions
      if (__ballot(on != 0) != 0) {
         while (on) {
            const int k = (__ffs((int)on) - 1) >> 3;
            on &= on - 1;
            const int q = p0 + k;
            int J;
            if (a.has2) {
               const int c2 = tile[q + (int)a.i2], p2 = tile[q + (int)a.i2 - (int)a.g2];
               J = T.jump2[c2 - p2 + 255];
               if (J & MM_JUMP_MATCH) {
                  J = mm_deep_jump(a.t, P, tile, q, (int)a.i2 - 1);
               }
            }
            else {
               J = mm_deep_jump(a.t, P, tile, q, (int)a.i1 - 1);
            }
            jj = (jj & ~(0xFFu << (8 * k))) | ((uint32_t)J << (8 * k));
         }
      }
      m = m || (jj & valid & 0x80808080u) != 0;
      jump32[u] = jj;
   }
   *any_match = __ballot(m) != 0;
   *tile_out = tile;
   mm_wave_sync();
   return W.jump + r;
}

// phase map of positions [0, npos) (domain positions lo ..., lo_mod = lo mod D) from their jumps J:
// lane e < D returns the exit phase of entry phase e (group maps in parallel, then composed)
__device__ __forceinline__ uint32_t mm_fwd_map(const MmTileArgs &a, MmWaveLds &W, const uint8_t *J, int npos, uint32_t lo_mod, int lane)
{
   const uint32_t D = a.plan.L - 1;
   mm_group_maps(a, W, npos, lo_mod, lane, J);
   uint32_t v = (uint32_t)lane;
   if ((uint32_t)lane < D) {
      const int ngroups = (npos + 63) >> 6;
      for (int g = 0; g < ngroups; g++) {
         v = W.gmap[g][v];
      }
   }
   return v;
}

__device__ __forceinline__ void mm_fwd_domain(const MmForwardArgs &a, uint64_t dom, uint64_t *start, int64_t *nv)
{
   if (a.dom_list) {
      dom = a.dom_list[dom];
   }
   const uint64_t b = a.t.g.whole ? 0 : dom / a.t.g.S;
   const uint32_t p = a.t.g.whole ? 0 : (uint32_t)(dom % a.t.g.S);
   *start = mm_domain_start(a.t.g, b, p);
   *nv = mm_domain_nv(a.t.g, b, p);
}

// the matches on the chain inside one tile, given the phase in which the chain enters it
__device__ __forceinline__ void mm_fwd_emit(const MmForwardArgs &a, MmWaveLds &W, const uint8_t *J, uint64_t start, int64_t lo, int npos,
                                            uint32_t entry, int lane)
{
   const uint32_t D = a.t.plan.L - 1;
   uint16_t *found = reinterpret_cast<uint16_t *>(W.tile);      // overwrites the staged bytes: nobody needs them any more
   const uint32_t list = blockIdx.x & (MM_CAND_LISTS - 1);
   const uint32_t lo_mod = mm_modd64(a.t, (uint64_t)lo);
   const int ngroups = (npos + 63) >> 6;
   mm_group_maps(a.t, W, npos, lo_mod, lane, J);
   if (lane == 0) {
      uint32_t ph = entry;
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
         const uint32_t j = J[p];
         if (j & MM_JUMP_MATCH) {
            found[first + nfound++] = (uint16_t)p;            // group g's finds live in found[64 g ...]
         }
         p += j & (MM_JUMP_MATCH - 1);
      }
   }
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

// the look-back word of batch b, once it is non-zero (wave uniform)
__device__ __forceinline__ unsigned long long mm_fwd_wait(const unsigned long long *status, uint64_t b, int lane)
{
   unsigned long long s = 0;
   if (lane == 0) {
      while ((s = __hip_atomic_load(status + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
         __builtin_amdgcn_s_sleep(4);
      }
   }
   return mm_uniform64(s);
}

__global__ __launch_bounds__(64 * MM_WAVES) void mm_forward(MmForwardArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmFwdTables T;
   __shared__ MmWaveLds Wv[MM_WAVES];
   __shared__ uint8_t tilemap[MM_WAVES][MM_FWD_BATCH][MM_MAXD];
   __shared__ unsigned long long next_batch;
   mm_plan_to_lds(P, a.t.plan);
   mm_fwd_tables(T, P, a);

   const uint32_t D = a.t.plan.L - 1;
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   MmWaveLds &W = Wv[wave];
   const uint64_t nbatches = a.ndom * a.bpd;

   for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) {
         next_batch = atomicAdd(a.ticket, (unsigned long long)MM_WAVES);
      }
      __syncthreads();
      const uint64_t item = next_batch + (uint64_t)wave;
      if (next_batch >= nbatches) {
         break;
      }
      if (item >= nbatches) {
         continue;
      }
      const uint64_t dom = item / a.bpd;
      const uint32_t b = (uint32_t)(item % a.bpd);
      uint64_t start; int64_t nv;
      mm_fwd_domain(a, dom, &start, &nv);
      start = mm_uniform64(start);
      const uint32_t t0 = b * MM_FWD_BATCH;
      const uint32_t t1 = t0 + MM_FWD_BATCH < a.tpd ? t0 + MM_FWD_BATCH : a.tpd;

      // ---- pass 1: the map of every tile of the batch, composed into the batch's map ----------
      uint32_t bm = (uint32_t)lane;                 // lane e < D: where entry phase e leaves the batch so far
      uint32_t flagged = 0;                         // tiles with a position that passed the whole compare loop
      for (uint32_t t = t0; t < t1; t++) {
         const int64_t lo = (int64_t)t * MM_TILE;
         uint32_t map = (uint32_t)lane;             // tiles past the domain's end: identity
         if (lo < nv) {
            const int npos = (int)(nv - lo < MM_TILE ? nv - lo : MM_TILE);
            bool any = false;
            const uint8_t *tile;
            const uint8_t *J = mm_fwd_jumps(a, P, T, W, start, lo, npos, lane, &any, &tile);
            map = mm_fwd_map(a.t, W, J, npos, mm_modd64(a.t, (uint64_t)lo), lane);
            flagged |= any ? 1u << (t - t0) : 0u;
            mm_wave_sync();
         }
         if (lane < MM_MAXD) {
            tilemap[wave][t - t0][lane] = (uint8_t)map;
         }
         mm_wave_sync();
         if ((uint32_t)lane < D) {
            bm = tilemap[wave][t - t0][bm];
         }
      }
      // ---- publish; find the phase in which the chain enters the batch --------------------------
      const uint32_t bm0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)bm);
      const bool constant = __ballot((uint32_t)lane < D && bm != bm0) == 0;
      uint32_t entry = 0;                           // first batch of a domain: the chain starts at its first position
      const bool need_entry = b != 0 && (flagged != 0 || !constant);
      if (b == 0 || constant) {
         const uint32_t exit_phase = bm0;           // (b == 0: the chain enters in phase 0, and lane 0 is the first lane)
         if (lane == 0) {
            __hip_atomic_store(a.status + item, MM_FWD_INCLUSIVE | ((unsigned long long)exit_phase << 8), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
         }
      }
      else {
         if (lane < MM_MAXD) {
            __hip_atomic_store(a.agg + item * MM_MAXD + lane, (uint8_t)bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         }
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
         if (lane == 0) {
            __hip_atomic_store(a.status + item, MM_FWD_AGGREGATE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         }
      }
      if (need_entry) {
         // decoupled look-back: f[e] = the phase at OUR entry when the chain enters batch k+1 in phase e
         uint32_t f = (uint32_t)lane;
         for (uint64_t k = item - 1;; k--) {
            const unsigned long long s = mm_fwd_wait(a.status, k, lane);
            if ((s & 3) == MM_FWD_INCLUSIVE) {
               entry = (uint32_t)__shfl((int)f, (int)((s >> 8) & 0xFF));
               break;
            }
            uint32_t mk = (uint32_t)lane;
            if (lane < MM_MAXD) {
               mk = __hip_atomic_load(a.agg + k * MM_MAXD + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            f = (uint32_t)__shfl((int)f, (int)(mk & 63));
            const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
            if (__ballot((uint32_t)lane < D && f != f0) == 0) {
               entry = f0;                          // every entry phase of batch k ends up here: no need to go further back
               break;
            }
         }
         entry = mm_uniform(entry);
      }
      if (b != 0 && !constant) {
         // now that the entry is known, tell the batches behind us where the chain leaves this one
         const uint32_t exit_phase = (uint32_t)__shfl((int)bm, (int)entry);
         if (lane == 0) {
            __hip_atomic_store(a.status + item, MM_FWD_INCLUSIVE | ((unsigned long long)exit_phase << 8), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
         }
      }
      // ---- pass 2 (rare): the tiles that hold a full match, walked with their true entry phase --
      if (flagged) {
         uint32_t ph = entry;
         for (uint32_t t = t0; t < t1; t++) {
            if ((flagged >> (t - t0)) & 1u) {
               const int64_t lo = (int64_t)t * MM_TILE;
               const int npos = (int)(nv - lo < MM_TILE ? nv - lo : MM_TILE);
               bool any = false;
               const uint8_t *tile;
               const uint8_t *J = mm_fwd_jumps(a, P, T, W, start, lo, npos, lane, &any, &tile);
               mm_fwd_emit(a, W, J, start, lo, npos, ph, lane);
            }
            ph = tilemap[wave][t - t0][ph];
         }
      }
   }
}

#endif
