// SPDX-License-Identifier: GPL-3.0-or-later
// mm_forward.h -- the forward engine, second generation: exact emulation of every chain of the
// ROM in ONE pass over it.  Included by mm_kernels.hip after mm_tiles.h (device code only).
//
// What it is for (unchanged): inputs the per-candidate path does not suit -- patterns without a
// SWAR key, candidate floods, prefixes too long for mm_hard_resolve, whole domains flagged by the
// resolvers.  Cost at most linear in the ROM, whatever the data.
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
//   * a tile's phase map comes from EXIT TABLES (mm_fwd_map), 32 steps of one LDS read each
//     whatever the pattern, instead of walking each of the D phases' chains;
//   * tiles are only walked for matches when one of their positions passed the whole compare
//     loop (rare): then the tile is staged again with the now known entry phase;
//   * keywords of up to MMH_MAX_KEYWORD = 128 symbols (D <= 127): maps are byte arrays handled by
//     lane e and lane e + 64 (template parameter MAXD); the per-candidate resolvers keep D <= 31,
//     longer keywords always come here.
//   * (round 3) most tiles are never mapped: the SPARSE SWEEP of pass 1 maps a batch's tiles from its
//     end backwards only while the batch's exit phase or the entry phase of a tile with something
//     to report is still open -- both are settled as soon as the map of a few tiles is constant --
//     and finds the tiles with something to report through the streaming filter's SWAR test
//     (mm_fwd_loud_mask in the kernel for 8-bit elements, a bitmap from the filter itself for 16-bit
//     ones).  1 GiB: 0.87 -> 0.43 ms (plain keyword), 0.97 -> 0.28 ms (wildcard keyword).
//   * (round 6) LOUD batches -- most tiles have something to report: two- and three-symbol keywords, planted floods -- are
//     WALKED: the batch sweeps its last tiles for its exit phase only, takes its entry phase from the look-back and goes
//     through its tiles once, mm_fwd_emit handing the phase on (before: every tile mapped, then every tile walked -- jumps
//     and exit tables twice).  Two-symbol keywords have one phase: no maps, no look-back, the finds straight off the
//     flags.  Inside a tile: the positions whose first compare holds are resolved by all lanes at once after the table
//     pass (not one at a time inside it), the compare loop keeps four steps' LDS reads in flight, the one phase goes
//     through the 64 groups by super-group tables (23 - 31 dependent reads, not 64), short keywords walk a group on a bit
//     mask in registers.  Forced engine, 1 GiB: planted flood 5.30 -> 1.86 ms, `qz` 3.10 -> 0.71, `q*v` 3.35 -> 1.49
//     (profiles/r06_forward_engine.log; where a wave's cycles go: MM_FWD_PROFILE below).
// Batches are handed out through tickets in order (blocks of 4 per workgroup, its waves taking them
// one by one): a waiting wave only ever waits for batches that running waves own or will take next:
// no residency assumption, no deadlock.
#ifndef MM_FORWARD_H
#define MM_FORWARD_H


constexpr int MM_FWD_BATCH = 16;               // tiles per batch (one wave) at most: 32 Ki positions (MmForwardArgs::batch: 16, or 4 for small inputs)
// positions per tile: a tile's positions sit in up to 3 slots further on (MmFwdLds), and 64 lanes own
// 64 groups of 32 slots -- 2044 + 3 slots still fit them
constexpr int MM_FWD_TILE = MM_TILE - 4;
constexpr unsigned long long MM_FWD_AGGREGATE = 1, MM_FWD_INCLUSIVE = 2;

// Dev builds (EXTRA=-DMM_FWD_PROFILE tools/build_variant.sh prof): where a wave's cycles go, phase by phase -- wave 0 of a
// few workgroups prints its sums when it is done.  MM_PROF(k): the cycles since the last mark belong to phase k.
#ifdef MM_FWD_PROFILE
__shared__ unsigned long long mm_prof[MM_WAVES][16];
__shared__ unsigned long long mm_prof_t[MM_WAVES];
#define MM_PROF(k)                                                                          \
   do {                                                                                     \
      if ((threadIdx.x & 63) == 0) {                                                        \
         const unsigned long long now_ = __builtin_readcyclecounter();                      \
         mm_prof[threadIdx.x >> 6][k] += now_ - mm_prof_t[threadIdx.x >> 6];                \
         mm_prof_t[threadIdx.x >> 6] = now_;                                                \
      }                                                                                     \
   } while (0)
#else
#define MM_PROF(k) do { } while (0)
#endif

// a wave's working set: one tile of MM_TILE positions of ELEM-byte elements
template <int ELEM>
struct MmFwdLds {
   static constexpr int kPositions = MM_TILE;
   static constexpr int kElem = ELEM;
   uint32_t tile_pad[1];                     // tile[-1]: mm_fwd_jumps reads the dword in front of any tile dword unconditionally
   // Slots: position p of the tile is slot q = p + r (r = 0..3: the jumps of four positions are stored as one
   // dword, see mm_fwd_jumps); groups of 32 slots are 36 bytes apart (MM_FWD_AT) so that 64 lanes working
   // on 64 groups in step hit different LDS banks (32 bytes apart: 16 of them on one bank).
   static constexpr int kPadded = (MM_TILE / 32) * 36 + 4;
   static_assert(MM_FWD_TILE + 3 <= MM_TILE, "a tile's slots must fit 64 groups of 32");
   static constexpr int kStaged = (MM_TILE + MMH_MAX_KEYWORD + 1) * ELEM + 16;
   uint32_t tile[((kStaged > kPadded ? kStaged : kPadded) + 3) / 4];   // staged bytes; later the exit tables / the finds
   uint8_t jump[kPadded];                    // J of every slot (| MM_JUMP_MATCH)
   uint8_t gentry[MM_TILE / 32 + 1];         // emit: how far behind its start the chain enters each group of 32 slots
};

// byte address of slot q in a padded array
#define MM_FWD_AT(q) ((q) + 4 * ((q) >> 5))

struct MmForwardArgs {
   MmTileArgs t;
   uint64_t ndom;            // domains worked on: nblocks * S in engine mode, 1 in whole-buffer mode, or the length of dom_list
   const uint32_t *dom_list; // nullptr: every domain; else the domains to work on
   uint32_t tpd;             // tiles per domain
   uint32_t bpd;             // batches per domain
   uint32_t batch;           // tiles per batch (<= MM_FWD_BATCH)
   uint8_t *agg;             // [ndom * bpd][MAXD] published batch maps (MAXD = 32, or 128 for keywords beyond 32 symbols)
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
   // the pre-pass's tile bitmap (launch_loud: the streaming filter over the ROM before this kernel): bit dom * tpd + t set when
   // a position of tile t of domain dom may pass the compare loop; null: no pre-pass (listed domains, keywords without a SWAR test)
   const uint32_t *loud;
   // 8-bit elements find the loud tiles of a batch themselves (mm_fwd_loud_mask: the kernel waits on LDS most of the time, the
   // memory system is idle): the streaming filter's first two SWAR conditions (FilterChoice, mm_kernels.h); 0: no such test
   uint32_t loud_shape;      // MM_F8_* shape with run-time shifts: 0x100 | MASK2 << 4 | number of conditions (1 or 2)
   uint32_t loud_iA;         // keyword position of condition 0
   uint32_t loud_pat[2];     // expected deltas, replicated over the bytes of a dword
   uint32_t loud_sh1;        // v_alignbit amount of condition 1
   uint32_t chunk;           // consecutive batches per ticket (a workgroup's block)
   // 16-bit elements, plain keywords (round 4, mm_fwd_exceptional16): a position jumps the default L - 1 -- and so keeps
   // the chain in its phase -- unless the delta of its first compare is one of the <= L listed ones; a tile without such a
   // position maps every phase onto itself and reports nothing, and is never staged.  quiet16: the test applies;
   // q16_bloom: three 32-bit filters on bits 0-4, 5-9 and 10-14 of the delta (a listed delta passes all three).
   uint32_t quiet16;
   uint32_t q16_bloom[3];
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
// downwards (everything above already holds): the jump, | MM_JUMP_MATCH when the loop reports a match.
// A step is three dependent LDS reads (its bridge, then the two elements) and a planted match runs L of them -- one after
// the other they were over half of a flooded tile's time (round 6, MM_FWD_PROFILE).  While three or more steps remain the
// reads of four steps are in flight together and the steps are looked at in the reference's order; the last one or two
// go one by one (keywords of three or four symbols have no more than that: reads for steps that do not exist cost
// padding floods, whose every position comes here, a fifth more time).
__device__ __forceinline__ int mm_deep_jump(const MmTileArgs &a, const MmPlanLds &P, const uint8_t *tile, int q, int from)
{
   const int S = (int)a.g.S;
   const bool be = a.g.big_endian != 0;
   int i = from;
   for (; i >= 2; i -= 4) {
      int at[4], br[4], di[4];
      uint32_t bad[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
         at[k] = i - k > 0 ? i - k : 0;                          // (steps below 0: step 0's reads again, not looked at)
         br[k] = P.bridge[at[k]];
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
         const int ci = mm_tile_elem(tile, q + at[k], S, be);
         const int pi = mm_tile_elem(tile, q + at[k] + br[k], S, be);
         di[k] = ci - pi;
         bad[k] = (uint32_t)(di[k] ^ P.expected[at[k]]) & P.cmp_mask[at[k]];
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
         if (i - k >= 0 && bad[k] != 0) {
            const int s = mm_tile_skip(a, P, di[k]);
            const int w = P.wst[at[k]];
            const int J = s < w ? s : w;
            return J < 1 ? 1 : J;
         }
      }
   }
   for (; i >= 0; --i) {
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
// position in W.jump: position p in slot q = p + *shift (padded layout, MM_FWD_AT).  *tile_out: the
// staged tile's first byte; *any_match: some position passed the whole compare loop.
template <class WL>
__device__ __forceinline__ void mm_fwd_jumps(const MmForwardArgs &a, const MmPlanLds &P, const MmFwdTables &T, WL &W, uint64_t start,
                                             int64_t lo, int npos, int lane, bool *any_match, const uint8_t **tile_out, int *shift)
{
   if (WL::kElem != 1 || !a.fast) {                  // (the table path is for 8-bit elements: the 16-bit kernels do not carry it)
      const uint8_t *tile = mm_tile_jumps(a.t, P, W, start, lo, npos, lane, true);
      bool m = false;
      for (int q = lane; q < npos; q += 64) {
         m = m || (W.jump[MM_FWD_AT(q)] & MM_JUMP_MATCH) != 0;
      }
      *any_match = __ballot(m) != 0;
      *tile_out = tile;
      *shift = 0;
      return;
   }
   MM_PROF(0);
   const int mis = mm_stage_tile(a.t, W, start, lo, npos, lane);
   mm_wave_sync();
   MM_PROF(2);
   const uint8_t *tile = reinterpret_cast<const uint8_t *>(W.tile) + mis;
   // position p compares LDS byte p + base with the byte g1 in front of it; LDS dword b4 + u holds
   // the compared bytes of slots 4u .. 4u + 3 (positions 4u - r + k)
   const int base = mis + (int)a.i1;
   const int r = base & 3, b4 = base >> 2;
   const int ndw = (npos + r + 3) >> 2;
   uint32_t *jump32 = reinterpret_cast<uint32_t *>(W.jump);
   const uint32_t sh1 = 32u - 8u * a.g1;
   // Pass 1: the table jump of every position; the positions whose first compare holds (1/256 of random ones) are only
   // noted, a bit per position in `hits` (a lane owns at most 8 dwords = 32 positions of a tile).  Pass 2: every lane works
   // off its own hits -- all lanes at once.  (Until round 6 a hit was resolved inside pass 1, the other 63 lanes waiting:
   // two thirds of the steps have a hit somewhere, and its second lookup and compare loop -- LDS round trips, one after the
   // other -- were paid eight times a tile instead of once or twice.)
   static_assert((MM_FWD_TILE + 3 + 3) / 4 <= 8 * 64, "a lane's hits of one tile fit 32 bits");
   uint32_t hits = 0;
   int round = 0;
   for (int u = lane; u < ndw; u += 64, round++) {
      const int mdw = b4 + u;
      const uint32_t w = W.tile[mdw];
      const uint32_t wp = W.tile[mdw - 1];                     // (mdw = 0: the pad dword in front of the tile; only positions < 0 use it)
      const uint32_t wq = mm_alignbit(w, wp, sh1);            // byte k: the partner of w's byte k
      uint32_t jj = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
         const int d = (int)((w >> (8 * k)) & 0xFF) - (int)((wq >> (8 * k)) & 0xFF);
         jj |= (uint32_t)T.jump1[d + 255] << (8 * k);
      }
      // Dword 0 may start up to three positions in front of the tile and the last one may end behind
      // it: their jumps are computed like the others (from whatever bytes are there) and never read.
      const uint32_t on = (jj >> 7) & 0x01010101u;            // first compare holds
      hits |= ((on * 0x10204080u) >> 28) << (4 * round);      // (bit 8 k -> bit 28 + k: no two products meet)
      jump32[u + (u >> 3)] = jj;                               // 8 dwords = one group of 32 slots, groups 9 dwords apart
   }
   bool m = false;
   // Dwords ALL of whose four positions hit -- padding under a keyword of equal symbols, a ramp under `abcd`: every
   // position, 32 a lane, each of them a handful of LDS round trips one after the other -- take the rest of the compare
   // loop four positions at a time: per step the four compared bytes and their four partners as two (unaligned) dwords,
   // the byte-wise difference against the step's expected one -- modular where the plan says so, exact (difference AND
   // borrow: c - p == e  <=>  (c - p) mod 256 == e mod 256 and (c < p) == (e < 0)) where it says so.  Four matches: the
   // dword's jumps are the match jump, done; anything else: the four go the way of all hits below.
   if (__ballot((hits & (hits >> 1) & (hits >> 2) & (hits >> 3) & 0x11111111u) != 0) != 0) {
      const uint32_t *t32 = W.tile;
      for (int round = 0; 64 * round < ndw; round++) {
         const int u = lane + 64 * round;
         const int p0 = 4 * u - r;
         bool dense = ((hits >> (4 * round)) & 15u) == 15u && p0 >= 0 && p0 + 3 < npos;
         if (__ballot(dense) == 0) {
            continue;
         }
         uint32_t holds = dense ? 0x80808080u : 0u;
         for (int i = (int)a.i1 - 1; i >= 0 && __ballot(holds != 0) != 0; --i) {
            const uint32_t mask = P.cmp_mask[i];
            if (mask == 0) {
               continue;                                       // (a wildcard's place: nothing is compared)
            }
            const int e = P.expected[i];
            const int at = mis + (dense ? p0 : 0) + i, pat = at + P.bridge[i];
            const uint32_t c = __builtin_amdgcn_alignbyte(t32[(at >> 2) + 1], t32[at >> 2], (uint32_t)at & 3u);
            const uint32_t q = __builtin_amdgcn_alignbyte(t32[(pat >> 2) + 1], t32[pat >> 2], (uint32_t)pat & 3u);
            const uint32_t x = mm_bytesub(c, q);
            const uint32_t ne = x ^ (((uint32_t)e & 0xFFu) * 0x01010101u);
            uint32_t ok = ~(((ne & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | ne) & 0x80808080u;      // bytes of ne that are zero, exactly
            if (mask == 0xFFFFFFFFu) {
               const uint32_t lt = ((~c & q) | (~(c ^ q) & x)) & 0x80808080u;            // borrow of c - q, byte by byte
               ok &= e >= 0 ? ~lt : lt;
            }
            holds &= ok;
         }
         if (holds == 0x80808080u) {
            jump32[u + (u >> 3)] = (a.t.plan.match_jump | (uint32_t)MM_JUMP_MATCH) * 0x01010101u;
            hits &= ~(15u << (4 * round));
            m = true;
         }
      }
   }
   if (__ballot(hits != 0) != 0) {
      while (hits) {
         const int bit = __ffs((int)hits) - 1;
         hits &= hits - 1;
         const int u = lane + 64 * (bit >> 2), k = bit & 3;
         const int q = 4 * u - r + k;
         int J = 1;
         if (q >= 0 && q < npos) {
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
         }
         m = m || (J & MM_JUMP_MATCH) != 0;
         W.jump[4 * (u + (u >> 3)) + k] = (uint8_t)J;          // (the lane's own dword, stored above)
      }
   }
   *any_match = __ballot(m) != 0;
   *tile_out = tile;
   *shift = r;
   mm_wave_sync();
   MM_PROF(3);
}

// x mod D for x < 2^16, any D <= 127 (mm_modd's 16-bit reciprocal is only exact for small D)
__device__ __forceinline__ uint32_t mm_fwd_modd(const MmTileArgs &a, uint32_t x)
{
   const uint32_t D = a.plan.L - 1;
   if (D == 1) {
      return 0;
   }
   const uint32_t q = __umulhi(x, a.inv_d32);                  // floor(x / D) or one more
   const int32_t rem = (int32_t)(x - q * D);
   return (uint32_t)(rem < 0 ? rem + (int32_t)D : rem);
}

// x / D for x < 2^16, any D <= 127
__device__ __forceinline__ uint32_t mm_fwd_divd(const MmTileArgs &a, uint32_t x)
{
   const uint32_t D = a.plan.L - 1;
   if (D == 1) {
      return x;                                                 // (2^32 / 1 + 1 does not fit inv_d32)
   }
   const uint32_t q = __umulhi(x, a.inv_d32);                  // floor(x / D) or one more
   return q * D > x ? q - 1 : q;
}

// Exit tables of slots [0, nq) from their jumps: the slots are cut into groups of 32, lane g owns
// group g and fills X[q] = how far behind the group's end the chain that visits q leaves it, from
// the group's last slot down: X[q] = q + J - end if that is >= 0, else X[q + J].  32 steps of one
// LDS read each whatever the pattern -- walking every phase's chain forward costs 64 / (mean jump)
// steps for each of D phases: fine for plain keywords, 4x more for wildcard patterns whose capped
// skips make the mean jump 1-2.  X overlays the staged bytes (padded like the jumps), which nobody
// needs once the jumps are known.  The group's 32 jumps are read as 8 dwords up front.
template <class WL>
__device__ __forceinline__ uint8_t *mm_fwd_exit_tables(WL &W, int nq, int lane)
{
   uint8_t *X = reinterpret_cast<uint8_t *>(W.tile);
   // (lanes behind the last group, and slots behind nq in the last one, work on whatever is there:
   // what they write is inside the buffer and never read by a real slot)
   const int len = nq - 32 * lane < 32 ? nq - 32 * lane : 32;    // <= 0 behind the last group
   const uint32_t *jw = reinterpret_cast<const uint32_t *>(W.jump) + 9 * lane;
   uint32_t jd[8];
#pragma unroll
   for (int i = 0; i < 8; i++) {
      jd[i] = jw[i];
   }
   uint8_t *Xg = X + 36 * lane;
#pragma unroll
   for (int k = 31; k >= 0; --k) {
      const int t = k + (int)((jd[k >> 2] >> (8 * (k & 3))) & (MM_JUMP_MATCH - 1));
      const int x = t >= len ? t - len : Xg[t];
      Xg[k] = (uint8_t)x;
   }
   mm_wave_sync();
   MM_PROF(4);
   return X;
}

// the chain that stands `off` slots behind the start of group g: where it stands on entering group g + 1
// (off >= the group's length: it jumps over the whole group -- only with D > 32 or in a short last group)
__device__ __forceinline__ uint32_t mm_fwd_through_group(const uint8_t *X, int nq, int g, uint32_t off)
{
   const uint32_t len = (uint32_t)(nq - 32 * g < 32 ? nq - 32 * g : 32);
   return off < len ? X[36 * g + off] : off - len;
}

// Phase map of positions [0, npos) (domain positions lo ..., lo_mod = lo mod D): v[h] = exit phase of
// entry phase lane + 64 h (identity for lanes that are no phase).  Threading a phase through the 64
// groups one after the other would be 64 dependent LDS reads; instead task (s, e) threads entry
// offset e through the 8 groups of super-group s (all of them at once, SX[s][e]), then phase lane
// goes through the 8 super-groups: 16 - 24 dependent reads for keywords of up to 17 symbols.
template <int NH, int MAXD, class WL>
__device__ __forceinline__ void mm_fwd_map(const MmTileArgs &a, WL &W, uint8_t (&SX)[MM_TILE / 256 + 1][MAXD + 4], int shift, int npos,
                                           uint32_t lo_mod, int lane, uint32_t (&v)[NH])
{
   const uint32_t D = a.plan.L - 1;
   const int nq = npos + shift;
   const uint8_t *X = mm_fwd_exit_tables(W, nq, lane);
   const int ngroups = (nq + 31) >> 5;
   const int nsuper = (ngroups + 7) >> 3;
   // SX[s][e]: the chain that stands e slots behind the start of super-group s (8 groups) -> where it stands behind its end.
   // (Super-group 0 is entered up to `shift` slots later than slot 0: its table has D + 3 entries.)
   const uint32_t width = D + 3;
   const uint32_t ntasks = (uint32_t)nsuper * width;
   for (uint32_t task = (uint32_t)lane; task < ntasks; task += 64) {
      uint32_t sg = 0, off = task;
      while (off >= width) {                                     // task / width (nsuper <= 9)
         off -= width;
         sg++;
      }
      const uint32_t e = off;
      const int g1 = 8 * (int)sg + 8 < ngroups ? 8 * (int)sg + 8 : ngroups;
      for (int g = 8 * (int)sg; g < g1; g++) {
         off = mm_fwd_through_group(X, nq, g, off);
      }
      SX[sg][e] = (uint8_t)off;                                  // e < D + 3 <= MAXD + 2
   }
   mm_wave_sync();
   const uint32_t end_mod = mm_fwd_modd(a, lo_mod + (uint32_t)npos);
#pragma unroll
   for (int h = 0; h < NH; h++) {
      const uint32_t e = (uint32_t)lane + 64u * h;
      v[h] = e;
      if (e < D) {
         // entry phase e enters at the first position >= 0 in that phase
         uint32_t off = e + D - lo_mod;
         off = (off >= D ? off - D : off) + (uint32_t)shift;
         for (int sg = 0; sg < nsuper; sg++) {
            // (a super-group of full groups is 256 slots long: an offset below D + 3 always lands inside
            // it; the last, possibly short one: the offset may fall behind it)
            const uint32_t len = (uint32_t)(nq - 256 * sg < 256 ? nq - 256 * sg : 256);
            off = off < len ? SX[sg][off] : off - len;
         }
         uint32_t ph = end_mod + off;                            // the chain left the window `off` positions behind its end
         v[h] = ph >= D ? ph - D : ph;
      }
   }
   mm_wave_sync();
   MM_PROF(5);
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

// mm_fwd_emit's super-group tables: keywords of up to MM_FWD_SXW - 2 symbols (D + 3 entry offsets per super-group)
constexpr int MM_FWD_SXW = 16;
struct MmFwdEmitLds {
   uint8_t sx[MM_TILE / 256][MM_FWD_SXW];   // [s][e]: the chain e slots behind the start of super-group s (8 groups) -> behind its end
   uint8_t sentry[MM_TILE / 256];           // how far behind the start of super-group s the true chain enters it
};

// the matches on the chain inside one tile, given the phase in which the chain enters it; returns the phase in which it
// leaves the tile (what the tile's map would say of `entry`)
template <class WL>
__device__ __forceinline__ uint32_t mm_fwd_emit(const MmForwardArgs &a, WL &W, MmFwdEmitLds &E, int shift, uint64_t start, int64_t lo, int npos,
                                                uint32_t entry, int lane)
{
   const uint32_t D = a.t.plan.L - 1;
   const uint32_t list = blockIdx.x & (MM_CAND_LISTS - 1);
   const uint32_t lo_mod = mm_modd64(a.t, (uint64_t)lo);
   const int nq = npos + shift;
   const int ngroups = (nq + 31) >> 5;
   const uint32_t first = 32u * (uint32_t)lane;
   const uint32_t end = first + 32 < (uint32_t)nq ? first + 32 : (uint32_t)nq;
   uint8_t *found = reinterpret_cast<uint8_t *>(W.tile);      // lane g's finds: found[32 g ...], offsets inside the group
   int nfound = 0;
   uint32_t beyond = 0;                                          // how far behind the tile's end the chain leaves it
   if (D == 1) {
      // Two-symbol keywords: every jump is 1, every position is on the chain -- no exit tables, no threading: the finds are
      // the slots that carry the flag, eight dwords of jumps per lane.  (Round 6: a two-symbol keyword floods any ROM, and
      // threading one phase through 64 groups was a third of its tiles' time.)
      mm_wave_sync();                                            // (the staged bytes are about to become `found`)
      const uint32_t *jw = reinterpret_cast<const uint32_t *>(W.jump) + 9 * lane;
#pragma unroll 1
      for (int i = 0; i < 8; i++) {
         uint32_t m = jw[i] & 0x80808080u;
         while (m) {
            const uint32_t q = first + 4u * i + ((uint32_t)(__ffs((int)m) - 1) >> 3);
            m &= m - 1;
            if (q >= (uint32_t)shift && q < end) {
               found[first + nfound++] = (uint8_t)(q - first);
            }
         }
      }
      MM_PROF(9);
   }
   else {
      const uint8_t *X = mm_fwd_exit_tables(W, nq, lane);
      uint32_t off0 = entry + D - lo_mod;                        // first position >= 0 in phase `entry` ...
      off0 = (off0 >= D ? off0 - D : off0) + (uint32_t)shift;    // ... as a slot
      const uint32_t width = D + 3;
      uint32_t q = end;                                          // where lane g's walk of group g starts
      if (width <= (uint32_t)MM_FWD_SXW) {
         // One phase through 64 groups is 64 dependent LDS reads on one lane (round 6: a fifth of a flooded tile's time).
         // As mm_fwd_map does it: every entry offset through the 8 groups of every super-group at once (8 x (D + 3) tasks),
         // lane 0 through the 8 super-groups, then lane g from its super-group's start to its group: 8 + 8 + 7 reads.
         const int nsuper = (ngroups + 7) >> 3;
         const uint32_t ntasks = (uint32_t)nsuper * width;
         for (uint32_t task = (uint32_t)lane; task < ntasks; task += 64) {
            uint32_t sg = 0, off = task;
            while (off >= width) {
               off -= width;
               sg++;
            }
            const uint32_t e = off;
            const int g1 = 8 * (int)sg + 8 < ngroups ? 8 * (int)sg + 8 : ngroups;
            for (int g = 8 * (int)sg; g < g1; g++) {
               off = mm_fwd_through_group(X, nq, g, off);
            }
            E.sx[sg][e] = (uint8_t)off;
         }
         mm_wave_sync();
         if (lane == 0) {
            uint32_t off = off0;
            for (int sg = 0; sg < nsuper; sg++) {
               E.sentry[sg] = (uint8_t)off;
               const uint32_t len = (uint32_t)(nq - 256 * sg < 256 ? nq - 256 * sg : 256);
               off = off < len ? E.sx[sg][off] : off - len;      // (off < D + 3: inside a full super-group)
            }
            beyond = off;
         }
         mm_wave_sync();
         if (lane < ngroups) {
            uint32_t off = E.sentry[lane >> 3];
            for (int g = lane & ~7; g < lane; g++) {
               off = mm_fwd_through_group(X, nq, g, off);
            }
            q = first + off;                                     // (off >= 32: the chain jumps over the group)
         }
      }
      else {
         if (lane == 0) {
            uint32_t off = off0;
            for (int g = 0; g < ngroups; g++) {
               W.gentry[g] = (uint8_t)(off < 255 ? off : 255);   // (>= 32: the chain jumps over the group)
               off = mm_fwd_through_group(X, nq, g, off);
            }
            beyond = off;
         }
         mm_wave_sync();
         q = first + W.gentry[lane < ngroups ? lane : 0];
      }
      beyond = (uint32_t)__builtin_amdgcn_readfirstlane((int)beyond);
      mm_wave_sync();
      MM_PROF(8);
      // lane g walks group g and notes the visited slots where the compare loop matched -- over the exit tables, which have
      // served their purpose
      if (D <= 8) {
         // Short keywords: a walk of 32 / (mean jump) dependent LDS reads -- 10 to 30 of them.  Instead the group's 32 jumps
         // from 8 dwords, and the set of visited slots as a bit mask, slot by slot in registers: slot k is visited when an
         // earlier visited slot jumps onto it (bit k + J <= 31 + 8).  No read depends on another, no lane on its data.
         const uint32_t *jw = reinterpret_cast<const uint32_t *>(W.jump) + 9 * lane;
         uint32_t jd[8];
#pragma unroll
         for (int i = 0; i < 8; i++) {
            jd[i] = jw[i];
         }
         unsigned long long visited = lane < ngroups && q - first < 32u ? 1ull << (q - first) : 0ull;
         uint32_t matching = 0;
#pragma unroll
         for (int k = 0; k < 32; k++) {
            const uint32_t j = (jd[k >> 2] >> (8 * (k & 3))) & 0xFFu;
            matching |= (j >> 7) << k;
            visited |= ((visited >> k) & 1ull) << (k + (j & 15u));   // (jumps <= D <= 8)
         }
         uint32_t mine = (uint32_t)visited & matching;
         mine &= end > first ? (end - first >= 32u ? 0xFFFFFFFFu : (1u << (end - first)) - 1u) : 0u;
         while (mine) {
            found[first + nfound++] = (uint8_t)(__ffs((int)mine) - 1);
            mine &= mine - 1;
         }
      }
      else if (lane < ngroups) {
         while (q < end) {
            const uint32_t j = W.jump[MM_FWD_AT(q)];
            if (j & MM_JUMP_MATCH) {
               found[first + nfound++] = (uint8_t)(q - first);
            }
            q += j & (MM_JUMP_MATCH - 1);
         }
      }
      MM_PROF(9);
   }
   // one atomic per tile reserves the output range; lanes copy their finds in group order
   int incl = nfound;
#pragma unroll
   for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      incl += lane >= d ? up : 0;
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
            const uint64_t j = (uint64_t)lo + first + found[first + k] - (uint32_t)shift;
            a.out[(uint64_t)list * a.list_cap + slot] = a.t.g.whole ? j : start + j * a.t.g.S + a.base_offset;
         }
      }
   }
   mm_wave_sync();
   MM_PROF(10);
   const uint32_t ph = mm_fwd_modd(a.t, lo_mod + (uint32_t)npos) + beyond;   // (as mm_fwd_map ends)
   return ph >= D ? ph - D : ph;
}

// The loud tiles among tiles [t0, t0 + ntiles) of the domain (positions [t0 TILE, lo1)): bit k set when a position of tile t0 + k
// passes the reference's whole compare loop.  The streaming filter's test (mm_f8_chunk: up to two SWAR conditions, a
// superset of the compare loop's first steps) on the positions' bytes straight from global memory, 4 KiB per wave and
// step -- 64 consecutive bytes per lane: four loads in flight per lane, and the dword in front of a chunk is the previous
// chunk's last one; what passes (2^-16 of random positions) runs the compare loop itself, unless its tile is known to be
// loud already (floods: every position passes).  8-bit elements only.  Wave uniform.
template <int SHAPE>
__device__ __forceinline__ uint32_t mm_fwd_loud_t(const MmForwardArgs &a, uint64_t start, int64_t lo0, int64_t lo1, int lane)
{
   const MmGeom &g = a.t.g;
   const uint64_t first = start + (uint64_t)lo0 + a.loud_iA;             // the anchor byte of position lo0 ...
   const uint64_t last = start + (uint64_t)(lo1 - 1) + a.loud_iA;        // ... and of the last position
   const uint32_t pat[4] = {a.loud_pat[0], a.loud_pat[1], 0u, 0u};
   const uint32_t sh[4] = {0u, a.loud_sh1, 0u, 0u};
   uint32_t mask = 0;
   for (uint64_t piece = first & ~(uint64_t)15; piece <= last; piece += 4096) {
      const uint64_t byte0 = piece + 64u * (uint32_t)lane;
      uint4 w[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
         w[k] = mm_load_chunk(g.rom, g.nbytes, byte0 + 16u * k);
      }
      const uint32_t back0 = byte0 >= 4 && byte0 <= g.nbytes ? *reinterpret_cast<const uint32_t *>(g.rom + byte0 - 4) : 0u;
      uint32_t some = 0;
      {
         uint32_t back = back0;
#pragma unroll
         for (int k = 0; k < 4; k++) {
            uint32_t h[4];
            some |= mm_f8_chunk<SHAPE>(w[k], back, pat, sh, h);
            back = w[k].w;
         }
      }
      const unsigned long long crowd = __ballot(some != 0);
      if (__popcll(crowd) > 4) {
         // hits all over the step (a flood: padding, low-entropy data): its tiles are loud, no questions asked -- loud may
         // say so of a tile too many (pass 2 then walks a tile that reports nothing), and running the compare loop from
         // global memory for every lane of every step cost floods a quarter more time than the whole engine without the sweep
         const uint64_t b0 = piece > first ? piece : first, b1 = piece + 4095 < last ? piece + 4095 : last;
         const uint32_t k0 = (uint32_t)(b0 - first) / MM_FWD_TILE, k1 = (uint32_t)(b1 - first) / MM_FWD_TILE;
         mask |= (2u << k1) - (1u << k0);
         if (mask == (2u << ((uint32_t)(last - first) / MM_FWD_TILE)) - 1u) {
            break;                                                         // every tile is loud already
         }
      }
      else if (crowd != 0) {
         uint32_t mine = 0;
         if (some) {
            // (rare: the flags again, chunk by chunk, from bytes loaded again: keeping the four chunks alive through
            // this branch costs the kernel a wave per SIMD)
            uint32_t back = back0;
#pragma unroll 1
            for (int k = 0; k < 4; k++) {
               const uint4 wk = mm_load_chunk(g.rom, g.nbytes, byte0 + 16u * (uint32_t)k);
               uint32_t h[4];
               mm_f8_chunk<SHAPE>(wk, back, pat, sh, h);
               back = wk.w;
               uint32_t bits = mm_f8_pack(h);                            // bit 8 b + d: byte b of dword d
               while (bits) {
                  const int bit = __ffs((int)bits) - 1;
                  bits &= bits - 1;
                  const uint64_t t = byte0 + (uint64_t)(16 * k + 4 * (bit & 7) + (bit >> 3));
                  if (t >= first && t <= last) {
                     const uint32_t tbit = 1u << ((uint32_t)(t - first) / MM_FWD_TILE);
                     if (((mask | mine) & tbit) == 0) {
                        bool matched = false;
                        mm_step(a.t.plan, [&](int64_t e) { return mm_elem(g, start, e); }, (int64_t)(t - a.loud_iA - start), &matched);
                        mine |= matched ? tbit : 0u;
                     }
                  }
               }
            }
         }
         if (__ballot(mine != 0) != 0) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
               mine |= (uint32_t)__shfl_xor((int)mine, d);
            }
            mask |= (uint32_t)__builtin_amdgcn_readfirstlane((int)mine);
         }
      }
   }
   return mask;
}

__device__ __forceinline__ uint32_t mm_fwd_loud_mask(const MmForwardArgs &a, uint64_t start, int64_t lo0, int64_t lo1, int lane)
{
   switch (a.loud_shape) {                                                // (wave uniform)
   case 0x101: return mm_fwd_loud_t<0x101>(a, start, lo0, lo1, lane);
   case 0x111: return mm_fwd_loud_t<0x111>(a, start, lo0, lo1, lane);
   case 0x102: return mm_fwd_loud_t<0x102>(a, start, lo0, lo1, lane);
   case 0x112: return mm_fwd_loud_t<0x112>(a, start, lo0, lo1, lane);
   case 0x122: return mm_fwd_loud_t<0x122>(a, start, lo0, lo1, lane);
   default: return mm_fwd_loud_t<0x132>(a, start, lo0, lo1, lane);
   }
}

// 16-bit elements, plain keywords.  A 16-bit delta is practically never in the skip table (<= L entries among 131071
// values), so nearly every position jumps the default L - 1: its chain stays in its phase, a tile's map is the identity and
// there is nothing for the sparse sweep to settle (maps never turn constant) -- round 3 mapped every tile on the general
// jump path, 3.9 ms per GiB.  But an identity map needs no mapping: the tiles among [t0, ...) that hold an EXCEPTIONAL
// position -- one whose first compare's delta x[h + i1] - x[h + i1 - 1] is listed (a shorter jump, or the compare holds
// and the loop goes on) -- are found from the bytes alone, 4 KiB per wave and step straight from global memory: both
// elements of a dword against three 32-bit filters (11 VALU operations per element), what passes (1.6 % with 8 listed
// deltas) against the list itself from bytes loaded again.  Returns bit k set when tile t0 + k holds such a position; the
// others are quiet: identity map, no report, never staged.  Positions [lo0, lo1) of the domain at `start`.  Wave uniform.
// Events (ev_pos != nullptr): every exceptional position in address order -- its offset from lo0 and its true jump from the
// reference's compare loop (| MM_JUMP_MATCH when the loop reports a match) --, at most `cap` of them; *overflow: more than
// that, or a flood (the caller then maps the exceptional tiles the general way).
__device__ __forceinline__ uint32_t mm_fwd_exceptional16(const MmForwardArgs &a, const MmPlanLds &P, uint64_t start, int64_t lo0, int64_t lo1,
                                                          int lane, uint32_t *ev_pos = nullptr, uint8_t *ev_jump = nullptr, uint32_t cap = 0,
                                                          uint32_t *n_events = nullptr, bool *overflow = nullptr)
{
   uint32_t nev = 0;
   bool over = ev_pos == nullptr;
   const MmGeom &g = a.t.g;
   const bool be = g.big_endian != 0;
   const int64_t i1 = (int64_t)a.i1;
   const uint32_t odd = (uint32_t)(start & 1);
   const uint64_t first = start + 2 * (uint64_t)(lo0 + i1);               // the compared element of position lo0 ...
   const uint64_t last = start + 2 * (uint64_t)(lo1 - 1 + i1);            // ... and of the last position
   const uint32_t b0 = a.q16_bloom[0], b1 = a.q16_bloom[1], b2 = a.q16_bloom[2];
   const uint32_t all = (uint32_t)(((lo1 - 1 - lo0) / MM_FWD_TILE) + 1);   // tiles asked about
   const uint32_t full = all >= 32 ? 0xFFFFFFFFu : (1u << all) - 1u;
   uint32_t mask = 0;
   // a lane's 64 bytes at byte0 hold the 32 elements at byte0 - odd + 2 k
   for (uint64_t piece = (first + odd) & ~(uint64_t)4095; piece <= last + odd && (mask != full || !over); piece += 4096) {
      const uint64_t byte0 = piece + 64u * (uint32_t)lane;
      uint32_t prev = byte0 >= 4 && byte0 <= g.nbytes ? *reinterpret_cast<const uint32_t *>(g.rom + byte0 - 4) : 0u;
      // the element in front of the lane's first one: bytes byte0 - 2, byte0 - 1 (even) / byte0 - 3, byte0 - 2 (odd)
      uint32_t pe = odd ? (prev >> 8) & 0xFFFFu : prev >> 16;
      pe = be ? ((pe >> 8) | (pe << 8)) & 0xFFFFu : pe;
      uint32_t pass = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
         const uint4 w4 = mm_load_chunk(g.rom, g.nbytes, byte0 + 16u * (uint32_t)q);
         const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
         for (int j = 0; j < 4; j++) {
            uint32_t pair = odd ? mm_alignbit(w[j], prev, 24) : w[j];       // two elements: low half first
            pair = be ? mm_bswap16x2(pair) : pair;
            prev = w[j];
            const uint32_t lo = pair & 0xFFFFu, hi = pair >> 16;
            const uint32_t d0 = lo - pe, d1 = hi - lo;                      // (two's complement: the filters were built that way)
            pe = hi;
            const uint32_t t0 = (b0 >> (d0 & 31)) & (b1 >> ((d0 >> 5) & 31)) & (b2 >> ((d0 >> 10) & 31)) & 1u;
            const uint32_t t1 = (b0 >> (d1 & 31)) & (b1 >> ((d1 >> 5) & 31)) & (b2 >> ((d1 >> 10) & 31)) & 1u;
            pass |= (t0 | (t1 << 1)) << (2 * (4 * q + j));
         }
      }
      const unsigned long long crowd = __ballot(pass != 0);
      if (crowd == 0) {
         continue;
      }
      const uint64_t e0 = piece > first + odd ? piece - odd : first, e1 = piece + 4095 - odd < last ? piece + 4095 - odd : last;
      if (__popcll(crowd) > 40) {
         // hits all over the step (low-entropy data, a flood): its tiles are exceptional, no questions asked
         const uint32_t k0 = (uint32_t)((e0 - first) / 2) / MM_FWD_TILE, k1 = (uint32_t)((e1 - first) / 2) / MM_FWD_TILE;
         mask |= ((k1 >= 31 ? 0u : (2u << k1)) - (1u << k0)) & full;
         over = true;
         continue;
      }
      // what passed the filters against the list itself (rare: from bytes loaded again -- indexing the 16 dwords above
      // with a run-time number would move them to scratch)
      uint32_t mine = 0, conf = 0;
      const int n = (int)a.t.plan.n_skip;
      while (pass) {
         const int bit = __ffs((int)pass) - 1;
         pass &= pass - 1;
         const uint64_t u = byte0 - odd + 2u * (uint32_t)bit;              // the element's first byte
         if (u < first || u > last) {
            continue;
         }
         const int64_t m = (int64_t)((u - start) >> 1);                    // its number in the domain
         const uint32_t tbit = 1u << ((uint32_t)(m - i1 - lo0) / MM_FWD_TILE);
         if (over && ((mask | mine) & tbit)) {
            continue;                                                      // (only the tile is asked for, and it is known)
         }
         const int d = mm_elem(g, start, m) - mm_elem(g, start, m - 1);
         bool listed = d == P.expected[i1];
         for (int k = 0; k < n; k++) {
            listed = listed || P.skip_diff[k] == d;
         }
         mine |= listed ? tbit : 0u;
         conf |= listed ? 1u << bit : 0u;
      }
      if (__ballot(mine != 0) != 0) {
         if (!over) {
            // the confirmed positions as events, in address order: lanes in order, a lane's own in order
            uint32_t total;
            uint32_t at = mm_wave_prefix((uint32_t)__popc(conf), &total);
            if (nev + total > cap) {
               over = true;
            }
            else {
               at += nev;
               while (conf) {
                  const int bit = __ffs((int)conf) - 1;
                  conf &= conf - 1;
                  const int64_t h = (int64_t)((byte0 - odd + 2u * (uint32_t)bit - start) >> 1) - i1;
                  bool matched = false;
                  const int J = mm_step(a.t.plan, [&](int64_t e) { return mm_elem(g, start, e); }, h, &matched);
                  ev_pos[at] = (uint32_t)(h - lo0);
                  ev_jump[at] = (uint8_t)(J | (matched ? MM_JUMP_MATCH : 0));
                  at++;
               }
               nev += total;
            }
         }
#pragma unroll
         for (int d = 1; d < 64; d <<= 1) {
            mine |= (uint32_t)__shfl_xor((int)mine, d);
         }
         mask |= (uint32_t)__builtin_amdgcn_readfirstlane((int)mine);
      }
   }
   if (n_events) {
      *n_events = nev;
      *overflow = over;
   }
   return mask & full;
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

// Decoupled look-back from batch `item` (not the first of its domain): the phase in which the chain enters it.  f (LDS, MAXD
// bytes): f[e] = the phase at OUR entry when the chain enters batch k + 1 in phase e; composing it with a batch's published
// map is one lookup per phase.  Ends at a batch whose exit phase is known, or as soon as the composition is constant.
template <int NH, int MAXD>
__device__ __forceinline__ uint32_t mm_fwd_lookback(const MmForwardArgs &a, uint64_t item, int lane, uint8_t *f)
{
   const uint32_t D = a.t.plan.L - 1;
   uint32_t entry = 0;
#pragma unroll
   for (int h = 0; h < NH; h++) {
      if (lane + 64 * h < MAXD) {
         f[lane + 64 * h] = (uint8_t)(lane + 64 * h);
      }
   }
   mm_wave_sync();
   for (uint64_t k = item - 1;; k--) {
      const unsigned long long st = mm_fwd_wait(a.status, k, lane);
      if ((st & 3) == MM_FWD_INCLUSIVE) {
         entry = f[(st >> 8) & 0xFF];
         break;
      }
      uint32_t fn[NH];
#pragma unroll
      for (int h = 0; h < NH; h++) {
         fn[h] = 0;
         if ((uint32_t)lane + 64u * h < D) {
            const uint32_t mk = __hip_atomic_load(a.agg + k * MAXD + lane + 64 * h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fn[h] = f[mk];
         }
      }
      mm_wave_sync();
      const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)fn[0]);
      bool varies = false;
#pragma unroll
      for (int h = 0; h < NH; h++) {
         if ((uint32_t)lane + 64u * h < D) {
            f[lane + 64 * h] = (uint8_t)fn[h];
            varies = varies || fn[h] != f0;
         }
      }
      mm_wave_sync();
      if (__ballot(varies) == 0) {
         entry = f0;                          // every entry phase of batch k ends up here: no need to go further back
         break;
      }
   }
   return mm_uniform(entry);
}

// ELEM: element bytes the tile buffers are sized for (1: 8-bit searches, 6 workgroups per CU instead of 4);
// MAXD: 32 for keywords of up to 32 symbols, 128 beyond (phase maps handled by lane e and lane e + 64)
template <int ELEM, int MAXD>
__global__ __launch_bounds__(64 * MM_WAVES) __attribute__((amdgpu_waves_per_eu(ELEM == 1 && MAXD == 32 ? 6 : 4))) void mm_forward(MmForwardArgs a)
{
   using WL = MmFwdLds<ELEM>;
   constexpr int NH = (MAXD + 63) / 64;
   __shared__ MmPlanLds P;
   __shared__ MmFwdTables T;
   __shared__ WL Wv[MM_WAVES];
   __shared__ uint8_t tilemap[MM_WAVES][MM_FWD_BATCH][MAXD];
   __shared__ uint8_t lookback[MM_WAVES][MAXD];
   __shared__ uint8_t lookback2[MM_WAVES][MAXD];      // the sweep's second composition (lookback: its first)
   __shared__ uint8_t centry[MM_WAVES][MM_FWD_BATCH]; // ... and the entry phases it finds
   __shared__ MmFwdEmitLds emit_lds[MM_WAVES];
   // (the super-group tables of mm_fwd_map overlay the tile's jumps: those have served their purpose once the exit
   // tables exist, and the 1.3 KiB this saves is what keeps six workgroups on a CU)
   static_assert(sizeof(uint8_t[MM_TILE / 256 + 1][MAXD + 4]) <= sizeof(WL::jump), "super-group tables overlay the jumps");
   __shared__ unsigned int q_taken, q_ready[2], q_readers[2];
   __shared__ unsigned long long q_base[2];
   mm_plan_to_lds(P, a.t.plan);
   mm_fwd_tables(T, P, a);

   const uint32_t D = a.t.plan.L - 1;
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   WL &W = Wv[wave];
   const uint64_t nbatches = a.ndom * a.bpd;

   // Hand-out (round 3; before: one ticket per workgroup and four batches, its four waves in step -- with swept
   // batches a tenth of the batches takes five times as long as the rest, and a workgroup whose waves wait for each
   // other spends a third of its rounds at the slow one's pace).  Now the workgroup draws BLOCKS of a.chunk consecutive
   // batches from the global ticket and its waves take batches out of the current block one by one through a counter
   // in LDS, nobody waiting for anybody: taken = batches the workgroup's waves have taken so far; the wave that takes
   // the first batch of block j draws the block's ticket and announces it in slot j % 2 (base, ready = j + 1), the
   // takers of the block's other batches spin on that word; the slot is reused by block j + 2, whose drawer first
   // waits until all of block j's takers have read it (readers).  The four waves of a workgroup so work on
   // neighbouring batches, as they did with the old hand-out.  (One ticket per wave and batch: 32 K atomics per GiB on
   // one address, and the waves of a CU spread over the ROM: 0.93 instead of 0.72 ms per GiB.)
   // No deadlock: tickets are drawn in order and a workgroup's batches are taken in order by waves that only ever wait
   // for smaller batches, so the smallest unfinished batch is always in the hands of a wave that can run, or will be
   // taken by the next wave of its workgroup that finishes.
   if (threadIdx.x == 0) {
      q_taken = 0;
      q_ready[0] = q_ready[1] = 0;
      q_readers[0] = q_readers[1] = 0;
   }
#ifdef MM_FWD_PROFILE
   if (threadIdx.x < 16 * MM_WAVES) {
      mm_prof[threadIdx.x >> 4][threadIdx.x & 15] = 0;
   }
   if ((threadIdx.x & 63) == 0) {
      mm_prof_t[threadIdx.x >> 6] = __builtin_readcyclecounter();
   }
#endif
   __syncthreads();
   for (;;) {
      unsigned long long mine = 0;
      if (lane == 0) {
         const uint32_t i = atomicAdd(&q_taken, 1u);
         const uint32_t j = i / a.chunk, o = i - j * a.chunk, slot = j & 1u;
         if (o == 0) {
            if (j >= 2) {
               while (__hip_atomic_load(&q_readers[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != a.chunk) {
                  __builtin_amdgcn_s_sleep(1);
               }
               __hip_atomic_store(&q_readers[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            q_base[slot] = atomicAdd(a.ticket, (unsigned long long)a.chunk);
            __hip_atomic_store(&q_ready[slot], j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
         }
         else {
            while (__hip_atomic_load(&q_ready[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != j + 1) {
               __builtin_amdgcn_s_sleep(1);
            }
         }
         mine = q_base[slot] + o;
         __hip_atomic_fetch_add(&q_readers[slot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      const uint64_t item = mm_uniform64(mine);
      if (item >= nbatches) {
         break;
      }
      MM_PROF(11);
      const uint64_t dom = item / a.bpd;
      const uint32_t b = (uint32_t)(item % a.bpd);
      uint64_t start; int64_t nv;
      mm_fwd_domain(a, dom, &start, &nv);
      start = mm_uniform64(start);
      const uint32_t t0 = b * a.batch;
      const uint32_t t1 = t0 + a.batch < a.tpd ? t0 + a.batch : a.tpd;

      // ---- pass 1: tile maps.  Two ways through the batch's tiles, one loop (the mapping code exists once):
      //
      // SWEEP (with the pre-pass's bitmap).  Two observations make most of a batch's tiles unnecessary to map:
      //  * the batch's map is the map of its LAST tiles alone once that is constant: whatever phase the chain enters
      //    the batch in, it leaves those tiles -- and so the batch -- in the same one.  Likewise the phase in which
      //    the chain enters a tile f is known once the map of the tiles right in front of f is constant.  (Wildcard
      //    keywords: one tile is enough -- their capped skips mix the phases quickly; plain keywords: 1 - 4 -- most
      //    of their jumps are L - 1, which keeps a chain in its phase);
      //  * a tile none of whose positions passes the compare loop has nothing to report, and which tiles those are
      //    the streaming filter has found out at its own speed before this kernel started (a.loud).
      // So the tiles are swept from the batch's end backwards.  While something still asks for maps -- ex: the batch's
      // exit phase, tg: the entry phase of the lowest loud tile met so far -- tiles are mapped and composed into
      // ex / tg; once nothing does, the sweep jumps to the next loud tile down or ends.  The entry phases found
      // (centry) and the maps made (tilemap) are then all pass 2 needs: a tile's entry phase is either in centry or
      // follows from the tile below through that one's map.
      //
      // FILL.  A sweep that arrives at the batch's first tile still asking (floods, low-entropy data: the chains do not
      // merge) has mapped every tile from where the question came up; nothing is lost: the tiles without a map are
      // mapped as well, bottom up, the batch's map is composed from all of them and the look-back over the batches in
      // front does the asking (round 2's way, which scans without the bitmap take from the start).
      uint32_t flagged = 0;                         // tiles to report from: loud ones (sweep), or with a position that passed the compare loop
      uint32_t have = 0;                            // tiles whose maps sit in tilemap[wave]
      uint32_t known = 0;                           // tiles whose entry phase sits in centry[wave]
      const int tl = nv > (int64_t)t0 * MM_FWD_TILE                      // the batch's last tile inside the domain (t0 - 1: none)
                        ? (int)((uint64_t)(nv - 1) / MM_FWD_TILE < t1 - 1 ? (uint64_t)(nv - 1) / MM_FWD_TILE : t1 - 1)
                        : (int)t0 - 1;
      bool sweep = (a.loud != nullptr || (ELEM == 1 && a.loud_shape != 0)) && tl >= (int)t0;
      uint32_t loud = 0;
      MM_PROF(0);
      if (ELEM == 1 && sweep && a.loud == nullptr) {
         const int64_t lo1 = (int64_t)(tl + 1) * MM_FWD_TILE;
         loud = mm_fwd_loud_mask(a, start, (int64_t)t0 * MM_FWD_TILE, lo1 < nv ? lo1 : nv, lane);
      }
      else if (sweep) {
         const uint64_t bit0 = dom * a.tpd + t0;
         const uint64_t two = (uint64_t)a.loud[bit0 >> 5] | ((uint64_t)a.loud[(bit0 >> 5) + 1] << 32);
         loud = (uint32_t)(two >> (bit0 & 31)) & ((2u << (tl - (int)t0)) - 1u);
         loud = (uint32_t)__builtin_amdgcn_readfirstlane((int)loud);
      }
      if (ELEM == 2 && a.quiet16 && !sweep && tl >= (int)t0) {
         // Plain 16-bit keyword.  The batch's few exceptional positions (listed delta: another jump than the default, or
         // the compare loop goes on) come as EVENTS -- position, true jump, match flag -- straight from the bytes; every
         // other position keeps its chain in its phase.  So the batch's phase map is the identity with the events applied
         // in order (the chain standing in phase h mod D at event h moves to (h + J) mod D), the matches are the matching
         // events the true chain visits, and nothing is staged, no jump tables, no exit tables: 1.84 -> ~0.5 ms per GiB.
         const int64_t lo0 = (int64_t)t0 * MM_FWD_TILE;
         const int64_t lo1 = (int64_t)(tl + 1) * MM_FWD_TILE;
         constexpr uint32_t kEvents = 192;
         uint32_t *ev_pos = W.tile;
         uint8_t *ev_jump = reinterpret_cast<uint8_t *>(W.tile + kEvents);
         static_assert(sizeof(W.tile) >= kEvents * 5, "the events live in the wave's tile buffer");
         uint32_t n_ev = 0;
         bool too_many = false;
         const uint32_t exceptional = mm_fwd_exceptional16(a, P, start, lo0, lo1 < nv ? lo1 : nv, lane, ev_pos, ev_jump, kEvents, &n_ev,
                                                          &too_many);
         n_ev = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_ev);
         if (__ballot(too_many) == 0) {
            mm_wave_sync();
            const uint32_t lo0_mod = mm_modd64(a.t, (uint64_t)lo0);
            uint32_t bm[NH];
#pragma unroll
            for (int h = 0; h < NH; h++) {
               bm[h] = (uint32_t)lane + 64u * h;
            }
            bool reports = false;
            for (uint32_t k = 0; k < n_ev; k++) {
               const uint32_t pos = ev_pos[k], jj = ev_jump[k];
               const uint32_t r = mm_fwd_modd(a.t, lo0_mod + pos);
               uint32_t to = r + (jj & (MM_JUMP_MATCH - 1));
               to = to >= D ? to - D : to;
               reports = reports || (jj & MM_JUMP_MATCH) != 0;
#pragma unroll
               for (int h = 0; h < NH; h++) {
                  bm[h] = bm[h] == r ? to : bm[h];
               }
            }
            const uint32_t bm0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)bm[0]);
            bool differs = false;
#pragma unroll
            for (int h = 0; h < NH; h++) {
               differs = differs || ((uint32_t)lane + 64u * h < D && bm[h] != bm0);
            }
            const bool constant = __ballot(differs) == 0;
            if (b == 0 || constant) {
               if (lane == 0) {
                  __hip_atomic_store(a.status + item, MM_FWD_INCLUSIVE | ((unsigned long long)bm0 << 8), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
               }
            }
            else {
#pragma unroll
               for (int h = 0; h < NH; h++) {
                  if (lane + 64 * h < MAXD) {
                     __hip_atomic_store(a.agg + item * MAXD + lane + 64 * h, (uint8_t)bm[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  }
               }
               asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
               if (lane == 0) {
                  __hip_atomic_store(a.status + item, MM_FWD_AGGREGATE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               }
            }
            uint32_t entry = 0;
            if (b != 0 && (reports || !constant)) {
               entry = mm_fwd_lookback<NH, MAXD>(a, item, lane, lookback[wave]);
            }
            if ((b != 0 && !constant) || reports) {
               // the true chain through the events: where it leaves the batch, and the matches it meets
               uint32_t ph = entry;
               const uint32_t list = blockIdx.x & (MM_CAND_LISTS - 1);
               for (uint32_t k = 0; k < n_ev; k++) {
                  const uint32_t pos = ev_pos[k], jj = ev_jump[k];
                  const uint32_t r = mm_fwd_modd(a.t, lo0_mod + pos);
                  if (ph != r) {
                     continue;
                  }
                  if ((jj & MM_JUMP_MATCH) && lane == 0) {
                     const unsigned long long slot = atomicAdd(a.list_count + list * MM_LIST_STRIDE, 1ull);
                     if (slot < a.list_cap) {
                        const uint64_t j = (uint64_t)lo0 + pos;
                        a.out[(uint64_t)list * a.list_cap + slot] = a.t.g.whole ? j : start + j * a.t.g.S + a.base_offset;
                     }
                  }
                  const uint32_t to = r + (jj & (MM_JUMP_MATCH - 1));
                  ph = to >= D ? to - D : to;
               }
               if (b != 0 && !constant && lane == 0) {
                  __hip_atomic_store(a.status + item, MM_FWD_INCLUSIVE | ((unsigned long long)ph << 8), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
               }
            }
            mm_wave_sync();
            continue;                                // the batch is done
         }
         // (too many events, or a flood: the exceptional tiles the general way, the quiet ones still for free)
         mm_wave_sync();
         const uint32_t quiet = ~exceptional & ((2u << (tl - (int)t0)) - 1u);
         for (uint32_t rest = quiet; rest; rest &= rest - 1) {
            const int k = __ffs((int)rest) - 1;
#pragma unroll
            for (int h = 0; h < NH; h++) {
               if (lane + 64 * h < MAXD) {
                  tilemap[wave][k][lane + 64 * h] = (uint8_t)(lane + 64 * h);
               }
            }
         }
         have |= quiet;
         mm_wave_sync();
      }
      // WALK (round 6).  A batch most of whose tiles are loud (short keywords: two or three symbols match somewhere in every
      // tile; planted floods) gains nothing from the sweep's second question: every tile is the target in turn, every tile
      // gets mapped AND walked -- jumps and exit tables twice.  Such a batch only sweeps for its exit phase (the last one to
      // four tiles), takes its entry phase from the look-back -- the batches in front publish theirs as early -- and walks
      // its tiles bottom up with the phase in hand: mm_fwd_emit says where the chain leaves a tile.  No maps but the
      // sweep's few.  (Chains that do not merge -- padding under a keyword of equal symbols -- never settle the exit
      // phase: the sweep arrives at the first tile still asking and the batch is filled as before.)
      MM_PROF(1);
      const bool walk = sweep && loud != 0 && 2 * __popc(loud) >= 32 - __clz((int)loud);
      if (walk) {
         flagged = loud;
      }
      if (D == 1 && sweep) {
         // Two-symbol keywords: ONE phase.  The batch's exit phase and every tile's entry phase are known without a map or a
         // look-back; the loud tiles are walked, the others not even staged.
         if (lane == 0) {
            __hip_atomic_store(a.status + item, MM_FWD_INCLUSIVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         }
         for (uint32_t rest = loud; rest; rest &= rest - 1) {
            const int64_t lo = (int64_t)(t0 + (uint32_t)__ffs((int)rest) - 1u) * MM_FWD_TILE;
            const int npos = (int)(nv - lo < MM_FWD_TILE ? nv - lo : MM_FWD_TILE);
            bool any = false;
            const uint8_t *tile;
            int shift;
            mm_fwd_jumps(a, P, T, W, start, lo, npos, lane, &any, &tile, &shift);
            mm_fwd_emit(a, W, emit_lds[wave], shift, start, lo, npos, 0u, lane);
         }
         continue;
      }
      bool need_ex = sweep, need_tg = false;
      uint32_t target = 0;
      uint8_t *ex = lookback[wave], *tg = lookback2[wave];
      if (sweep) {
#pragma unroll
         for (int h = 0; h < NH; h++) {
            if (lane + 64 * h < MAXD) {
               ex[lane + 64 * h] = (uint8_t)(lane + 64 * h);
            }
         }
         mm_wave_sync();
      }
      for (int t = sweep ? tl : (int)t0;;) {
         bool mapped = false, any = false;
         if (sweep) {
            if (t < (int)t0) {
               if (!need_ex && !need_tg) {
                  break;                               // swept: every question answered
               }
               sweep = false;                          // still asking at the batch's first tile: fill
               flagged = 0;                            // (filled batches report from the tiles mm_fwd_jumps names)
               t = (int)t0;
               continue;
            }
            if (!need_ex && !need_tg) {
               // nothing asks for maps: on to the next loud tile down
               const uint32_t below = loud & ((2u << (t - (int)t0)) - 1u);
               if (below == 0 || walk) {
                  break;
               }
               t = (int)t0 + 31 - __clz((int)below);
            }
            else {
               mapped = true;
            }
            any = ((loud >> (t - (int)t0)) & 1u) != 0;
         }
         else {
            while (t <= tl && ((have >> (t - (int)t0)) & 1u) != 0) {
               // (a map the sweep made: only the tile's flag is missing)
               flagged |= loud & (1u << (t - (int)t0));
               t++;
            }
            if (t > tl) {
               break;
            }
            mapped = true;
         }
         if (mapped) {
            const int64_t lo = (int64_t)t * MM_FWD_TILE;
            const int npos = (int)(nv - lo < MM_FWD_TILE ? nv - lo : MM_FWD_TILE);
            uint32_t map[NH];
            const uint8_t *tile;
            int shift;
            bool passed = false;
            mm_fwd_jumps(a, P, T, W, start, lo, npos, lane, &passed, &tile, &shift);
            mm_fwd_map<NH, MAXD>(a.t, W, (*reinterpret_cast<uint8_t (*)[MM_TILE / 256 + 1][MAXD + 4]>(W.jump)), shift, npos,
                                 mm_modd64(a.t, (uint64_t)lo), lane, map);
            any = sweep ? any : passed;
            uint32_t en[NH], gn[NH];
#pragma unroll
            for (int h = 0; h < NH; h++) {
               en[h] = gn[h] = 0;
               if (lane + 64 * h < MAXD) {
                  tilemap[wave][t - (int)t0][lane + 64 * h] = (uint8_t)map[h];
               }
               if (sweep && (uint32_t)lane + 64u * h < D) {
                  en[h] = need_ex ? ex[map[h]] : 0u;
                  gn[h] = need_tg ? tg[map[h]] : 0u;
               }
            }
            have |= 1u << (t - (int)t0);
            mm_wave_sync();
            if (sweep) {
               const uint32_t e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)en[0]);
               const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)gn[0]);
               bool evaries = false, gvaries = false;
#pragma unroll
               for (int h = 0; h < NH; h++) {
                  if ((uint32_t)lane + 64u * h < D) {
                     if (need_ex) {
                        ex[lane + 64 * h] = (uint8_t)en[h];
                     }
                     if (need_tg) {
                        tg[lane + 64 * h] = (uint8_t)gn[h];
                     }
                     evaries = evaries || en[h] != e0;
                     gvaries = gvaries || gn[h] != g0;
                  }
               }
               mm_wave_sync();
               if (need_ex && __ballot(evaries) == 0) {
                  // the tiles from here to the batch's end take every phase to e0: tell the batches behind us at once
                  if (lane == 0) {
                     __hip_atomic_store(a.status + item, MM_FWD_INCLUSIVE | ((unsigned long long)e0 << 8), __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
                  }
                  need_ex = false;
               }
               if (need_tg && __ballot(gvaries) == 0) {
                  if (lane == 0) {
                     centry[wave][target - t0] = (uint8_t)g0;
                  }
                  known |= 1u << (target - t0);
                  need_tg = false;
               }
            }
         }
         if (any) {
            flagged |= 1u << (t - (int)t0);
            if (sweep && !walk) {
               // the lowest tile that reports so far: the ones above get their entry phases through the maps in between
               // (all made: the question never went away)
               target = (uint32_t)t;
               need_tg = true;
#pragma unroll
               for (int h = 0; h < NH; h++) {
                  if (lane + 64 * h < MAXD) {
                     tg[lane + 64 * h] = (uint8_t)(lane + 64 * h);
                  }
               }
               mm_wave_sync();
            }
         }
         t += sweep ? -1 : 1;
      }
      const bool swept = sweep;
      MM_PROF(6);
      uint32_t entry = 0;                           // first batch of a domain: the chain starts at its first position
      if (!swept) {
      // (tiles past the domain's end: identity)
      for (int t = tl + 1; t < (int)t1; t++) {
#pragma unroll
         for (int h = 0; h < NH; h++) {
            if (lane + 64 * h < MAXD) {
               tilemap[wave][t - (int)t0][lane + 64 * h] = (uint8_t)(lane + 64 * h);
            }
         }
      }
      mm_wave_sync();
      uint32_t bm[NH];                              // bm[h]: where entry phase lane + 64 h leaves the batch
#pragma unroll
      for (int h = 0; h < NH; h++) {
         bm[h] = (uint32_t)lane + 64u * h;
      }
      for (uint32_t t = t0; t < t1; t++) {
#pragma unroll
         for (int h = 0; h < NH; h++) {
            if ((uint32_t)lane + 64u * h < D) {
               bm[h] = tilemap[wave][t - t0][bm[h]];
            }
         }
      }
      // ---- publish; find the phase in which the chain enters the batch --------------------------
      const uint32_t bm0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)bm[0]);
      bool differs = false;
#pragma unroll
      for (int h = 0; h < NH; h++) {
         differs = differs || ((uint32_t)lane + 64u * h < D && bm[h] != bm0);
      }
      const bool constant = __ballot(differs) == 0;
      if (b == 0 || constant) {
         // (b == 0: the chain enters in phase 0, and lane 0 is the first lane)
         if (lane == 0) {
            __hip_atomic_store(a.status + item, MM_FWD_INCLUSIVE | ((unsigned long long)bm0 << 8), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
         }
      }
      else {
#pragma unroll
         for (int h = 0; h < NH; h++) {
            if (lane + 64 * h < MAXD) {
               __hip_atomic_store(a.agg + item * MAXD + lane + 64 * h, (uint8_t)bm[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
         }
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
         if (lane == 0) {
            __hip_atomic_store(a.status + item, MM_FWD_AGGREGATE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         }
      }
      MM_PROF(12);
      if (b != 0 && (flagged != 0 || !constant)) {
         // decoupled look-back: f[e] = the phase at OUR entry when the chain enters batch k+1 in phase e
         // (kept in LDS: composing it with a batch's map is one lookup per phase)
         uint8_t *f = lookback[wave];
#pragma unroll
         for (int h = 0; h < NH; h++) {
            if (lane + 64 * h < MAXD) {
               f[lane + 64 * h] = (uint8_t)(lane + 64 * h);
            }
         }
         mm_wave_sync();
         for (uint64_t k = item - 1;; k--) {
            const unsigned long long st = mm_fwd_wait(a.status, k, lane);
            if ((st & 3) == MM_FWD_INCLUSIVE) {
               entry = f[(st >> 8) & 0xFF];
               break;
            }
            uint32_t fn[NH];
#pragma unroll
            for (int h = 0; h < NH; h++) {
               fn[h] = 0;
               if ((uint32_t)lane + 64u * h < D) {
                  const uint32_t mk = __hip_atomic_load(a.agg + k * MAXD + lane + 64 * h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  fn[h] = f[mk];
               }
            }
            mm_wave_sync();
            const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)fn[0]);
            bool varies = false;
#pragma unroll
            for (int h = 0; h < NH; h++) {
               if ((uint32_t)lane + 64u * h < D) {
                  f[lane + 64 * h] = (uint8_t)fn[h];
                  varies = varies || fn[h] != f0;
               }
            }
            mm_wave_sync();
            if (__ballot(varies) == 0) {
               entry = f0;                          // every entry phase of batch k ends up here: no need to go further back
               break;
            }
         }
         entry = mm_uniform(entry);
      }
      MM_PROF(7);
      if (b != 0 && !constant) {
         // now that the entry is known, tell the batches behind us where the chain leaves this one
         uint32_t ph = entry;
         for (uint32_t t = t0; t < t1; t++) {
            ph = tilemap[wave][t - t0][ph];
         }
         if (lane == 0) {
            __hip_atomic_store(a.status + item, MM_FWD_INCLUSIVE | ((unsigned long long)ph << 8), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
         }
      }
      }
      if (swept && walk) {
         // ---- the walk: entry phase from the batches in front, then tile by tile up to the last loud one
         MM_PROF(6);
         uint32_t ph = b != 0 ? mm_fwd_lookback<NH, MAXD>(a, item, lane, lookback[wave]) : 0u;
         MM_PROF(7);
         const int last = (int)t0 + 31 - __clz((int)loud);
         for (int t = (int)t0; t <= last; t++) {
            const uint32_t bit = 1u << (t - (int)t0);
            if ((loud & bit) == 0 && (have & bit) != 0) {
               ph = tilemap[wave][t - (int)t0][ph];           // (a quiet tile the sweep mapped)
               continue;
            }
            const int64_t lo = (int64_t)t * MM_FWD_TILE;
            const int npos = (int)(nv - lo < MM_FWD_TILE ? nv - lo : MM_FWD_TILE);
            bool any = false;
            const uint8_t *tile;
            int shift;
            mm_fwd_jumps(a, P, T, W, start, lo, npos, lane, &any, &tile, &shift);
            ph = mm_fwd_emit(a, W, emit_lds[wave], shift, start, lo, npos, ph, lane);   // (a quiet tile: nothing to report, the phase moves on)
         }
         continue;
      }
      // ---- pass 2 (rare): the tiles to report from, walked with their true entry phase: the batch's entry phase
      // carried through every tile's map (filled batches), or what the sweep found (centry) carried on through the
      // maps it made
      if (flagged) {
         uint32_t ph = entry;
         bool ph_known = !swept;
         for (uint32_t t = t0; (int)t <= tl && (flagged >> (t - t0)) != 0; t++) {
            const uint32_t bit = 1u << (t - t0);
            if (known & bit) {
               ph = centry[wave][t - t0];
               ph_known = true;
            }
            if (flagged & bit) {
               const int64_t lo = (int64_t)t * MM_FWD_TILE;
               const int npos = (int)(nv - lo < MM_FWD_TILE ? nv - lo : MM_FWD_TILE);
               bool any = false;
               const uint8_t *tile;
               int shift;
               mm_fwd_jumps(a, P, T, W, start, lo, npos, lane, &any, &tile, &shift);
               mm_fwd_emit(a, W, emit_lds[wave], shift, start, lo, npos, ph, lane);
            }
            ph_known = ph_known && (have & bit) != 0;
            if (ph_known) {
               ph = tilemap[wave][t - t0][ph];
            }
         }
      }
   }
#ifdef MM_FWD_PROFILE
   if (threadIdx.x == 0 && blockIdx.x % 389 == 0) {
      MM_PROF(11);
      const unsigned long long *q = mm_prof[0];
      printf("mm_forward wg %4u wave 0, kilocycles: other %llu  loud %llu  stage %llu  jumps %llu  exit-tables %llu  map %llu  sweep %llu  look-back %llu  "
             "emit: thread %llu  walk %llu  output %llu | ticket %llu  fill/publish %llu\n", blockIdx.x, q[0] >> 10, q[1] >> 10, q[2] >> 10, q[3] >> 10,
             q[4] >> 10, q[5] >> 10, q[6] >> 10, q[7] >> 10, q[8] >> 10, q[9] >> 10, q[10] >> 10, q[11] >> 10, q[12] >> 10);
   }
#endif
}

#endif
