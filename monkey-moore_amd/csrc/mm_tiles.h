// SPDX-License-Identifier: GPL-3.0-or-later
// mm_tiles.h -- exact emulation of the reference's skip chain, tile by tile.
// Included by mm_kernels.hip (device code only).
//
// Chain state = the next position the reference will visit.  Every jump is in
// [1, D] with D = L-1, so at any boundary a the next visited position lies in
// [a, a+D) and is identified by its PHASE (position mod D).  The effect of a run
// of positions is a map Z_D -> Z_D (entry phase -> exit phase); maps of adjacent
// runs compose.
//
// One wavefront turns a window of up to MM_TILE positions into its map (mm_tile_map):
//   1. coalesced copy of the window's bytes into LDS, all loads in flight at once;
//   2. position-parallel: iteration t, lane l runs the reference's compare loop
//      at position 64t+l (consecutive lanes read consecutive LDS bytes:
//      conflict-free) and stores the jump J of its position in LDS.  The first
//      compare, which settles all but ~1/256 of the positions, runs straight-line
//      from wave-uniform registers;
//   3. phase-parallel: lane e < D starts at the first position in phase e and
//      follows the chain through the stored jumps, p += J[p], until it leaves the
//      window; where it lands is the map's value for e.  npos / (mean jump) dependent
//      LDS reads -- ~25 for a 256-position window of ordinary data -- whatever the
//      pattern: walking the exceptional positions (J != D) one by one on the scalar
//      unit instead costs ~40 scalar instructions each, which was fine for plain
//      keywords (4 % exceptional) and 3-10x slower for wildcard patterns, whose
//      capped skips make nearly every position exceptional.
// The map lives in a VGPR (lane e holds map[e]); pulling a phase SET back through it
// is one ballot:  A' = { e : map[e] in A } = ballot(lane < D && (A >> map) & 1).
//
// Users:
//   mm_resolve        (one wave per filter candidate) keeps the set A of phases
//                     that lead to visiting the candidate and pulls it back window
//                     by window: empty -> not on the chain, everything -> on the
//                     chain however the chain entered, domain start -> the chain
//                     starts in phase 0 (monkey_moore.cpp:329).  Two short windows
//                     (<= 768 positions) settle a match in ordinary data; the rest go to
//   mm_resolve2       (one workgroup per candidate) whose waves map the rest of the
//                     candidate's 2048-position tile in parallel; what even that
//                     cannot settle (constant / periodic data) goes to
//   mm_hard_resolve   (several workgroups per candidate) which maps the whole
//                     prefix of the domain in parallel and pulls A through it.
#ifndef MM_TILES_H
#define MM_TILES_H

constexpr int MM_TILE = 2048;                 // positions per tile
constexpr int MM_WAVES = 4;                   // waves per workgroup in the tile kernels
constexpr int MM_MAXD = 32;                   // bytes of a stored phase map of the forward engine's narrow instantiation (keywords of up to 32 symbols)
constexpr int MM_RMAXD = MM_RESOLVER_MAX_KEYWORD;   // ... and in the per-candidate machinery: keywords of up to 64 symbols (D <= 63, a phase
                                              // set is one 64-bit word = a ballot over the map's lanes); longer ones go to the forward engine
typedef uint64_t mm_set_t;                    // a set of phases
constexpr int MM_FAST_STEPS = 3;              // look-back windows of the per-candidate resolver: 64 (128 for keywords beyond 16 symbols),
                                              // then 256, then 512 positions
constexpr int MM_MID_CHUNK = 512;             // mm_resolve2: a workgroup's waves map chunks of this many positions in parallel
constexpr int MM_MID_CAP = 2048;              // candidates mm_resolve hands to mm_resolve2 per scan
constexpr int MM_HARD_PARTS = 64;             // workgroups per hard candidate
constexpr int MM_HARD_CAP = 32;               // hard candidates handled per scan (round 6 measured 128: a periodic keyword -- `abcd`, ~30
                                              // hard candidates per GiB of random bytes -- then settles inside the split pipeline's parts, and
                                              // takes 5.2 ms per 4 GiB instead of 4.2: the flag pass + forward engine on the domains is faster)
constexpr int MM_HARD_MAX_TILES = 8192;       // longest prefix (in tiles) mm_hard_resolve maps
constexpr int MM_JUMP_MATCH = 0x80;           // flag in a stored jump: the compare loop reported a match here

struct MmPlanLds {
   int32_t expected[MMH_MAX_KEYWORD];
   uint32_t cmp_mask[MMH_MAX_KEYWORD];
   int32_t skip_diff[MMH_MAX_KEYWORD];
   int32_t skip_val[MMH_MAX_KEYWORD];       // already max(.,1)
   int8_t bridge[MMH_MAX_KEYWORD];
   uint8_t wst[MMH_MAX_KEYWORD];
   uint8_t skip8[512];                      // dense bad-character table, 8-bit elements
};

// a wave's working set for windows of up to NPOS positions of elements of up to ELEM bytes
template <int NPOS, int ELEM = 2>
struct MmWaveLdsT {
   static constexpr int kPositions = NPOS;
   uint32_t tile_pad[1];                     // tile[-1]: the forward engine reads the dword in front of any tile dword unconditionally
   uint32_t tile[((NPOS + MMH_MAX_KEYWORD + 1) * ELEM + 16 + 3) / 4];
   uint8_t jump[NPOS + 8];                   // J of every position of the window (| MM_JUMP_MATCH); the forward engine
                                             // stores four at a time and shifts the array by 0..3 bytes for that
   uint8_t gmap[NPOS / 64][MM_RMAXD];        // phase map of every group of 64 positions (long windows)
   uint8_t gentry[NPOS / 64];                // forward engine: the chain's phase on entering each group
};
using MmWaveLds = MmWaveLdsT<MM_TILE>;       // whole tiles: hard resolver, forward engine
using MmWaveLdsShort = MmWaveLdsT<512>;      // look-back windows of <= 512 positions: mm_resolve, mm_resolve2, the fused scan

// tell the compiler a value is the same in every lane (keeps it in SGPRs / on the scalar unit)
__device__ __forceinline__ uint32_t mm_uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t mm_uniform64(uint64_t v)
{
   return (uint64_t)mm_uniform((uint32_t)v) | ((uint64_t)mm_uniform((uint32_t)(v >> 32)) << 32);
}

__device__ __forceinline__ void mm_wave_sync()
{
   __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
   __builtin_amdgcn_wave_barrier();
}

// block-cooperative; ends with a __syncthreads()
__device__ __forceinline__ void mm_plan_to_lds(MmPlanLds &P, const mmh_plan_desc &pl)
{
   for (int i = threadIdx.x; i < MMH_MAX_KEYWORD; i += blockDim.x) {
      P.expected[i] = pl.expected[i];
      P.cmp_mask[i] = pl.cmp_mask[i];
      P.bridge[i] = pl.bridge[i];
      P.wst[i] = pl.wst[i];
      P.skip_diff[i] = pl.skip_diff[i];
      int sv = pl.skip_val[i];
      P.skip_val[i] = sv < 1 ? 1 : sv;
   }
   if (pl.elem_bytes == 1) {
      int dflt = pl.default_skip;
      dflt = dflt < 1 ? 1 : dflt;
      for (int i = threadIdx.x; i < 512; i += blockDim.x) {
         P.skip8[i] = (uint8_t)dflt;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < (int)pl.n_skip; i += blockDim.x) {
         int s = pl.skip_val[i];
         P.skip8[pl.skip_diff[i] + 255] = (uint8_t)(s < 1 ? 1 : s);
      }
   }
   __syncthreads();
}

// --------------------------------------------------------------------------
// one tile -> its map
// --------------------------------------------------------------------------

// elements [lo, lo + npos + L - 1) of the domain starting at byte `start`, as raw
// bytes.  The copy is dword aligned on the SOURCE: LDS byte (k + mis) <-> ROM byte
// start + lo*S + k, where mis (returned, wave uniform) is the source misalignment.
// All loads of a lane are issued before the first LDS store (one memory latency
// per tile instead of one per 256 bytes).
template <class WL>
__device__ __forceinline__ int mm_stage_tile(const MmTileArgs &a, WL &W, uint64_t start, int64_t lo, int npos, int lane)
{
   const int S = (int)a.g.S;
   const uint64_t src0 = start + (uint64_t)lo * S;
   // one element more than the tile's own compares need: the caller may evaluate the
   // compare loop AT position npos (the candidate) as well
   int nstage = (npos + (int)a.plan.L) * S;
   if (src0 + (uint64_t)nstage > a.g.nbytes) {
      nstage = (int)(a.g.nbytes - src0);       // never read past the dword that holds the ROM's last byte
   }
   const int mis = (int)(((uintptr_t)(a.g.rom + src0)) & 3);
   const uint32_t *s4 = reinterpret_cast<const uint32_t *>(a.g.rom + src0 - mis);
   const int nw = (nstage + mis + 3) >> 2;
   constexpr int NU = (int)(sizeof(W.tile) / 4 + 63) / 64;      // (5 loads per lane for the short windows, 17 for whole tiles)
   uint32_t v[NU];
#pragma unroll
   for (int u = 0; u < NU; u++) {
      const int k = lane + 64 * u;
      v[u] = k < nw ? s4[k] : 0u;
   }
#pragma unroll
   for (int u = 0; u < NU; u++) {
      const int k = lane + 64 * u;
      if (k < nw) {
         W.tile[k] = v[u];
      }
   }
   return mis;
}

// element idx of the staged tile; `tile` already points at the tile's first byte
__device__ __forceinline__ int mm_tile_elem(const uint8_t *tile, int idx, int S, bool be)
{
   if (S == 1) {
      return tile[idx];
   }
   const int a = tile[2 * idx], b = tile[2 * idx + 1];
   return be ? (a << 8 | b) : (b << 8 | a);
}

__device__ __forceinline__ int mm_tile_skip(const MmTileArgs &a, const MmPlanLds &P, int d)
{
   if (a.g.S == 1) {
      return P.skip8[d + 255];
   }
   int s = a.plan.default_skip;
   s = s < 1 ? 1 : s;
   // (three 64-bit filters on bits 0-5, 6-11 and 12-17 of the delta: with one, a tenth of the lanes passed and every wave
   // ran the list below for every position -- 16-bit searches on the forward engine spent most of their time here)
   if ((a.skip_bloom >> (d & 63)) & (a.skip_bloom2 >> (((uint32_t)d >> 6) & 63)) & (a.skip_bloom3 >> (((uint32_t)d >> 12) & 63)) & 1) {
      const int n = (int)a.plan.n_skip;
      for (int k = 0; k < n; k++) {
         s = P.skip_diff[k] == d ? P.skip_val[k] : s;
      }
   }
   return s;
}

// x mod D for x < 4096 without a division: inv_d = ceil(2^20 / D) is exact while x (inv_d D - 2^20) < 2^20, i.e. for
// x < 2^20 / 126 whatever D <= 127 (and x inv_d < 2^32).  (Until round 5: ceil(2^16 / D), exact for x < 2100 only while
// D <= 31 -- with the resolvers' keywords of up to 64 symbols a 2048-position tile of the hard resolver came out wrong.)
__device__ __forceinline__ uint32_t mm_modd(const MmTileArgs &a, uint32_t x)
{
   return x - ((x * a.inv_d) >> 20) * (a.plan.L - 1);
}

// x mod D for any x (domain positions): a multiply for 32-bit values, the division only beyond
__device__ __forceinline__ uint32_t mm_modd64(const MmTileArgs &a, uint64_t x)
{
   const uint32_t D = a.plan.L - 1;
   if (D == 1) {
      return 0;                                                    // (2^32 / 1 + 1 does not fit inv_d32)
   }
   if ((x >> 32) != 0) {
      return (uint32_t)(x % D);
   }
   const uint32_t q = __umulhi((uint32_t)x, a.inv_d32);          // floor(x / D) or one more
   const int32_t r = (int32_t)((uint32_t)x - q * D);
   return (uint32_t)(r < 0 ? r + (int32_t)D : r);
}

// mm_locate with the divisions taken out: element sizes are 1 or 2, block sizes are powers of
// two in practice (the general case keeps the division)
__device__ __forceinline__ bool mm_locate_fast(const MmTileArgs &a, uint64_t o, uint64_t *b, uint32_t *p, int64_t *j)
{
   const uint32_t sshift = a.g.S - 1;                              // S = 1 or 2
   if (a.g.whole) {
      if (o & sshift) {
         return false;
      }
      *b = 0; *p = 0; *j = (int64_t)(o >> sshift);
      return *j < (int64_t)(a.g.nbytes >> sshift) - (int64_t)a.g.L + 1;
   }
   if (o >= a.g.nbytes) {
      // a SWAR survivor in the padding behind the ROM (the streaming code looks at whole 16-byte chunks):
      // block o / B does not exist -- and "bytes remaining" below would wrap around
      return false;
   }
   const uint64_t blk = a.block_shift < 64 ? o >> a.block_shift : o / a.g.block_bytes;
   const uint64_t first = a.block_shift < 64 ? blk << a.block_shift : blk * a.g.block_bytes;
   const uint64_t r = o - first;
   *b = blk; *p = (uint32_t)(r & sshift); *j = (int64_t)(r >> sshift);
   // mm_domain_nv (mm_internal.h) with shifts
   const uint64_t full = a.g.block_bytes + ((uint64_t)(a.g.L - 1) << sshift);
   const uint64_t remaining = a.g.nbytes - first;
   const uint64_t size = remaining < full ? remaining : full;
   uint64_t count = size >> sshift;
   if ((uint64_t)*p + (count << sshift) > size) {
      count -= 1;
   }
   return *j < (int64_t)count - (int64_t)a.g.L + 1;
}

// the reference's compare loop at position q of the staged tile: does it report a match?
__device__ __forceinline__ bool mm_tile_matches(const MmTileArgs &a, const MmPlanLds &P, const uint8_t *tile, int q)
{
   const int L = (int)a.plan.L, S = (int)a.g.S;
   const bool be = a.g.big_endian != 0;
   for (int i = L - 1; i >= 0; --i) {
      const int ci = mm_tile_elem(tile, q + i, S, be);
      const int pi = mm_tile_elem(tile, q + i + P.bridge[i], S, be);
      if (((uint32_t)((ci - pi) ^ P.expected[i]) & P.cmp_mask[i]) != 0) {
         return false;
      }
   }
   return true;
}

// Steps 1 and 2 of the header: stage positions [lo, lo + npos) of the domain at byte `start`
// and leave the jump of every position (| MM_JUMP_MATCH where the compare loop matched) in
// W.jump.  Returns the staged tile's first byte.
// padded: position q's jump goes to W.jump[q + 4 (q / 32)] (the forward engine's bank-conflict-free layout)
template <class WL>
__device__ __forceinline__ const uint8_t *mm_tile_jumps(const MmTileArgs &a, const MmPlanLds &P, WL &W, uint64_t start,
                                                        int64_t lo, int npos, int lane, bool padded = false)
{
   const int mis = mm_stage_tile(a, W, start, lo, npos, lane);
   mm_wave_sync();

   const int L = (int)a.plan.L, S = (int)a.g.S;
   const bool be = a.g.big_endian != 0;
   const uint8_t *tile = reinterpret_cast<const uint8_t *>(W.tile) + mis;
   // the first compare (keyword position L-1) from wave-uniform registers
   const int e1 = a.plan.expected[L - 1], b1 = a.plan.bridge[L - 1], w1 = a.plan.wst[L - 1];
   const uint32_t m1 = a.plan.cmp_mask[L - 1];
   const int match_jump = (int)a.plan.match_jump;

   const int nballots = (npos + 63) >> 6;
   // software pipeline: the element pair of iteration t+1 is read while t is finished
   int c = 0, pv = 0;
   if (lane < npos) {
      c = mm_tile_elem(tile, lane + L - 1, S, be);
      pv = mm_tile_elem(tile, lane + L - 1 + b1, S, be);
   }
   for (int t = 0; t < nballots; t++) {
      const int q = 64 * t + lane;
      const int qn = q + 64;
      int cn = 0, pn = 0;
      if (qn < npos) {
         cn = mm_tile_elem(tile, qn + L - 1, S, be);
         pn = mm_tile_elem(tile, qn + L - 1 + b1, S, be);
      }
      const bool live = q < npos;
      const int d = c - pv;
      const bool deep = live && ((uint32_t)(d ^ e1) & m1) == 0;
      int J = 1;
      if (live) {
         const int s = mm_tile_skip(a, P, d);
         J = s < w1 ? s : w1;
      }
      if (__ballot(deep) != 0) {
         if (deep) {
            J = match_jump | MM_JUMP_MATCH;
            for (int i = L - 2; i >= 0; --i) {
               const int ci = mm_tile_elem(tile, q + i, S, be);
               const int pi = mm_tile_elem(tile, q + i + P.bridge[i], S, be);
               const int di = ci - pi;
               if (((uint32_t)(di ^ P.expected[i]) & P.cmp_mask[i]) != 0) {
                  const int s = mm_tile_skip(a, P, di);
                  const int w = P.wst[i];
                  J = s < w ? s : w;
                  break;
               }
            }
         }
      }
      if (live) {
         // (a jump below 1 cannot come out of a plan of mm_plan.cpp; the walk below must never stall)
         W.jump[padded ? q + 4 * (q >> 5) : q] = (uint8_t)((J & (MM_JUMP_MATCH - 1)) == 0 ? 1 : J);
      }
      c = cn;
      pv = pn;
   }
   mm_wave_sync();
   return tile;
}

// Long windows: the phase map of every group of 64 positions, W.gmap[g][e] = phase in which
// the chain that enters group g in phase e leaves it.  ngroups * D independent little walks
// (<= 64 / mean-jump steps each) spread over the lanes; composing the group maps afterwards
// costs one LDS lookup per group.  Needs W.jump (mm_tile_jumps).
template <class WL>
__device__ __forceinline__ void mm_group_maps(const MmTileArgs &a, WL &W, int npos, uint32_t lo_mod, int lane,
                                              const uint8_t *J = nullptr)
{
   J = J ? J : W.jump;
   const uint32_t D = a.plan.L - 1;
   const uint32_t ngroups = (uint32_t)(npos + 63) >> 6;
   const uint32_t ntasks = ngroups * D;                        // <= 32 * 63
   for (uint32_t task = (uint32_t)lane; task < ntasks; task += 64) {
      const uint32_t g = (task * a.inv_d) >> 20;                // task / D
      const uint32_t e = task - g * D;
      const uint32_t first = 64 * g;
      const uint32_t end = first + 64 < (uint32_t)npos ? first + 64 : (uint32_t)npos;
      uint32_t off = e + D - mm_modd(a, lo_mod + first);        // first position of the group in phase e
      off = off >= D ? off - D : off;
      uint32_t p = first + off;
      while (p < end) {
         p += J[p] & (MM_JUMP_MATCH - 1);
      }
      uint32_t v = mm_modd(a, lo_mod + end) + (p - end);        // it left the group at p in [end, end + D)
      v = v >= D ? v - D : v;
      W.gmap[g][e] = (uint8_t)v;
   }
   mm_wave_sync();
}

// Map of positions [lo, lo + npos) of the domain at byte `start`; lo_mod = lo mod D.  Lane
// e < D returns the exit phase of entry phase e (other lanes: their own number, i.e. the
// identity).  *end_matches: does the compare loop match AT position npos (the candidate in
// front of which the window ends)?  Only meaningful when the domain holds L elements from
// there, which the caller checked.
template <class WL>
__device__ __forceinline__ uint32_t mm_tile_map(const MmTileArgs &a, const MmPlanLds &P, WL &W, uint64_t start, int64_t lo,
                                                int npos, uint32_t lo_mod, int lane, bool *end_matches = nullptr, bool want_end = true)
{
   start = mm_uniform64(start);
   lo = (int64_t)mm_uniform64((uint64_t)lo);
   npos = (int)mm_uniform((uint32_t)npos);
   lo_mod = mm_uniform(lo_mod);
   const uint8_t *tile = mm_tile_jumps(a, P, W, start, lo, npos, lane);
   const uint32_t D = a.plan.L - 1;

   uint32_t v = (uint32_t)lane;
   if (npos > 512) {
      // long window: group maps in parallel, then lane e follows phase e through them
      mm_group_maps(a, W, npos, lo_mod, lane);
      if ((uint32_t)lane < D) {
         const int ngroups = (npos + 63) >> 6;
         for (int g = 0; g < ngroups; g++) {
            v = W.gmap[g][v];
         }
      }
   }
   else if ((uint32_t)lane < D) {
      uint32_t p = (uint32_t)lane + D - lo_mod;            // first position of the window in phase `lane`
      p = p >= D ? p - D : p;
      while (p < (uint32_t)npos) {
         p += W.jump[p] & (MM_JUMP_MATCH - 1);
      }
      // the chain left the window at p in [npos, npos + D): its phase
      v = mm_modd(a, lo_mod + (uint32_t)npos) + (p - (uint32_t)npos);
      v = v >= D ? v - D : v;
   }
   // (want_end: a caller that only sometimes wants the answer passes its flag instead of a pointer that is sometimes null --
   // the compiler kept such a flag in scratch memory: 8 bytes per lane in every resolver kernel until round 5)
   if (end_matches && want_end) {
      *end_matches = mm_tile_matches(a, P, tile, npos);
   }
   mm_wave_sync();                                        // the wave's LDS may be restaged now
   return v;
}

// A' = { e < D : map[e] in A }; `map` is the per-lane value mm_tile_map returned
__device__ __forceinline__ mm_set_t mm_pull_back(mm_set_t A, uint32_t D, uint32_t map, int lane)
{
   return (mm_set_t)__ballot((uint32_t)lane < D && ((A >> map) & 1ull) != 0);
}

// all D phases
__device__ __forceinline__ mm_set_t mm_full_set(uint32_t D) { return D >= 64 ? ~0ull : ((1ull << D) - 1ull); }

// --------------------------------------------------------------------------
// fast resolver: one wave per candidate, bounded look-back
// --------------------------------------------------------------------------

struct MmResolveArgs {
   MmTileArgs t;
   const uint64_t *cand;                      // MM_CAND_LISTS candidate lists of list_cap entries (filter output)
   const unsigned long long *list_count;      // their counters, MM_LIST_STRIDE words apart
   uint64_t list_cap;
   unsigned long long *total_out;             // compact candidate count (~0: a list overflowed), for the later stages
   uint64_t out_cap;
   // One result slot per candidate: the reported value of a match, MM_NO_MATCH otherwise.
   // (A shared append counter would serialise: one device-scope atomic address takes
   // ~11 ns per add, i.e. ~50 us for the bench's 4 K candidates.)
   uint64_t *out;
   unsigned long long *tiles_walked;          // MM_STAT_STRIPES striped statistics counters
   uint64_t base_offset;                      // added to reported byte offsets
   uint32_t max_candidates;                   // above this the host switches engines
   // candidates the two short windows could not settle, for mm_resolve2
   uint64_t *mid_off;                         // [MM_MID_CAP] candidate byte offset
   uint64_t *mid_hi;                          // [MM_MID_CAP] frontier (domain position)
   mm_set_t *mid_set;                         // [MM_MID_CAP] acceptable phases at the frontier
   uint32_t *mid_slot;                        // [MM_MID_CAP] the candidate's result slot
   unsigned int *mid_count;
   // Flag pass (after a scan whose left-over lists overflowed): instead of handing undecided
   // candidates on, set the bit of their domain -- the host then runs the forward engine on
   // exactly those domains.  nullptr in a normal scan.
   uint32_t *flag_bits;
};

constexpr uint64_t MM_NO_MATCH = ~0ull;

__device__ __forceinline__ uint64_t mm_report_value(const MmGeom &g, uint64_t o, uint64_t base_offset)
{
   return g.whole ? o / g.S : o + base_offset;
}

// shared by the waves of a resolver workgroup
struct MmResolveLds {
   unsigned long long excl[MM_CAND_LISTS];   // candidates in the lists before list c
   unsigned long long ncand;                 // all of them (~0: a list overflowed)
   unsigned int walked;                      // windows mapped by this workgroup
};

// (the plain resolver kernels read the candidate lists of an EARLIER launch: ordinary loads)
__device__ __forceinline__ unsigned long long mm_load_shared(const unsigned long long *p)
{
   return *p;
}

// The filter's MM_CAND_LISTS (= 64 = one per lane) lists get one compact numbering: candidate
// ci is entry ci - excl[c] of the list c with excl[c] <= ci < excl[c+1].  Wave 0 reads the
// counters (64 cache lines) once for the workgroup; call before a __syncthreads().
template <class A>
__device__ __forceinline__ void mm_resolve_prefix(const A &a, MmResolveLds &R)
{
   if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      const unsigned long long my_count = mm_load_shared(&a.list_count[lane * MM_LIST_STRIDE]);
      unsigned long long incl = my_count;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
         const unsigned long long v = __shfl_up(incl, d);
         incl += lane >= d ? v : 0ull;
      }
      R.excl[lane] = incl - my_count;
      const bool overflow = __ballot(my_count > a.list_cap) != 0;
      if (lane == 63) {
         R.ncand = overflow ? ~0ull : incl;
         R.walked = 0;
      }
   }
}

// candidate number ci of the compact numbering (wave uniform)
template <class A>
__device__ __forceinline__ uint64_t mm_candidate(const A &a, unsigned long long excl, uint64_t ci)
{
   const int list = __popcll(__ballot(excl <= ci)) - 1;
   const unsigned long long *slot = reinterpret_cast<const unsigned long long *>(a.cand) + (uint64_t)list * a.list_cap +
                                    (ci - __shfl(excl, list));
   return mm_uniform64(mm_load_shared(slot));
}

// One candidate, one wave: is byte offset o reported by the reference?  1 yes (on the chain of its
// domain and the compare loop matches), 0 no, -1 the two short windows could not tell (*hi, *set:
// where the pull-back stands, for mm_resolve2; *dom: its domain), -2 not an alignment of any domain.
template <class A, class WL>
__device__ __forceinline__ int mm_resolve_candidate(const A &a, const MmPlanLds &P, WL &W, uint64_t o, int lane,
                                                    unsigned long long *walked, int64_t *hi_out, mm_set_t *set_out, uint64_t *dom_out)
{
   const uint32_t D = a.t.plan.L - 1;
   const mm_set_t full = mm_full_set(D);
   uint64_t b; uint32_t p; int64_t jc;
   if (!mm_locate_fast(a.t, o, &b, &p, &jc)) {
      return -2;                               // SWAR survivor that is not an alignment of any domain (file tail, 16-bit odd boundary)
   }
   const uint64_t start = mm_domain_start(a.t.g, b, p);
   *dom_out = a.t.g.whole ? 0 : b * a.t.g.S + p;

   mm_set_t S = 1ull << mm_modd64(a.t, (uint64_t)jc);
   int64_t hi = jc;
   int verdict = -1;                           // 1 visited, 0 not visited, -1 undecided
   if (jc == 0) {
      // first alignment of its domain: always visited, but is it a match?  (no window to
      // stage in front of it: read the elements directly)
      bool matched;
      mm_step(a.t.plan, [&](int64_t k) { return mm_elem(a.t.g, start, k); }, 0, &matched);
      verdict = matched ? 1 : 0;
   }
   // Look-back windows of 64 (keywords beyond 16 symbols: 128), then 256, then 512 positions in front of the frontier:
   // a true match pulls every phase onto itself within a few keyword lengths (that is what the bad-character
   // rule is for), so the first, short window usually settles it -- one pass of the position-parallel jump loop
   // instead of up to four (round 2 started with an aligned window of up to 256 positions: at 64 K candidates the
   // tail kernel was bound by the instructions of that loop).
   for (int step = 0; step < MM_FAST_STEPS && hi > 0 && verdict < 0; step++) {
      const int64_t size = step == 0 ? (a.t.plan.L <= 16 ? 64 : a.t.plan.L <= 32 ? 128 : 256) : (int64_t)128 << step;
      const int64_t lo = hi > size ? hi - size : 0;
      bool is_match = true;
      const uint32_t map = mm_tile_map(a.t, P, W, start, lo, (int)(hi - lo), mm_modd64(a.t, (uint64_t)lo), lane, &is_match, step == 0);
      (*walked)++;
      if (!is_match) {
         verdict = 0;                           // survived the SWAR conditions only
         break;
      }
      S = mm_pull_back(S, D, map, lane);
      hi = lo;
      if (S == full || S == 0) {
         verdict = S ? 1 : 0;
         break;
      }
   }
   if (verdict < 0 && hi == 0) {
      verdict = (S & 1ull) ? 1 : 0;            // domain start: the chain is in phase 0
   }
   *hi_out = hi;
   *set_out = S;
   return verdict;
}

// ---- keywords of 65 .. 128 symbols (round 6): D = 64 .. 127 phases, two per lane ------------------------------------
// Until round 5 such keywords always ran on the forward engine (8 - 11 ms per 4 GiB against 0.73 on the candidate path).
// The look-back windows work the same way with lane e following phases e AND e + 64 and the phase set as two ballots;
// only this first resolver knows such maps -- the stored maps of mm_resolve2 / mm_hard_resolve stay 64 bytes -- so a
// candidate the windows do not settle is counted but not handed on, and the host reads any left-over of such a keyword
// as "more than the second phase takes": it runs the forward engine on the domains concerned (run_flagged_domains), as
// it does for any such overflow.
struct MmSet2 {
   uint64_t lo, hi;                           // phases 0 .. 63, 64 .. 127
};

// Map of the (<= 512) positions [lo, lo + npos): *v0 / *v1 = exit phase of entry phase lane / lane + 64 (lanes that are
// no phase: their own number).  See mm_tile_map.
template <class WL>
__device__ __forceinline__ void mm_tile_map2(const MmTileArgs &a, const MmPlanLds &P, WL &W, uint64_t start, int64_t lo, int npos,
                                             uint32_t lo_mod, int lane, uint32_t *v0, uint32_t *v1, bool *end_matches, bool want_end)
{
   start = mm_uniform64(start);
   lo = (int64_t)mm_uniform64((uint64_t)lo);
   npos = (int)mm_uniform((uint32_t)npos);
   lo_mod = mm_uniform(lo_mod);
   const uint8_t *tile = mm_tile_jumps(a, P, W, start, lo, npos, lane);
   const uint32_t D = a.plan.L - 1;
   const uint32_t end_mod = mm_modd(a, lo_mod + (uint32_t)npos);
   uint32_t v[2];
#pragma unroll
   for (int h = 0; h < 2; h++) {
      const uint32_t e = (uint32_t)lane + 64u * h;
      v[h] = e;
      if (e < D) {
         uint32_t p = e + D - lo_mod;                       // first position of the window in phase e
         p = p >= D ? p - D : p;
         while (p < (uint32_t)npos) {
            p += W.jump[p] & (MM_JUMP_MATCH - 1);
         }
         uint32_t x = end_mod + (p - (uint32_t)npos);       // the chain left the window at p in [npos, npos + D)
         v[h] = x >= D ? x - D : x;
      }
   }
   *v0 = v[0];
   *v1 = v[1];
   if (end_matches && want_end) {
      *end_matches = mm_tile_matches(a, P, tile, npos);
   }
   mm_wave_sync();
}

__device__ __forceinline__ bool mm_set2_has(const MmSet2 &A, uint32_t v)
{
   return ((v < 64 ? A.lo >> v : A.hi >> (v - 64)) & 1ull) != 0;
}

// mm_resolve_candidate for D = 64 .. 127.  1 / 0 / -2 as there; -1: the windows (256, 512, 512 positions) could not tell.
template <class A, class WL>
__device__ __forceinline__ int mm_resolve_candidate_long(const A &a, const MmPlanLds &P, WL &W, uint64_t o, int lane,
                                                         unsigned long long *walked, uint64_t *dom_out)
{
   const uint32_t D = a.t.plan.L - 1;
   const MmSet2 full = {~0ull, D >= 128 ? ~0ull : ((1ull << (D - 64)) - 1ull)};
   uint64_t b; uint32_t p; int64_t jc;
   if (!mm_locate_fast(a.t, o, &b, &p, &jc)) {
      return -2;
   }
   const uint64_t start = mm_domain_start(a.t.g, b, p);
   *dom_out = a.t.g.whole ? 0 : b * a.t.g.S + p;
   const uint32_t ph = mm_modd64(a.t, (uint64_t)jc);
   MmSet2 S = {ph < 64 ? 1ull << ph : 0ull, ph >= 64 ? 1ull << (ph - 64) : 0ull};
   int64_t hi = jc;
   int verdict = -1;
   if (jc == 0) {
      bool matched;
      mm_step(a.t.plan, [&](int64_t k) { return mm_elem(a.t.g, start, k); }, 0, &matched);
      verdict = matched ? 1 : 0;
   }
   for (int step = 0; step < MM_FAST_STEPS && hi > 0 && verdict < 0; step++) {
      const int64_t size = step == 0 ? 256 : 512;
      const int64_t lo = hi > size ? hi - size : 0;
      bool is_match = true;
      uint32_t v0, v1;
      mm_tile_map2(a.t, P, W, start, lo, (int)(hi - lo), mm_modd64(a.t, (uint64_t)lo), lane, &v0, &v1, &is_match, step == 0);
      (*walked)++;
      if (!is_match) {
         verdict = 0;
         break;
      }
      const uint64_t nlo = __ballot((uint32_t)lane < D && mm_set2_has(S, v0));
      const uint64_t nhi = __ballot((uint32_t)lane + 64u < D && mm_set2_has(S, v1));
      S.lo = nlo;
      S.hi = nhi;
      hi = lo;
      if ((S.lo == full.lo && S.hi == full.hi) || (S.lo | S.hi) == 0) {
         verdict = (S.lo | S.hi) ? 1 : 0;
         break;
      }
   }
   if (verdict < 0 && hi == 0) {
      verdict = (S.lo & 1ull) ? 1 : 0;         // domain start: the chain is in phase 0
   }
   return verdict;
}

// the resolver of a kernel instantiation: LONG = keywords beyond 64 symbols (hi = -1 marks what mm_resolve_hand_over gets)
template <bool LONG, class A, class WL>
__device__ __forceinline__ int mm_resolve_any(const A &a, const MmPlanLds &P, WL &W, uint64_t o, int lane, unsigned long long *walked,
                                              int64_t *hi_out, mm_set_t *set_out, uint64_t *dom_out)
{
   if constexpr (LONG) {
      *hi_out = -1;
      *set_out = 0;
      return mm_resolve_candidate_long(a, P, W, o, lane, walked, dom_out);
   }
   else {
      return mm_resolve_candidate(a, P, W, o, lane, walked, hi_out, set_out, dom_out);
   }
}

// lane 0 of the wave that could not settle candidate ci: hand it to mm_resolve2 (or, in the flag
// pass, mark its domain)
template <class A>
__device__ __forceinline__ void mm_resolve_hand_over(const A &a, uint64_t o, uint64_t ci, int64_t hi, mm_set_t set, uint64_t dom)
{
   if (a.flag_bits) {
      atomicOr(&a.flag_bits[dom >> 5], 1u << (dom & 31));
      return;
   }
   if (hi < 0) {
      // (a keyword beyond 64 symbols: counted, not stored -- the second phase does not know its maps, and the host reads
      // ANY left-over of such a keyword as "more than the second phase takes", finish_pipeline)
      atomicAdd(a.mid_count, 1u);
      return;
   }
   unsigned int slot = atomicAdd(a.mid_count, 1u);   // beyond MM_MID_CAP: the host sees the count and switches engines
   if (slot < MM_MID_CAP) {
      a.mid_off[slot] = o;
      a.mid_hi[slot] = (uint64_t)hi;
      a.mid_set[slot] = set;
      a.mid_slot[slot] = (uint32_t)ci;
   }
}

template <bool LONG>
__device__ __forceinline__ void mm_resolve_body(const MmResolveArgs &a, const MmPlanLds &P, MmWaveLdsShort &W, MmResolveLds &R)
{
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   const unsigned long long excl = R.excl[lane];
   const unsigned long long ncand = R.ncand;
   if (blockIdx.x == 0 && threadIdx.x == 0) {
      *a.total_out = ncand;
   }
   if (ncand > a.out_cap || ncand > a.max_candidates) {
      return;                                  // dense input: the host runs another engine
   }
   const uint64_t nwaves = (uint64_t)gridDim.x * MM_WAVES;
   unsigned long long walked = 0;

   for (uint64_t ci = (uint64_t)blockIdx.x * MM_WAVES + wave; ci < ncand; ci += nwaves) {
      // wave uniform from here on: the index arithmetic below runs on the scalar unit
      const uint64_t o = mm_candidate(a, excl, ci);
      int64_t hi = 0; mm_set_t set = 0; uint64_t dom = 0;
      const int verdict = mm_resolve_any<LONG>(a, P, W, o, lane, &walked, &hi, &set, &dom);
      if (lane == 0) {
         a.out[ci] = verdict == 1 ? mm_report_value(a.t.g, o, a.base_offset) : MM_NO_MATCH;
         if (verdict == -1) {
            mm_resolve_hand_over(a, o, ci, hi, set, dom);
         }
      }
   }
   // statistics: one global atomic per workgroup (4 K of them on 16 addresses would sit on the kernel's tail)
   if (lane == 0 && walked) {
      atomicAdd(&R.walked, (unsigned int)walked);
   }
   __syncthreads();
   if (threadIdx.x == 0 && R.walked) {
      atomicAdd(a.tiles_walked + (blockIdx.x % MM_STAT_STRIPES), (unsigned long long)R.walked);
   }
}

#ifndef MM_FILTER_SHAPE_UNIT   // (not a template: defined once, in mm_kernels.hip)
__global__ __launch_bounds__(64 * MM_WAVES) void mm_resolve(MmResolveArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLdsShort Wv[MM_WAVES];
   __shared__ MmResolveLds R;
   mm_resolve_prefix(a, R);
   mm_plan_to_lds(P, a.t.plan);                 // ends with a __syncthreads()
   mm_resolve_body<false>(a, P, Wv[threadIdx.x >> 6], R);
}

// ... for keywords of 65 .. 128 symbols
__global__ __launch_bounds__(64 * MM_WAVES) void mm_resolve_long(MmResolveArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLdsShort Wv[MM_WAVES];
   __shared__ MmResolveLds R;
   mm_resolve_prefix(a, R);
   mm_plan_to_lds(P, a.t.plan);
   mm_resolve_body<true>(a, P, Wv[threadIdx.x >> 6], R);
}
#endif

// --------------------------------------------------------------------------
// second resolver: one workgroup per left-over candidate, waves in parallel
// --------------------------------------------------------------------------
//
// A candidate mm_resolve could not settle has its frontier `hi` somewhere inside a tile of
// MM_TILE positions.  The waves of a workgroup map the rest of that tile, [T, hi) with T the
// tile's first position, in chunks of MM_MID_CHUNK positions AT THE SAME TIME (a chunk's map
// does not depend on how the chain enters it); wave 0 then pulls the phase set back through
// the chunk maps, top down.  Still undecided at T > 0 -> mm_hard_resolve.

struct MmResolve2Args {
   MmTileArgs t;
   const uint64_t *mid_off;
   const uint64_t *mid_hi;
   const mm_set_t *mid_set;
   const uint32_t *mid_slot;
   const unsigned int *mid_count;
   uint64_t *out;                             // result slots (see MmResolveArgs)
   unsigned long long *tiles_walked;
   uint64_t base_offset;
   uint64_t *hard_off;                        // [MM_HARD_CAP] candidate byte offset
   uint64_t *hard_hi;                         // [MM_HARD_CAP] frontier (domain position, multiple of MM_TILE)
   mm_set_t *hard_set;                        // [MM_HARD_CAP] acceptable phases at the frontier
   uint32_t *hard_slot;                       // [MM_HARD_CAP] the candidate's result slot
   unsigned int *hard_count;
};

__device__ __forceinline__ void mm_resolve2_body(const MmResolve2Args &a, const MmPlanLds &P, MmWaveLdsShort &W,
                                                 uint8_t (&maps)[MM_TILE / MM_MID_CHUNK][MM_RMAXD])
{
   const uint32_t D = a.t.plan.L - 1;
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   const mm_set_t full = mm_full_set(D);
   unsigned int n = *a.mid_count;
   n = n > MM_MID_CAP ? 0u : n;                // overflow: the host discards this scan and switches engines
   unsigned long long walked = 0;
   for (unsigned int e = blockIdx.x; e < n; e += gridDim.x) {
      const uint64_t o = a.mid_off[e];
      const int64_t hi = (int64_t)a.mid_hi[e];
      uint64_t b; uint32_t p; int64_t jc;
      mm_locate_fast(a.t, o, &b, &p, &jc);
      const uint64_t start = mm_domain_start(a.t.g, b, p);
      const int64_t T = (hi - 1) & ~(int64_t)(MM_TILE - 1);
      const int nchunks = (int)((hi - T + MM_MID_CHUNK - 1) / MM_MID_CHUNK);
      for (int k = wave; k < nchunks; k += MM_WAVES) {
         const int64_t lo = T + (int64_t)k * MM_MID_CHUNK;
         const int npos = (int)(hi - lo < MM_MID_CHUNK ? hi - lo : MM_MID_CHUNK);
         const uint32_t map = mm_tile_map(a.t, P, W, start, lo, npos, mm_modd64(a.t, (uint64_t)lo), lane);
         if (lane < MM_RMAXD) {
            maps[k][lane] = (uint8_t)map;
         }
         walked++;
      }
      __syncthreads();
      if (wave == 0) {
         mm_set_t A = a.mid_set[e];
         int verdict = -1;                       // 1 visited, 0 not visited, -1 undecided
         for (int k = nchunks - 1; k >= 0; k--) {
            A = mm_pull_back(A, D, maps[k][lane & (MM_RMAXD - 1)], lane);
            if (A == full || A == 0) {
               verdict = A ? 1 : 0;
               break;
            }
         }
         if (verdict < 0 && T == 0) {
            verdict = (A & 1ull) ? 1 : 0;        // domain start: the chain is in phase 0
         }
         if (lane == 0) {
            if (verdict == 1) {
               a.out[a.mid_slot[e]] = mm_report_value(a.t.g, o, a.base_offset);   // the slot already says "no match"
            }
            else if (verdict < 0) {
               unsigned int slot = atomicAdd(a.hard_count, 1u);
               if (slot < MM_HARD_CAP) {
                  a.hard_off[slot] = o;
                  a.hard_hi[slot] = (uint64_t)T;
                  a.hard_set[slot] = A;
                  a.hard_slot[slot] = a.mid_slot[e];
               }
            }
         }
      }
      __syncthreads();                           // maps[] is rewritten by the next candidate
   }
   if (lane == 0 && walked) {
      atomicAdd(a.tiles_walked + (blockIdx.x % MM_STAT_STRIPES), walked);
   }
}

#ifndef MM_FILTER_SHAPE_UNIT   // (not a template: defined once, in mm_kernels.hip)
__global__ __launch_bounds__(64 * MM_WAVES) void mm_resolve2(MmResolve2Args a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLdsShort Wv[MM_WAVES];
   __shared__ uint8_t maps[MM_TILE / MM_MID_CHUNK][MM_RMAXD];
   if (*a.mid_count == 0) {
      return;
   }
   mm_plan_to_lds(P, a.t.plan);
   mm_resolve2_body(a, P, Wv[threadIdx.x >> 6], maps);
}
#endif

// --------------------------------------------------------------------------
// hard resolver: MM_HARD_PARTS workgroups map a candidate's whole prefix
// --------------------------------------------------------------------------
//
// grid = (MM_HARD_PARTS, MM_HARD_CAP).  The 4*PARTS waves of candidate i take the
// tiles of [0, hi) round-robin and write each tile map (64 B) to scratch; the
// workgroup that finishes last pulls the phase set back through them.

struct MmHardArgs {
   MmTileArgs t;
   const uint64_t *hard_off;
   const uint64_t *hard_hi;
   const mm_set_t *hard_set;
   const uint32_t *hard_slot;
   const unsigned int *hard_count;
   unsigned int *done;                        // [MM_HARD_CAP] arrival tickets (zeroed per scan)
   unsigned int *overflow;                    // set when a prefix is too long: the host runs another engine
   uint8_t *scratch;                          // [MM_HARD_CAP][MM_HARD_MAX_TILES][MM_RMAXD]
   uint64_t *out;                             // result slots (see MmResolveArgs)
   unsigned long long *tiles_walked;
   uint64_t base_offset;
};

__device__ __forceinline__ void mm_hard_tiles(const MmHardArgs &a, const MmPlanLds &P, MmWaveLds &W, uint64_t start,
                                              uint64_t ntiles, uint8_t *maps)
{
   const int D = (int)a.t.plan.L - 1;
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   unsigned long long walked = 0;
   for (uint64_t t = (uint64_t)blockIdx.x * MM_WAVES + wave; t < ntiles; t += MM_HARD_PARTS * MM_WAVES) {
      const int64_t lo = (int64_t)(t * MM_TILE);
      const uint32_t map = mm_tile_map(a.t, P, W, start, lo, MM_TILE, mm_modd64(a.t, (uint64_t)lo), lane);
      if (lane < MM_RMAXD) {
         maps[t * MM_RMAXD + lane] = (uint8_t)map;
      }
      walked++;
   }
   if (lane == 0 && walked) {
      atomicAdd(a.tiles_walked + (blockIdx.x % MM_STAT_STRIPES), walked);
   }
}

#ifndef MM_FILTER_SHAPE_UNIT   // (not a template: defined once, in mm_kernels.hip)
__global__ __launch_bounds__(64 * MM_WAVES) void mm_hard_resolve(MmHardArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLds Wv[MM_WAVES];
   __shared__ uint8_t pull[128][MM_RMAXD];
   __shared__ int is_last;
   __shared__ mm_set_t sh_set;
   __shared__ int sh_verdict;                            // 1 visited, 0 not visited, -1 undecided
   const int D = (int)a.t.plan.L - 1;
   const int wave = threadIdx.x >> 6;
   const int lane = threadIdx.x & 63;
   const unsigned int i = blockIdx.y;
   const unsigned int nhard = *a.hard_count;
   if (nhard > MM_HARD_CAP) {
      if (threadIdx.x == 0 && blockIdx.x == 0 && i == 0) {
         *a.overflow = 1;
      }
      return;
   }
   if (i >= nhard) {
      return;
   }
   const uint64_t o = a.hard_off[i];
   const uint64_t hi = a.hard_hi[i];
   const uint64_t ntiles = hi / MM_TILE;
   if (ntiles > MM_HARD_MAX_TILES) {
      if (threadIdx.x == 0 && blockIdx.x == 0) {
         *a.overflow = 1;
      }
      return;
   }
   mm_plan_to_lds(P, a.t.plan);
   uint64_t b; uint32_t p; int64_t jc;
   mm_locate_fast(a.t, o, &b, &p, &jc);
   const uint64_t start = mm_domain_start(a.t.g, b, p);
   uint8_t *maps = a.scratch + (uint64_t)i * MM_HARD_MAX_TILES * MM_RMAXD;

   mm_hard_tiles(a, P, Wv[wave], start, ntiles, maps);

   // last-arriver pattern (agent-scope release / acquire around the ticket)
   __threadfence();
   __syncthreads();
   if (threadIdx.x == 0) {
      unsigned int ticket = atomicAdd(&a.done[i], 1u);
      is_last = ticket == MM_HARD_PARTS - 1;
   }
   __syncthreads();
   if (!is_last) {
      return;
   }
   __threadfence();

   const mm_set_t full = mm_full_set((uint32_t)D);
   if (threadIdx.x == 0) {
      sh_set = a.hard_set[i];
      sh_verdict = -1;
   }
   int64_t t_hi = (int64_t)ntiles;                       // maps [0, t_hi) still to pull through (block uniform)
   while (true) {
      __syncthreads();
      if (sh_verdict >= 0) {
         break;
      }
      if (t_hi == 0) {
         __syncthreads();
         if (threadIdx.x == 0) {
            sh_verdict = (sh_set & 1ull) ? 1 : 0;         // domain start: the chain is in phase 0
         }
         continue;
      }
      const int64_t t_lo = t_hi > 128 ? t_hi - 128 : 0;
      const int n = (int)(t_hi - t_lo);
      {
         // volatile: these bytes were written by other workgroups during this launch
         const volatile uint32_t *src = reinterpret_cast<const volatile uint32_t *>(maps + t_lo * MM_RMAXD);
         uint32_t *dst = reinterpret_cast<uint32_t *>(&pull[0][0]);
         for (int k = threadIdx.x; k < n * (MM_RMAXD / 4); k += blockDim.x) {
            dst[k] = src[k];
         }
      }
      __syncthreads();
      if (wave == 0) {
         mm_set_t A = sh_set;
         for (int k = n - 1; k >= 0; k--) {
            const int v = pull[k][lane & (MM_RMAXD - 1)];
            A = (mm_set_t)__ballot(lane < D && ((A >> v) & 1ull));
            if (A == full || A == 0) {
               break;
            }
         }
         if (lane == 0) {
            sh_set = A;
            if (A == full || A == 0) {
               sh_verdict = A ? 1 : 0;
            }
         }
      }
      t_hi = t_lo;
   }
   if (threadIdx.x == 0 && sh_verdict == 1) {
      a.out[a.hard_slot[i]] = mm_report_value(a.t.g, o, a.base_offset);
   }
}
#endif

#endif
