// mm_tiles.h -- exact emulation of the reference's skip chain, tile by tile.
// Included by mm_kernels.hip (device code only).
//
// Chain state = the next position the reference will visit.  Every jump is in
// [1, D] with D = L-1, so at any boundary a the next visited position lies in
// [a, a+D) and is identified by its PHASE (position mod D).  Processing position
// j moves "phase j mod D" to "phase (j + J(j)) mod D" and leaves the others
// alone; a jump of exactly D changes nothing ("exceptional" positions are the
// ones with J != D: ~L/256 of them on random bytes).  The effect of a run of
// positions is a map Z_D -> Z_D; maps of adjacent runs compose.
//
// One wavefront turns a tile of MM_TILE positions into its map (mm_tile_map):
//   1. coalesced copy of the tile's bytes into LDS, all loads in flight at once;
//   2. position-parallel: iteration t, lane l runs the reference's compare loop
//      at position 64t+l (consecutive lanes read consecutive LDS bytes:
//      conflict-free).  The first compare, which settles all but ~1/256 of the
//      positions, runs straight-line from wave-uniform registers;
//   3. the ballot of the exceptional lanes is walked IN POSITION ORDER on the
//      scalar unit: the map lives in SGPRs as packed 4-bit (D <= 16) or 8-bit
//      phases and "every entry equal to r becomes r2" is a handful of SWAR
//      scalar instructions.  No per-lane maps, no composition pass.
//
// Users:
//   mm_resolve        (one wave per filter candidate) keeps the set A of phases
//                     that lead to visiting the candidate and pulls it back tile
//                     by tile, A' = {e : map(e) in A}: empty -> not on the chain,
//                     everything -> on the chain however the chain entered,
//                     domain start -> the chain starts in phase 0
//                     (monkey_moore.cpp:329).  Bounded look-back; candidates it
//                     cannot settle go to
//   mm_hard_resolve   (several workgroups per candidate) which maps the whole
//                     prefix of the domain in parallel and pulls A through it.
#ifndef MM_TILES_H
#define MM_TILES_H

constexpr int MM_TILE = 2048;                 // positions per tile (32 ballots of 64)
constexpr int MM_WAVES = 4;                   // waves per workgroup in the tile kernels
constexpr int MM_MAXD = MMH_MAX_KEYWORD;      // bytes of a stored tile map
constexpr int MM_FAST_STEPS = 4;              // look-back windows of mm_resolve: <= 256, 512, 1024, 2048 positions
constexpr int MM_HARD_PARTS = 64;             // workgroups per hard candidate
constexpr int MM_HARD_CAP = 32;               // hard candidates handled per scan
constexpr int MM_HARD_MAX_TILES = 8192;       // longest prefix (in tiles) mm_hard_resolve maps

// wave-uniform constants shared by the tile kernels (kernel arguments -> SGPRs)
struct MmTileArgs {
   MmGeom g;
   mmh_plan_desc plan;
   uint32_t inv_d;          // ceil(65536 / D): x / D == (x * inv_d) >> 16 for x < 2048
   uint64_t skip_bloom;     // bit (d & 63) set for every listed bad-character diff
};

struct MmPlanLds {
   int32_t expected[MMH_MAX_KEYWORD];
   uint32_t cmp_mask[MMH_MAX_KEYWORD];
   int32_t skip_diff[MMH_MAX_KEYWORD];
   int32_t skip_val[MMH_MAX_KEYWORD];       // already max(.,1)
   int8_t bridge[MMH_MAX_KEYWORD];
   uint8_t wst[MMH_MAX_KEYWORD];
   uint8_t skip8[512];                      // dense bad-character table, 8-bit elements
};

struct MmWaveLds {
   uint32_t tile[((MM_TILE + MMH_MAX_KEYWORD + 1) * 2 + 16) / 4];
};

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
// phase maps in scalar registers
// --------------------------------------------------------------------------
//
// BITS = 4: 16 entries in one 64-bit word (D <= 16); BITS = 8: 32 entries in four.
// update(r, r2): every entry that currently equals r becomes r2 -- exact per-field
// zero detection (no borrow across fields), then a masked xor.

template <int BITS>
struct MmPhaseMap {
   static constexpr int WORDS = BITS == 4 ? 1 : 4;
   static constexpr int PER_WORD = 64 / BITS;
   static constexpr uint64_t ONES = BITS == 4 ? 0x1111111111111111ull : 0x0101010101010101ull;
   static constexpr uint64_t LOW = BITS == 4 ? 0x7777777777777777ull : 0x7F7F7F7F7F7F7F7Full;
   static constexpr uint64_t HIGH = BITS == 4 ? 0x8888888888888888ull : 0x8080808080808080ull;
   static constexpr uint64_t FIELD = (1ull << BITS) - 1;
   uint64_t w[WORDS];

   __device__ __forceinline__ void identity()
   {
#pragma unroll
      for (int k = 0; k < WORDS; k++) {
         uint64_t v = 0;
#pragma unroll
         for (int e = 0; e < PER_WORD; e++) {
            v |= (uint64_t)((k * PER_WORD + e) & (int)FIELD) << (BITS * e);
         }
         w[k] = v;
      }
   }

   __device__ __forceinline__ void update(uint32_t r, uint32_t r2)
   {
      const uint64_t rr = (uint64_t)r * ONES;
      const uint64_t flip = (uint64_t)(r ^ r2) * ONES;
#pragma unroll
      for (int k = 0; k < WORDS; k++) {
         const uint64_t x = w[k] ^ rr;
         const uint64_t z = ~(((x & LOW) + LOW) | x) & HIGH;          // top bit of a field set iff field == r
         const uint64_t m = (z >> (BITS - 1)) * FIELD;                 // whole-field mask
         w[k] ^= m & flip;
      }
   }

   __device__ __forceinline__ uint32_t get(int e) const
   {
      uint64_t word = w[0];
      if (WORDS > 1) {
         const int k = e / PER_WORD;
         word = k == 0 ? w[0] : (k == 1 ? w[1 % WORDS] : (k == 2 ? w[2 % WORDS] : w[3 % WORDS]));
      }
      return (uint32_t)((word >> (BITS * (e % PER_WORD))) & FIELD);
   }
};

// --------------------------------------------------------------------------
// one tile -> its map
// --------------------------------------------------------------------------

// elements [lo, lo + npos + L - 1) of the domain starting at byte `start`, as raw
// bytes.  The copy is dword aligned on the SOURCE: LDS byte (k + mis) <-> ROM byte
// start + lo*S + k, where mis (returned, wave uniform) is the source misalignment.
// All loads of a lane are issued before the first LDS store (one memory latency
// per tile instead of one per 256 bytes).
__device__ __forceinline__ int mm_stage_tile(const MmTileArgs &a, MmWaveLds &W, uint64_t start, int64_t lo, int npos, int lane)
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
   constexpr int NU = (int)(sizeof(W.tile) / 4 + 63) / 64;
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
   if ((a.skip_bloom >> (d & 63)) & 1) {
      const int n = (int)a.plan.n_skip;
      for (int k = 0; k < n; k++) {
         s = P.skip_diff[k] == d ? P.skip_val[k] : s;
      }
   }
   return s;
}

// x mod D for x < 2048 without a division
__device__ __forceinline__ uint32_t mm_modd(const MmTileArgs &a, uint32_t x)
{
   return x - ((x * a.inv_d) >> 16) * (a.plan.L - 1);
}

// map of positions [lo, lo + npos) of the domain at byte `start`; lo_mod = lo mod D.
// The result is wave uniform.
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

template <int BITS>
__device__ __forceinline__ void mm_tile_map(const MmTileArgs &a, const MmPlanLds &P, MmWaveLds &W, uint64_t start, int64_t lo,
                                            int npos, uint32_t lo_mod, int lane, MmPhaseMap<BITS> &M,
                                            bool *end_matches = nullptr)
{
   start = mm_uniform64(start);
   lo = (int64_t)mm_uniform64((uint64_t)lo);
   npos = (int)mm_uniform((uint32_t)npos);
   lo_mod = mm_uniform(lo_mod);
   const int mis = mm_stage_tile(a, W, start, lo, npos, lane);
   mm_wave_sync();

   const int L = (int)a.plan.L, S = (int)a.g.S;
   const uint32_t D = (uint32_t)(L - 1);
   const bool be = a.g.big_endian != 0;
   const uint8_t *tile = reinterpret_cast<const uint8_t *>(W.tile) + mis;
   // the first compare (keyword position L-1) from wave-uniform registers
   const int e1 = a.plan.expected[L - 1], b1 = a.plan.bridge[L - 1], w1 = a.plan.wst[L - 1];
   const uint32_t m1 = a.plan.cmp_mask[L - 1];
   const int match_jump = (int)a.plan.match_jump;

   M.identity();
   uint32_t ph0 = lo_mod;                                 // phase of position 64t (scalar)
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
      int J = (int)D;                                     // lanes past the end: no-op
      if (live) {
         const int s = mm_tile_skip(a, P, d);
         J = s < w1 ? s : w1;
      }
      if (__ballot(deep) != 0) {
         if (deep) {
            J = match_jump;
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
      // exceptional positions of this group, in position order, on the scalar unit
      unsigned long long mask = __ballot(J != (int)D);
      while (mask) {
         const int bit = __builtin_ctzll(mask);
         mask &= mask - 1;
         const uint32_t Jb = (uint32_t)__builtin_amdgcn_readlane(J, bit);
         const uint32_t r = mm_modd(a, ph0 + (uint32_t)bit);
         uint32_t r2 = r + Jb;
         r2 = r2 >= D ? r2 - D : r2;
         M.update(r, r2);
      }
      ph0 = mm_modd(a, ph0 + 64u);
      c = cn;
      pv = pn;
   }
   if (end_matches) {
      // position npos itself (the candidate in front of which this window ends); only
      // meaningful when the domain holds L elements from there, which the caller checked
      *end_matches = mm_tile_matches(a, P, tile, npos);
   }
   mm_wave_sync();                                        // the tile buffer may be restaged now
}

// A' = { e < D : M[e] in A }
template <int BITS>
__device__ __forceinline__ uint32_t mm_pull_back(uint32_t A, int D, const MmPhaseMap<BITS> &M)
{
   uint32_t r = 0;
   for (int e = 0; e < D; e++) {
      r |= ((A >> M.get(e)) & 1u) << e;
   }
   return r;
}

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
   // candidates the bounded look-back could not settle
   uint64_t *hard_off;                        // [MM_HARD_CAP] candidate byte offset
   uint64_t *hard_hi;                         // [MM_HARD_CAP] frontier (domain position, multiple of MM_TILE)
   uint32_t *hard_set;                        // [MM_HARD_CAP] acceptable phases at the frontier
   uint32_t *hard_slot;                       // [MM_HARD_CAP] the candidate's result slot
   unsigned int *hard_count;
};

constexpr uint64_t MM_NO_MATCH = ~0ull;

__device__ __forceinline__ uint64_t mm_report_value(const MmGeom &g, uint64_t o, uint64_t base_offset)
{
   return g.whole ? o / g.S : o + base_offset;
}

template <int BITS>
__device__ __forceinline__ void mm_resolve_body(const MmResolveArgs &a, const MmPlanLds &P, MmWaveLds &W)
{
   const int D = (int)a.t.plan.L - 1;
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   // the filter's MM_CAND_LISTS (= 64 = one per lane) lists get one compact numbering:
   // candidate ci is entry ci - excl[c] of the list c with excl[c] <= ci < excl[c+1]
   const unsigned long long my_count = a.list_count[lane * MM_LIST_STRIDE];
   unsigned long long incl = my_count;
#pragma unroll
   for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long v = __shfl_up(incl, d);
      incl += lane >= d ? v : 0ull;
   }
   const unsigned long long excl = incl - my_count;
   unsigned long long ncand = __shfl(incl, 63);
   if (__ballot(my_count > a.list_cap) != 0) {
      ncand = ~0ull;                           // a list overflowed
   }
   if (blockIdx.x == 0 && threadIdx.x == 0) {
      *a.total_out = ncand;
   }
   if (ncand > a.out_cap || ncand > a.max_candidates) {
      return;                                  // dense input: the host runs another engine
   }
   const uint64_t nwaves = (uint64_t)gridDim.x * MM_WAVES;
   const uint32_t full = D >= 32 ? 0xFFFFFFFFu : ((1u << D) - 1u);
   unsigned long long walked = 0;

   for (uint64_t ci = (uint64_t)blockIdx.x * MM_WAVES + wave; ci < ncand; ci += nwaves) {
      const int list = __popcll(__ballot(excl <= ci)) - 1;
      const uint64_t o = a.cand[(uint64_t)list * a.list_cap + (ci - __shfl(excl, list))];
      uint64_t b; uint32_t p; int64_t jc;
      if (!mm_locate(a.t.g, o, &b, &p, &jc)) {
         // SWAR survivor that is not an alignment of any domain (file tail, 16-bit odd boundary)
         if (lane == 0) {
            a.out[ci] = MM_NO_MATCH;
         }
         continue;
      }
      const uint64_t start = mm_domain_start(a.t.g, b, p);

      uint32_t A = 1u << (uint32_t)(jc % D);
      int64_t hi = jc;
      int verdict = -1;                        // 1 visited, 0 not visited, -1 undecided
      if (jc == 0) {
         // first alignment of its domain: always visited, but is it a match?  (no window to
         // stage in front of it: read the elements directly)
         bool matched;
         mm_step(a.t.plan, [&](int64_t k) { return mm_elem(a.t.g, start, k); }, 0, &matched);
         verdict = matched ? 1 : 0;
      }
      // Look-back windows grow 256 -> 2048 positions and end on multiples of their size:
      // a true match pulls every phase onto itself within a few keyword lengths (that is
      // what the bad-character rule is for), so the first short window usually settles
      // it; after the last step the frontier is tile aligned for mm_hard_resolve.
      for (int step = 0; step < MM_FAST_STEPS && hi > 0 && verdict < 0; step++) {
         const int64_t gran = (int64_t)(MM_TILE >> (MM_FAST_STEPS - 1 - step));
         const int64_t lo = ((hi - 1) / gran) * gran;
         MmPhaseMap<BITS> M;
         bool is_match = true;
         mm_tile_map<BITS>(a.t, P, W, start, lo, (int)(hi - lo), (uint32_t)(lo % D), lane, M, step == 0 ? &is_match : nullptr);
         walked++;
         if (!is_match) {
            verdict = 0;                        // survived the SWAR conditions only
            break;
         }
         A = mm_pull_back<BITS>(A, D, M);
         hi = lo;
         if (A == full || A == 0) {
            verdict = A ? 1 : 0;
            break;
         }
      }
      if (verdict < 0 && hi == 0) {
         verdict = (A & 1u) ? 1 : 0;           // domain start: the chain is in phase 0
      }
      if (lane == 0) {
         a.out[ci] = verdict == 1 ? mm_report_value(a.t.g, o, a.base_offset) : MM_NO_MATCH;
         if (verdict < 0) {
            unsigned int slot = atomicAdd(a.hard_count, 1u);
            if (slot < MM_HARD_CAP) {
               a.hard_off[slot] = o;
               a.hard_hi[slot] = (uint64_t)hi;
               a.hard_set[slot] = A;
               a.hard_slot[slot] = (uint32_t)ci;
            }
         }
      }
   }
   if (lane == 0 && walked) {
      atomicAdd(a.tiles_walked + (blockIdx.x % MM_STAT_STRIPES), walked);
   }
}

__global__ __launch_bounds__(64 * MM_WAVES) void mm_resolve(MmResolveArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLds Wv[MM_WAVES];
   mm_plan_to_lds(P, a.t.plan);
   if (a.t.plan.L - 1 <= 16) {
      mm_resolve_body<4>(a, P, Wv[threadIdx.x >> 6]);
   }
   else {
      mm_resolve_body<8>(a, P, Wv[threadIdx.x >> 6]);
   }
}

// --------------------------------------------------------------------------
// hard resolver: MM_HARD_PARTS workgroups map a candidate's whole prefix
// --------------------------------------------------------------------------
//
// grid = (MM_HARD_PARTS, MM_HARD_CAP).  The 4*PARTS waves of candidate i take the
// tiles of [0, hi) round-robin and write each tile map (32 B) to scratch; the
// workgroup that finishes last pulls the phase set back through them.

struct MmHardArgs {
   MmTileArgs t;
   const uint64_t *hard_off;
   const uint64_t *hard_hi;
   const uint32_t *hard_set;
   const uint32_t *hard_slot;
   const unsigned int *hard_count;
   unsigned int *done;                        // [MM_HARD_CAP] arrival tickets (zeroed per scan)
   unsigned int *overflow;                    // set when a prefix is too long: the host runs another engine
   uint8_t *scratch;                          // [MM_HARD_CAP][MM_HARD_MAX_TILES][MM_MAXD]
   uint64_t *out;                             // result slots (see MmResolveArgs)
   unsigned long long *tiles_walked;
   uint64_t base_offset;
};

template <int BITS>
__device__ __forceinline__ void mm_hard_tiles(const MmHardArgs &a, const MmPlanLds &P, MmWaveLds &W, uint64_t start,
                                              uint64_t ntiles, uint8_t *maps)
{
   const int D = (int)a.t.plan.L - 1;
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   unsigned long long walked = 0;
   for (uint64_t t = (uint64_t)blockIdx.x * MM_WAVES + wave; t < ntiles; t += MM_HARD_PARTS * MM_WAVES) {
      MmPhaseMap<BITS> M;
      const int64_t lo = (int64_t)(t * MM_TILE);
      mm_tile_map<BITS>(a.t, P, W, start, lo, MM_TILE, (uint32_t)(lo % D), lane, M);
      if (lane < MM_MAXD) {
         maps[t * MM_MAXD + lane] = (uint8_t)M.get(lane);
      }
      walked++;
   }
   if (lane == 0 && walked) {
      atomicAdd(a.tiles_walked + (blockIdx.x % MM_STAT_STRIPES), walked);
   }
}

__global__ __launch_bounds__(64 * MM_WAVES) void mm_hard_resolve(MmHardArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLds Wv[MM_WAVES];
   __shared__ uint8_t pull[256][MM_MAXD];
   __shared__ int is_last;
   __shared__ uint32_t sh_set;
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
   mm_locate(a.t.g, o, &b, &p, &jc);
   const uint64_t start = mm_domain_start(a.t.g, b, p);
   uint8_t *maps = a.scratch + (uint64_t)i * MM_HARD_MAX_TILES * MM_MAXD;

   if (D <= 16) {
      mm_hard_tiles<4>(a, P, Wv[wave], start, ntiles, maps);
   }
   else {
      mm_hard_tiles<8>(a, P, Wv[wave], start, ntiles, maps);
   }

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

   const uint32_t full = D >= 32 ? 0xFFFFFFFFu : ((1u << D) - 1u);
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
            sh_verdict = (sh_set & 1u) ? 1 : 0;           // domain start: the chain is in phase 0
         }
         continue;
      }
      const int64_t t_lo = t_hi > 256 ? t_hi - 256 : 0;
      const int n = (int)(t_hi - t_lo);
      {
         // volatile: these bytes were written by other workgroups during this launch
         const volatile uint32_t *src = reinterpret_cast<const volatile uint32_t *>(maps + t_lo * MM_MAXD);
         uint32_t *dst = reinterpret_cast<uint32_t *>(&pull[0][0]);
         for (int k = threadIdx.x; k < n * (MM_MAXD / 4); k += blockDim.x) {
            dst[k] = src[k];
         }
      }
      __syncthreads();
      if (wave == 0) {
         uint32_t A = sh_set;
         for (int k = n - 1; k >= 0; k--) {
            const int v = pull[k][lane & (MM_MAXD - 1)];
            A = (uint32_t)__ballot(lane < D && ((A >> v) & 1u));
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
