// SPDX-License-Identifier: GPL-3.0-or-later
// mm_tail2.h -- the tail of a big-ROM scan over the BUCKETED candidate store (mm_internal.h MM_BUCKET_*).
// Included by mm_kernels.hip after mm_fused.h (device code only).
//
// mm_scan_tail (mm_fused.h) ranks a candidate by comparing it with every other candidate: each workgroup reads the
// whole candidate set once per round -- 34 KiB at the bench ROM's 4 K candidates, fine; 512 KiB per workgroup and
// round at 64 K, where the ordering had to be left to a radix sort behind the scan and the scan fell below 0.6 of the
// HBM roofline (profiles/r03_candidate_density.log, "before").  Here the streaming kernel has already dropped every
// candidate into the bucket of its ROM neighbourhood, buckets are in offset order, and a wave needs three loads to
// know a candidate and its place in the ascending list, whatever the number of candidates:
//     the 64 member counts of the candidate's super-bucket   -> which bucket, which slot, candidates in front of the bucket
//     the bucket's members (one per lane and round)          -> the candidate itself and how many members are smaller
// (the candidates in front of the super-bucket: every workgroup sums the 4096 bucket counters once, 16 KiB from L2).
// No workgroup-wide step is left inside the candidate loop: waves run on their own, with 20 .. 40 registers fewer
// than mm_scan_tail (one candidate per wave: 64; the grouped variants below: 80 / 96) -- the kernel fits beside the
// streaming kernel of the NEXT scan (scans in flight).
//
// Everything else is mm_scan_tail's: the reference's compare loop + chain membership through the two look-back windows
// (mm_resolve_candidate), the verdict stored in slot `rank` of the published block, left-overs handed to the second
// phase, the last workgroup writes the header, zeroes the control block and the bucket counters and raises the flag.
// Lists of more than MM_DIRECT_PUBLISH slots are only written to the device-side copy of the block (header word 4,
// MM_HDR_ON_DEVICE): the host fetches them with one copy.  A bucket that overflowed (MM_BUCKET_CAP members: a flood),
// or more candidates than the block holds: nothing is resolved, header word 4 carries no MM_HDR_SPARSE and the host
// runs the list-based path.
#ifndef MM_TAIL2_H
#define MM_TAIL2_H

struct MmTail2Lds {
   unsigned int super_excl[MM_MAX_BUCKETS / MM_SUPER + 1];   // candidates in front of every super-bucket; [nsuper] = all
   unsigned int holes, walked;
   int last_block;
   uint32_t flood_lo[MM_WAVES], flood_hi[MM_WAVES];          // first / last thread (= 16 buckets) with an overflowing bucket, per wave
};

// ---- several candidates per wave (round 4) --------------------------------------------------------------------------
// The tail is bound by the latency of a wave's chain per candidate, not by instruction issue (profiles/
// r04_tail_kernel_counters.txt: 0.26 instructions per cycle and SIMD; the cold loads are a third of the time, the
// instructions + LDS chain of the window 40 %).  A true match settles within two keyword lengths in front of it, so a
// window of 32 positions is enough for keywords of up to 16 symbols -- and then HALF a wave is enough for a candidate
// (a quarter and 16 positions for keywords of up to 13, an eighth and 8 positions up to 4): SUBW lanes per candidate,
// 64 / SUBW candidates per wave, one instruction stream, their cold loads in flight at once (profiles/
// r04_tail_kernel_counters_grouped.txt: 400 wave instructions per candidate instead of 1170).  What a part of the wave cannot settle (its SUBW positions leave
// the phase set mixed, the domain's first alignments, survivors that are no alignment) is left to the caller, who resolves
// that candidate the old way: verdict MM_SUB_AGAIN.
constexpr int MM_SUB_AGAIN = -3;

// The verdicts (1 reported, 0 not, MM_SUB_AGAIN) of the wave's 64 / SUBW candidates in v[] (wave uniform).  o: the
// candidate of this lane's part of the wave; have: there is one.
template <int SUBW, class A>
__device__ __forceinline__ void mm_resolve_sub(const A &a, const MmPlanLds &P, MmWaveLdsShort &W, uint64_t o, bool have, int lane,
                                               int (&verdicts)[64 / SUBW])
{
   constexpr int NSUB = 64 / SUBW;
   constexpr int STRIDE = SUBW + 8;                                       // dwords of tile / bytes of jumps per part
   static_assert(sizeof(W.tile) >= NSUB * STRIDE * 4 && sizeof(W.jump) >= NSUB * STRIDE, "the parts share the wave's LDS");
   const uint32_t D = a.t.plan.L - 1;
   const uint32_t full = (1u << D) - 1u;                                  // (D <= 15 here)
   const int L = (int)a.t.plan.L, S = (int)a.t.g.S;
   const bool be = a.t.g.big_endian != 0;
   const int part = lane / SUBW, hl = lane % SUBW;
   constexpr int npos = SUBW;
   uint64_t b = 0; uint32_t p = 0; int64_t jc = 0;
   const bool mine = have && mm_locate_fast(a.t, o, &b, &p, &jc) && jc >= npos;            // (else: the old way)
   const uint64_t start = mm_domain_start(a.t.g, b, p);
   const int64_t lo = jc - npos;
   const uint32_t lo_mod = mine ? mm_modd64(a.t, (uint64_t)lo) : 0u;
   // stage positions [lo, jc] + the keyword's reach: LDS byte (k + mis) of the part's tile <-> ROM byte src0 + k
   const uint64_t src0 = start + (uint64_t)(mine ? lo : 0) * S;
   int nstage = (npos + L) * S;                                           // (+ 6 <= 4 SUBW: L <= 29 / 13 / 5 for 32 / 16 / 8 lanes)
   if (mine && src0 + (uint64_t)nstage > a.t.g.nbytes) {
      nstage = (int)(a.t.g.nbytes - src0);
   }
   const int mis = (int)(((uintptr_t)(a.t.g.rom + src0)) & 3);
   const int nw = (nstage + mis + 3) >> 2;                                // <= SUBW dwords
   uint32_t *part_tile = W.tile + STRIDE * part;
   uint32_t v0 = 0;
   if (mine && hl < nw) {
      v0 = reinterpret_cast<const uint32_t *>(a.t.g.rom + src0 - mis)[hl];
   }
   part_tile[hl] = v0;
   mm_wave_sync();
   const uint8_t *tile = reinterpret_cast<const uint8_t *>(part_tile) + mis;
   uint8_t *jump = W.jump + STRIDE * part;
   // the jump of every position of the window (one lane each), as mm_tile_jumps computes it
   {
      const int e1 = a.t.plan.expected[L - 1], b1 = a.t.plan.bridge[L - 1], w1 = a.t.plan.wst[L - 1];
      const uint32_t m1 = a.t.plan.cmp_mask[L - 1];
      const int q = hl;
      int J = 1;
      bool deep = false;
      if (mine) {
         const int c = mm_tile_elem(tile, q + L - 1, S, be);
         const int pv = mm_tile_elem(tile, q + L - 1 + b1, S, be);
         const int d = c - pv;
         deep = ((uint32_t)(d ^ e1) & m1) == 0;
         const int s = mm_tile_skip(a.t, P, d);
         J = s < w1 ? s : w1;
      }
      if (__ballot(deep) != 0) {
         if (deep) {
            J = (int)a.t.plan.match_jump | MM_JUMP_MATCH;
            for (int i = L - 2; i >= 0; --i) {
               const int ci = mm_tile_elem(tile, q + i, S, be);
               const int pi = mm_tile_elem(tile, q + i + P.bridge[i], S, be);
               const int di = ci - pi;
               if (((uint32_t)(di ^ P.expected[i]) & P.cmp_mask[i]) != 0) {
                  const int s = mm_tile_skip(a.t, P, di);
                  const int w = P.wst[i];
                  J = s < w ? s : w;
                  break;
               }
            }
         }
      }
      jump[q] = (uint8_t)((J & (MM_JUMP_MATCH - 1)) == 0 ? 1 : J);
   }
   mm_wave_sync();
   // lane hl < D follows the chain of entry phase hl through the part's jumps
   uint32_t v = (uint32_t)hl;
   if (mine && (uint32_t)hl < D) {
      uint32_t q = (uint32_t)hl + D - lo_mod;
      q = q >= D ? q - D : q;
      while (q < (uint32_t)npos) {
         q += jump[q] & (MM_JUMP_MATCH - 1);
      }
      v = mm_modd(a.t, lo_mod + (uint32_t)npos) + (q - (uint32_t)npos);
      v = v >= D ? v - D : v;
   }
   // does the compare loop match AT the candidate?  (every lane of the part works it out)
   const bool is_match = mine && mm_tile_matches(a.t, P, tile, npos);
   const uint32_t Sset = mine ? 1u << mm_modd64(a.t, (uint64_t)jc) : 0u;
   const unsigned long long pulled = __ballot(mine && (uint32_t)hl < D && ((Sset >> v) & 1u) != 0);
   const unsigned long long mine_mask = __ballot(mine), match_mask = __ballot(is_match);
   mm_wave_sync();                                                        // the wave's LDS may be restaged now
#pragma unroll
   for (int h = 0; h < NSUB; h++) {
      const uint32_t set = (uint32_t)(pulled >> (SUBW * h)) & (uint32_t)((1ull << SUBW) - 1ull);
      int verdict = MM_SUB_AGAIN;
      if ((mine_mask >> (SUBW * h)) & 1ull) {
         if (!((match_mask >> (SUBW * h)) & 1ull)) {
            verdict = 0;                                                   // survived the SWAR conditions only
         }
         else if (set == full) {
            verdict = 1;
         }
         else if (set == 0) {
            verdict = 0;
         }
      }
      verdicts[h] = verdict;
   }
}

template <int OCC, int SUBW, bool LONG = false>
__global__ __launch_bounds__(64 * MM_WAVES, OCC) void mm_scan_tail2(MmFusedArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLdsShort Wv[MM_WAVES];
   __shared__ MmTail2Lds T;
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   constexpr uint32_t nsuper = MM_MAX_BUCKETS / MM_SUPER;                 // 64: one per lane

   // ---- candidates in front of every super-bucket ------------------------------------------------------------------
   // Thread t sums 16 bucket counters (4 x 16 bytes, L2 resident: the streaming kernel's atomics left them there), four
   // neighbouring threads make a super-bucket's sum, wave 0 scans the 64 sums while the plan goes to LDS.
   static_assert(MM_MAX_BUCKETS == 64 * MM_SUPER && MM_SUPER == 64, "one super-bucket per lane, 16 counters per thread");
   {
      const uint4 *row = reinterpret_cast<const uint4 *>(a.bcount) + (uint64_t)threadIdx.x * 4;
      unsigned int sum = 0;
      bool flooded = false;                                               // one of this thread's 16 buckets holds more than its store
#pragma unroll
      for (int k = 0; k < 4; k++) {
         const uint4 v = row[k];                                          // (buckets behind the ROM's last one were never touched: zero)
         sum += v.x + v.y + v.z + v.w;
         flooded = flooded || v.x > MM_BUCKET_CAP || v.y > MM_BUCKET_CAP || v.z > MM_BUCKET_CAP || v.w > MM_BUCKET_CAP;
      }
      sum += (unsigned int)__shfl_xor((int)sum, 1);
      sum += (unsigned int)__shfl_xor((int)sum, 2);
      if ((threadIdx.x & 3) == 0) {
         T.super_excl[(threadIdx.x >> 2) + 1] = sum;
      }
      // WHERE a flood is (round 6): the first and the last thread with an overflowing bucket, i.e. the flood's extent in
      // units of 16 buckets -- the host sends the forward engine there and nowhere else (mm_capi_split.h)
      const unsigned long long fl = __ballot(flooded);
      if (lane == 0) {
         T.flood_lo[wave] = fl ? (uint32_t)(64 * wave + __builtin_ctzll(fl)) : 0xFFFFu;
         T.flood_hi[wave] = fl ? (uint32_t)(64 * wave + 63 - __builtin_clzll(fl)) : 0u;
      }
   }
   if (threadIdx.x == 0) {
      T.super_excl[0] = 0;
      T.holes = 0;
      T.walked = 0;
   }
   __syncthreads();
   if (threadIdx.x < 64) {
      unsigned int incl = T.super_excl[lane + 1];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
         const unsigned int up = (unsigned int)__shfl_up((int)incl, d);
         incl += lane >= d ? up : 0u;
      }
      T.super_excl[lane + 1] = incl;                                      // candidates up to and including super-bucket `lane`
   }
   mm_plan_to_lds(P, a.t.plan);                                           // ends with a __syncthreads()
   const uint64_t ncand = T.super_excl[nsuper];
   const bool overflow = __hip_atomic_load(a.boverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
   const bool resolvable = !overflow && ncand <= a.out_cap && ncand <= a.max_candidates && ncand <= a.max_rank;
   const bool direct = ncand <= a.direct_limit;                           // slots go to pinned host memory as well
   unsigned long long walked = 0;
   unsigned int holes = 0;
   const uint64_t nwaves_all = (uint64_t)gridDim.x * MM_WAVES;
   bool grouped = false;
   if constexpr (SUBW < 64) {
   if (resolvable && ncand > a.group_min) {
      grouped = true;
      // Waves take 64 / SUBW NEIGHBOURS of the compact numbering: mostly members of one bucket, so one walk to the bucket
      // and one pass over its members serve all of them.
      constexpr int NSUB = 64 / SUBW;
      constexpr int SW = SUBW;
      const uint64_t nwaves = nwaves_all;
      auto super_of = [&](uint64_t ci) { return 63u - (uint32_t)__builtin_clzll(__ballot(T.super_excl[lane] <= ci)); };
      uint64_t ci = ((uint64_t)blockIdx.x * MM_WAVES + wave) * NSUB;
      uint32_t s_next = ci < ncand ? super_of(ci) : 0u;
      unsigned int n_next = ci < ncand ? a.bcount[s_next * MM_SUPER + (uint32_t)lane] : 0u;
      for (; ci < ncand; ci += nwaves * NSUB) {
         const uint32_t s0 = s_next;
         const unsigned int n0 = n_next;
         if (ci + nwaves * NSUB < ncand) {
            s_next = super_of(ci + nwaves * NSUB);
            n_next = a.bcount[s_next * MM_SUPER + (uint32_t)lane];
         }
         uint64_t o[NSUB];
         uint32_t rank[NSUB];
         bool have[NSUB], placed[NSUB];
#pragma unroll
         for (int k = 0; k < NSUB; k++) {
            have[k] = ci + k < ncand;
            placed[k] = false;
            o[k] = 0;
            rank[k] = 0;
         }
         // candidate number cn: the bucket it lives in, then it and the neighbours of the group that share the bucket
         auto lookup = [&](int k_first, uint32_t s, unsigned int n_lane) {
            const uint64_t cn = ci + k_first;
            const uint32_t in_super = (uint32_t)(cn - T.super_excl[s]);
            unsigned int incl = n_lane;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
               const unsigned int up = (unsigned int)__shfl_up((int)incl, d);
               incl += lane >= d ? up : 0u;
            }
            const unsigned int excl = incl - n_lane;
            const unsigned long long starts = __ballot(excl <= in_super && n_lane != 0);
            const int bl = starts ? 63 - __builtin_clzll(starts) : 0;
            const uint32_t bucket = s * MM_SUPER + (uint32_t)bl;
            const uint32_t before = (uint32_t)T.super_excl[s] + (uint32_t)__shfl((int)excl, bl);
            const uint32_t members = (uint32_t)__shfl((int)n_lane, bl);
            const uint32_t slot = in_super - (uint32_t)__shfl((int)excl, bl);
            const unsigned long long *mem = reinterpret_cast<const unsigned long long *>(a.bcand) + (uint64_t)bucket * MM_BUCKET_CAP;
            // (numbers cn, cn + 1, ... are slots slot, slot + 1, ... of this bucket -- in arrival order; each gets its place in
            // the ascending list by counting the members in front of it, as ever)
#pragma unroll
            for (int k = 0; k < NSUB; k++) {
               if (k >= k_first && have[k] && !placed[k] && slot + (uint32_t)(k - k_first) < members) {
                  o[k] = mm_uniform64(mem[slot + (uint32_t)(k - k_first)]);
                  rank[k] = before;
               }
            }
            for (uint32_t k0 = 0; k0 < members; k0 += 64) {
               const uint32_t kk = k0 + (uint32_t)lane;
               const unsigned long long mk = kk < members ? mem[kk] : ~0ull;
#pragma unroll
               for (int k = 0; k < NSUB; k++) {
                  if (k >= k_first && have[k] && !placed[k] && slot + (uint32_t)(k - k_first) < members) {
                     rank[k] += (uint32_t)__popcll(__ballot(mk < o[k]));
                  }
               }
            }
#pragma unroll
            for (int k = 0; k < NSUB; k++) {
               if (k >= k_first && have[k] && !placed[k] && slot + (uint32_t)(k - k_first) < members) {
                  placed[k] = true;
               }
            }
         };
         lookup(0, s0, n0);
#pragma unroll
         for (int k = 1; k < NSUB; k++) {
            if (have[k] && !placed[k]) {                                  // (the group straddles buckets: seldom)
               const uint32_t sk = super_of(ci + k);
               lookup(k, sk, a.bcount[sk * MM_SUPER + (uint32_t)lane]);
            }
         }
         uint64_t o_lane = o[0];
         bool have_lane = have[0];
#pragma unroll
         for (int k = 1; k < NSUB; k++) {
            o_lane = lane / SW == k ? o[k] : o_lane;
            have_lane = lane / SW == k ? have[k] : have_lane;
         }
         int verdicts[NSUB];
         mm_resolve_sub<SW>(a, P, Wv[wave], o_lane, have_lane, lane, verdicts);
#pragma unroll
         for (int k = 0; k < NSUB; k++) {
            if (!have[k]) {
               break;
            }
            walked++;
            int verdict = verdicts[k];
            int64_t hi = 0; mm_set_t set = 0; uint64_t dom = 0;
            if (verdict == MM_SUB_AGAIN) {
               verdict = mm_resolve_candidate(a, P, Wv[wave], o[k], lane, &walked, &hi, &set, &dom);
            }
            if (lane == 0) {
               const uint64_t value = verdict == 1 ? mm_report_value(a.t.g, o[k], a.base_offset) : MM_NO_MATCH;
               a.out[ci + k] = value;
               if (direct) {
                  __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.host_result) + MM_RESULT_HEADER_WORDS + rank[k],
                                     (unsigned long long)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
               }
               a.dev_result[MM_RESULT_HEADER_WORDS + rank[k]] = value;
               holes += verdict != 1 ? 1u : 0u;
               if (verdict == -1) {
                  mm_resolve_hand_over(a, o[k], ci + k, hi, set, dom);
               }
            }
         }
      }
   }
   }
   if (resolvable && !grouped) {
      const uint64_t nwaves = (uint64_t)gridDim.x * MM_WAVES;
      // (the 64 bucket counters of the NEXT candidate's super-bucket are fetched while this one is resolved: one of the
      // three dependent loads in front of a candidate off the critical path)
      auto super_of = [&](uint64_t ci) { return 63u - (uint32_t)__builtin_clzll(__ballot(T.super_excl[lane] <= ci)); };
      // (candidate ci, ci + nwaves, ...: round 4 measured RUNS of consecutive candidates per wave instead -- neighbours in the
      // ROM, three of a candidate's four dependent loads from lines just touched -- and it was slower, `water` 1.01 ->
      // 1.03 ms, `th*s` 1.69 -> 1.92: what a run wins in cache hits it loses to the waves that draw a run of expensive
      // candidates; striding spreads those.  The grouped loop above does take neighbours -- 2 to 8 of them, side by side in
      // one wave, sharing the walk to their bucket: a different trade)
      uint64_t ci = (uint64_t)blockIdx.x * MM_WAVES + wave;
      uint32_t s = ci < ncand ? super_of(ci) : 0u;
      unsigned int n_next = ci < ncand ? a.bcount[s * MM_SUPER + (uint32_t)lane] : 0u;
      for (; ci < ncand; ci += nwaves) {
         const uint32_t in_super = (uint32_t)(ci - T.super_excl[s]);
         const uint32_t s_this = s;
         const unsigned int n_lane = n_next;
         if (ci + nwaves < ncand) {
            s = super_of(ci + nwaves);
            n_next = a.bcount[s * MM_SUPER + (uint32_t)lane];
         }
         unsigned int incl = n_lane;
#pragma unroll
         for (int d = 1; d < 64; d <<= 1) {
            const unsigned int up = (unsigned int)__shfl_up((int)incl, d);
            incl += lane >= d ? up : 0u;
         }
         const unsigned int excl = incl - n_lane;
         // (the last non-empty bucket that starts at or in front of the candidate; ci < ncand: there is one)
         const unsigned long long starts = __ballot(excl <= in_super && n_lane != 0);
         const int bl = starts ? 63 - __builtin_clzll(starts) : 0;
         const uint32_t bucket = s_this * MM_SUPER + (uint32_t)bl;
         const uint32_t before = (uint32_t)T.super_excl[s_this] + (uint32_t)__shfl((int)excl, bl);   // candidates in front of the bucket
         const uint32_t members = (uint32_t)__shfl((int)n_lane, bl);
         const uint32_t slot = in_super - (uint32_t)__shfl((int)excl, bl);
         // the candidate itself, and how many of the bucket's members lie in front of it (64 members per round; one round
         // unless the neighbourhood is crowded)
         const unsigned long long *mem = reinterpret_cast<const unsigned long long *>(a.bcand) + (uint64_t)bucket * MM_BUCKET_CAP;
         const uint64_t o = mm_uniform64(mem[slot]);
         uint32_t rank = before;
         for (uint32_t k0 = 0; k0 < members; k0 += 64) {
            const uint32_t k = k0 + (uint32_t)lane;
            const unsigned long long mk = k < members ? mem[k] : ~0ull;
            rank += (uint32_t)__popcll(__ballot(mk < o));
         }
         int64_t hi = 0; mm_set_t set = 0; uint64_t dom = 0;
         const int verdict = mm_resolve_any<LONG>(a, P, Wv[wave], o, lane, &walked, &hi, &set, &dom);
         if (lane == 0) {
            const uint64_t value = verdict == 1 ? mm_report_value(a.t.g, o, a.base_offset) : MM_NO_MATCH;
            a.out[ci] = value;
            if (direct) {
               // system-scope store: written through to host memory, complete once this wave's vmcnt drains
               __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.host_result) + MM_RESULT_HEADER_WORDS + rank,
                                  (unsigned long long)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            a.dev_result[MM_RESULT_HEADER_WORDS + rank] = value;
            holes += verdict != 1 ? 1u : 0u;
            if (verdict == -1) {
               mm_resolve_hand_over(a, o, ci, hi, set, dom);
            }
         }
      }
   }
   // ---- statistics, header, flag (as mm_scan_tail_phase) ------------------------------------------------------------
   if (lane == 0) {
      if (walked) {
         atomicAdd(&T.walked, (unsigned int)walked);
      }
      if (holes) {
         atomicAdd(&T.holes, holes);
      }
   }
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's result stores have arrived
   __syncthreads();
   if (threadIdx.x == 0) {
      if (T.walked) {
         atomicAdd(a.tiles_walked + (blockIdx.x % MM_STAT_STRIPES), (unsigned long long)T.walked);
      }
      if (T.holes) {
         atomicAdd(a.ctrl + MM_CTRL_NOMATCH, (unsigned long long)T.holes);
      }
      // The statistics must have been PERFORMED before this workgroup counts as arrived: the two atomics above return
      // nothing and go to other L2 channels than the arrival counter -- nothing orders them in front of the arrival,
      // and the last workgroup reads the hole count right behind its own.  Found by the validation of round 4 (one
      // scan in ~300 000 of the fuzz published "matches + 1" a few holes too high: the host then delivered stale
      // slots behind the compacted list).  One L2 round trip per workgroup, off the candidates' path.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      T.last_block = mm_arrive_last(a.ctrl + MM_CTRL_ARRIVE_END, gridDim.x);
   }
   __syncthreads();
   if (!T.last_block) {
      return;
   }
   if (threadIdx.x < 64) {
      unsigned long long v = 0;
      if (lane < 32) {
         v = __hip_atomic_load(a.ctrl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      unsigned long long tiles = lane >= MM_CTRL_TILES && lane < MM_CTRL_TILES + MM_STAT_STRIPES ? v : 0;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
         tiles += __shfl_xor(tiles, d);
      }
      const unsigned long long nomatch = __shfl(v, MM_CTRL_NOMATCH), mid = __shfl(v, MM_CTRL_MID);
      unsigned long long h = 0;
      switch (lane) {
      case 0: h = overflow ? ~0ull : ncand; break;         // candidates = slots (~0: a bucket overflowed)
      case 1: {                                            // a flood's extent: bit 32 | first | last << 16, in units of 16 buckets
         uint32_t lo = 0xFFFFu, hi = 0u;
#pragma unroll
         for (int w = 0; w < MM_WAVES; w++) {
            lo = T.flood_lo[w] < lo ? T.flood_lo[w] : lo;
            hi = T.flood_hi[w] > hi ? T.flood_hi[w] : hi;
         }
         h = overflow && lo != 0xFFFFu ? (1ull << 32) | lo | ((unsigned long long)hi << 16) : 0;
         break;
      }
      case 2: h = tiles; break;
      case 4: h = (resolvable ? MM_HDR_SPARSE : 0) | (resolvable && !direct ? MM_HDR_ON_DEVICE : 0); break;
      case 5: h = mid; break;
      case 6: h = resolvable ? ncand - nomatch + 1 : 0; break;   // matches + 1 (0: not ordered here)
      default: break;
      }
      if (lane < (int)MM_RESULT_HEADER_WORDS) {
         a.host_result[lane] = h;
         a.dev_result[lane] = h;
      }
      if (lane == 0) {
         T.walked = (unsigned int)(mid & 0xFFFFFFFFull);   // (for the decision below)
      }
   }
   __syncthreads();
   // The control block and the bucket counters go back to zero for the next scan.  Left-overs: the second phase
   // needs the hand-over counters and MM_CTRL_TOTAL (the rank kernels order a.out[0 .. ncand)); it never looks at
   // the buckets.  Not resolvable: the host starts over with the list-based kernels (they zero what they need).
   const bool keep = resolvable && T.walked != 0;
   if (keep) {
      if (threadIdx.x == 0) {
         a.ctrl[MM_CTRL_NOMATCH] = 0;
         a.ctrl[MM_CTRL_BOVERFLOW] = 0;
         a.ctrl[MM_CTRL_TOTAL] = ncand;
      }
      for (uint32_t k = MM_CTRL_ARRIVE_BARRIER + threadIdx.x; k < MM_CTRL_WORDS; k += blockDim.x) {
         a.ctrl[k] = 0;
      }
   }
   else {
      for (uint32_t k = threadIdx.x; k < a.ctrl_words; k += blockDim.x) {
         a.ctrl[k] = 0;
      }
   }
   for (uint32_t k = threadIdx.x; k < a.nbuckets; k += blockDim.x) {
      a.bcount[k] = 0;
   }
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
   __syncthreads();
   if (threadIdx.x == 0) {
      __threadfence_system();
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.host_result) + MM_HDR_FLAG_WORD, (unsigned long long)a.seq,
                         __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
   }
}

// (Round 4 tried a kernel behind this one that copied long lists to pinned memory with system-scope stores and raised the
// flag -- instead of the host's DMA copy once it has seen the flag: worth 10 us of a 1 ms scan at 90 K candidates, and as
// a no-op behind every other scan it sat 4-16 us on the scan's stream (profiles/r04 interim kernel stats): taken out
// again.  What it taught stays in the code: a plain store to pinned memory may stay dirty in the L2 of the XCD that issued
// it, and one workgroup's __threadfence_system() only writes back its own XCD's L2 -- slots meant for the host are stored
// at system scope.)

#endif
