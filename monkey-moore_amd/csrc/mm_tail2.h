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
// than mm_scan_tail -- the kernel fits beside the streaming kernel of the NEXT scan (scans in flight).
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
};

template <int OCC>
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
#pragma unroll
      for (int k = 0; k < 4; k++) {
         const uint4 v = row[k];                                          // (buckets behind the ROM's last one were never touched: zero)
         sum += v.x + v.y + v.z + v.w;
      }
      sum += (unsigned int)__shfl_xor((int)sum, 1);
      sum += (unsigned int)__shfl_xor((int)sum, 2);
      if ((threadIdx.x & 3) == 0) {
         T.super_excl[(threadIdx.x >> 2) + 1] = sum;
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
   if (resolvable) {
      const uint64_t nwaves = (uint64_t)gridDim.x * MM_WAVES;
      // (the 64 bucket counters of the NEXT candidate's super-bucket are fetched while this one is resolved: one of the
      // three dependent loads in front of a candidate off the critical path)
      auto super_of = [&](uint64_t ci) { return 63u - (uint32_t)__builtin_clzll(__ballot(T.super_excl[lane] <= ci)); };
      // (candidate ci, ci + nwaves, ...: round 4 measured RUNS of consecutive candidates per wave instead -- neighbours in the
      // ROM, three of a candidate's four dependent loads from lines just touched -- and it was slower, `water` 1.01 ->
      // 1.03 ms, `th*s` 1.69 -> 1.92: what a run wins in cache hits it loses to the waves that draw a run of expensive
      // candidates; striding spreads those)
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
         int64_t hi = 0; uint32_t set = 0; uint64_t dom = 0;
         const int verdict = mm_resolve_candidate(a, P, Wv[wave], o, lane, &walked, &hi, &set, &dom);
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
