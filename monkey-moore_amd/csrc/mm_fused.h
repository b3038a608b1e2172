// SPDX-License-Identifier: GPL-3.0-or-later
// mm_fused.h -- the tail of a scan in one place, and small scans in ONE launch.  Included by
// mm_kernels.hip after mm_tiles.h (device code only).
//
// Round 1 ran three dependent launches behind the streaming kernel (mm_resolve, mm_rank_count,
// mm_rank_scatter: 32 us on the bench ROM, most of it launch and dependency latency) and the host
// then waited on an event (~12 us).  Now:
//   * mm_scan_tail_phase: wave w takes candidate w -- the reference's compare loop + chain
//     membership through the two short look-back windows (mm_resolve_candidate, the code of
//     mm_resolve) AND its rank among all candidates (offsets are unique: rank = number of smaller
//     ones; the workgroup's threads share out the candidate set and hold it in registers, the loads
//     are in flight while the resolver runs).  The ordering needs no verdicts, so nothing waits for
//     anything: the result goes straight to slot `rank` of the published block (pinned host memory
//     + its device-side copy): the reported value, or a hole (~0) for a candidate that is not
//     reported.  Holes are rare (bench ROM: none); the host drops them while copying out.  The
//     last workgroup to finish (two-level arrival counter) writes the header, zeroes the control
//     block for the next scan and raises the scan's sequence number in pinned host memory -- the
//     host polls that word instead of waiting for a HIP event.
//   * mm_scan_tail: that phase as ONE kernel behind mm_filter_* (large ROMs).
//   * mm_scan_fused: streaming filter, grid barrier and that phase in one launch, for ROMs of a
//     few MiB, where launch latency is all there is (a 128 KiB scan: 11 us on the device instead
//     of 20).  On big ROMs it does not pay: a thousand workgroups' arrivals and cache-bypassing
//     reads of the candidate lists (written in the same launch, cdna guideline 16) cost more than
//     the launch they save (measured: +35 us on 4 GiB).
// Candidates the two windows cannot settle are handed to the second phase exactly as mm_resolve
// does (mid list; the host launches mm_resolve2 / mm_hard_resolve and the rank kernels then).
//
// The fused kernel's grid barrier needs every workgroup resident at once: the grid is small and
// sized well below the device's capacity (host side, launch_fused), every spin is bounded by a
// wall-clock timeout, and a timeout makes the kernel give up cleanly (header word 4 bit 1) -- the
// host then runs the plain kernels on the candidate lists, which are complete by then.  Two fused
// kernels at once could starve each other of slots, so the host never has more than one in
// flight per process (a process-wide try-lock; the loser takes the two-launch path).
#ifndef MM_FUSED_H
#define MM_FUSED_H

// header word 4 of the published block
constexpr unsigned long long MM_HDR_SPARSE = 1;     // the list holds one slot per candidate, ~0 = hole
constexpr unsigned long long MM_HDR_GAVE_UP = 2;    // a barrier timed out: nothing was resolved
constexpr unsigned long long MM_HDR_ON_DEVICE = 4;  // the slots were only written to the device-side copy (long lists): the host fetches them

struct MmFusedArgs {
   MmTileArgs t;
   // the streaming code's members (names as in MmFilterArgs)
   uint32_t pat[4];
   uint32_t sh[4];
   uint32_t iA;
   uint32_t ncond;
   uint32_t verify;
   uint64_t *cand;
   unsigned long long *list_count;
   uint64_t list_cap;
   unsigned int *dom_count;            // always null here
   const uint32_t *skip_bits;          // always null here
   uint32_t *loud_bits;                // always null here
   uint32_t loud_tpd, loud_tile;
   uint64_t *bcand;                    // bucketed store (mm_tail2.h); null in the single-launch kernel
   unsigned int *bcount;
   unsigned long long *boverflow;
   uint32_t bshift;
   uint32_t nbuckets;
   uint32_t direct_limit;              // mm_scan_tail2: lists of up to this many slots are also stored straight into pinned host memory
   uint64_t ngroups;
   uint32_t groups_per_span;
   uint64_t edge_first;
   // the resolver's members (names as in MmResolveArgs)
   uint64_t out_cap;
   uint64_t *out;                      // result slot of candidate ci (for the second phase and its ordering)
   unsigned long long *tiles_walked;
   uint64_t base_offset;
   uint32_t max_candidates;
   uint64_t *mid_off;
   uint64_t *mid_hi;
   mm_set_t *mid_set;
   uint32_t *mid_slot;
   unsigned int *mid_count;
   uint32_t *flag_bits;                // always null here
   // the fused kernel's own
   unsigned long long *ctrl;           // control block (mm_internal.h)
   uint64_t *host_result;              // pinned: [8 header words][slots]
   uint64_t *dev_result;               // the same block in HBM, for the multi-GPU gather
   uint32_t max_rank;                  // slots the published block holds
   uint32_t group_min;                 // mm_scan_tail2: more candidates than this are taken several to a wave
   uint32_t ctrl_words;
   uint64_t seq;                       // raised in host_result[MM_HDR_FLAG_WORD] when everything is published
   uint64_t timeout_ticks;             // wall_clock64() ticks (100 MHz) a barrier may take
   uint32_t has_edge;                  // bytes behind the last whole 4 KiB group
};

// Thread 0 of every workgroup calls this once per counter set (`lines`: MM_ARRIVE_LINES lines of
// MM_LIST_STRIDE words); true for exactly one caller, the last of all nblocks to arrive.  Two
// levels, MM_ARRIVE_FAN arrivals per address (see mm_internal.h).
__device__ __forceinline__ bool mm_arrive_last(unsigned long long *lines, uint32_t nblocks)
{
   const uint32_t group = blockIdx.x / MM_ARRIVE_FAN;
   const uint32_t ngroups = (nblocks + MM_ARRIVE_FAN - 1) / MM_ARRIVE_FAN;
   const uint32_t members = group + 1 < ngroups ? (uint32_t)MM_ARRIVE_FAN : nblocks - group * MM_ARRIVE_FAN;
   unsigned long long *mine = lines + (1 + group) * MM_LIST_STRIDE;
   if (__hip_atomic_fetch_add(mine, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 != members) {
      return false;
   }
   return __hip_atomic_fetch_add(lines, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == ngroups;
}

// Every workgroup arrives once; returns false when the others did not within the timeout.  All
// waves must have drained their write-through stores before (s_waitcnt vmcnt(0)): the arrival is
// what publishes them.  Waiting workgroups poll one of MM_CAND_LISTS release flags (16 pollers
// per cache line at most, not a thousand on the counter the late arrivals still have to reach).
__device__ __forceinline__ bool mm_grid_barrier(unsigned long long *ctrl, uint32_t nblocks, uint64_t timeout_ticks)
{
   __shared__ unsigned long long verdict_sh;
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
   __syncthreads();
   if (threadIdx.x < 64) {                                  // wave 0
      const int lane = threadIdx.x;
      const unsigned long long now = wall_clock64();
      unsigned long long *flags = ctrl + MM_CTRL_RELEASE;
      unsigned long long verdict = 0;                       // wave uniform
      bool last = false;
      if (lane == 0) {
         last = mm_arrive_last(ctrl + MM_CTRL_ARRIVE_BARRIER, nblocks);
      }
      if (__ballot(last) != 0) {
         if (lane == 0) {
            // (statistics: the moment the slowest workgroup left the streaming phase)
            __hip_atomic_store(ctrl + MM_CTRL_T_BARRIER, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long expected = 0;
            verdict = __hip_atomic_compare_exchange_strong(ctrl + MM_CTRL_DECISION, &expected, 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT) ? 1ull : expected;
         }
         verdict = mm_uniform64(verdict);
         __hip_atomic_store(flags + lane * MM_LIST_STRIDE, verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one flag per lane
      }
      else {
         unsigned long long *flag = flags + (blockIdx.x & (MM_CAND_LISTS - 1)) * MM_LIST_STRIDE;
         for (unsigned spins = 1;; spins++) {
            if (lane == 0) {
               verdict = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            verdict = mm_uniform64(verdict);
            if (verdict != 0) {
               break;
            }
            if ((spins & 63) == 0 && wall_clock64() - now > timeout_ticks) {
               // give up -- unless the last arrival has just decided otherwise; whoever decides tells everybody
               unsigned long long won = 0;
               if (lane == 0) {
                  unsigned long long expected = 0;
                  won = __hip_atomic_compare_exchange_strong(ctrl + MM_CTRL_DECISION, &expected, 2ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT) ? 1ull : 0ull;
               }
               if (mm_uniform64(won)) {
                  __hip_atomic_store(flags + lane * MM_LIST_STRIDE, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               }
            }
            __builtin_amdgcn_s_sleep(8);
         }
      }
      if (lane == 0) {
         verdict_sh = verdict;
      }
   }
   __syncthreads();
   return verdict_sh == 1;
}

// ---- the tail of a scan: every candidate resolved, ranked and published by one wave ------------
//
// Shared by the fused kernel (behind its grid barrier: SAME_LAUNCH, the candidate lists were
// written during this launch and are read past the caches) and by mm_scan_tail, the kernel that
// follows mm_filter_* on large ROMs (plain cached loads).

constexpr int MM_TAIL_KEYS = 10;               // candidate keys a thread holds in registers per ranking pass (2560 per workgroup)

template <bool SAME_LAUNCH>
__device__ __forceinline__ unsigned long long mm_tail_load(const unsigned long long *p)
{
   return SAME_LAUNCH ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

struct MmTailLds {
   unsigned long long wkey[MM_WAVES];          // the candidates the workgroup's waves are working on (~0: none)
   unsigned int smaller[MM_WAVES];             // how many candidates have a smaller offset
   unsigned int holes, maxcount;
   int last_block;
};

// thread t holds entries q, q + 4, q + 8, ... (q = t / 64) of candidate list t % 64, from entry `base` on
template <bool SAME_LAUNCH>
__device__ __forceinline__ void mm_tail_keys(const MmFusedArgs &a, unsigned long long count, unsigned long long base,
                                             unsigned long long (&v)[MM_TAIL_KEYS])
{
   const unsigned long long *list = reinterpret_cast<const unsigned long long *>(a.cand) + (uint64_t)(threadIdx.x & 63) * a.list_cap;
#pragma unroll
   for (int u = 0; u < MM_TAIL_KEYS; u++) {
      const unsigned long long j = base + (threadIdx.x >> 6) + 4ull * u;
      v[u] = j < count ? mm_tail_load<SAME_LAUNCH>(list + j) : ~0ull;      // ~0 is never smaller than a candidate
   }
}

template <bool SAME_LAUNCH, bool LONG = false>
__device__ __forceinline__ void mm_scan_tail_phase(const MmFusedArgs &a, const MmPlanLds &P, MmWaveLdsShort *Wv, MmResolveLds &R,
                                                   MmTailLds &T, bool together)
{
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   if (threadIdx.x == 0) {
      T.holes = 0;
   }
   // the 64 list counters: one (cache-bypassing, in the fused kernel) load per lane of wave 0
   if (threadIdx.x < 64) {
      const unsigned long long my_count = mm_tail_load<SAME_LAUNCH>(a.list_count + lane * MM_LIST_STRIDE);
      unsigned long long incl = my_count;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
         const unsigned long long up = __shfl_up(incl, d);
         incl += lane >= d ? up : 0ull;
      }
      R.excl[lane] = incl - my_count;
      const bool overflow = __ballot(my_count > a.list_cap) != 0;
      unsigned long long most = my_count;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
         const unsigned long long other = __shfl_xor(most, d);
         most = other > most ? other : most;
      }
      if (lane == 63) {
         R.ncand = overflow ? ~0ull : incl;
         R.walked = 0;
         T.maxcount = (unsigned int)(most > 0xFFFFFFFFull ? 0xFFFFFFFFull : most);
      }
   }
   __syncthreads();
   const unsigned long long excl = R.excl[lane];
   const unsigned long long ncand = R.ncand;
   const bool resolvable = together && ncand <= a.out_cap && ncand <= a.max_candidates && ncand <= a.max_rank;
   unsigned long long walked = 0;
   if (resolvable) {
      // this thread's list and how many entries it has
      const uint32_t c = threadIdx.x & 63;
      const unsigned long long my_list_count = (c + 1 < MM_CAND_LISTS ? R.excl[c + 1] : ncand) - R.excl[c];
      const uint64_t nwaves = (uint64_t)gridDim.x * MM_WAVES;
      const uint64_t rounds = (ncand + nwaves - 1) / nwaves;          // block-uniform trip count: the ranking synchronises the workgroup
      for (uint64_t rd = 0; rd < rounds; rd++) {
         const uint64_t first = rd * nwaves + (uint64_t)blockIdx.x * MM_WAVES;
         if (first >= ncand) {
            break;                              // (uniform in the workgroup: none of its waves has a candidate)
         }
         const uint64_t ci = first + wave;
         const bool live = ci < ncand;
         uint64_t o = ~0ull;
         if (live) {
            const int list = __popcll(__ballot(excl <= ci)) - 1;
            const unsigned long long *slot = reinterpret_cast<const unsigned long long *>(a.cand) + (uint64_t)list * a.list_cap +
                                             (ci - __shfl(excl, list));
            o = mm_uniform64(mm_tail_load<SAME_LAUNCH>(slot));
         }
         if (lane == 0) {
            T.wkey[wave] = o;
            T.smaller[wave] = 0;
         }
         int verdict = 0;
         int64_t hi = 0; mm_set_t set = 0; uint64_t dom = 0;
         if (live) {
            verdict = mm_resolve_any<LONG>(a, P, Wv[wave], o, lane, &walked, &hi, &set, &dom);
         }
         // Ranking: offsets are unique, so a candidate's place in the ascending list is the number of
         // candidates with a smaller offset.  The workgroup's 256 threads share out the candidate set
         // (thread t: a quarter of list t % 64), hold their share in registers and count for all four
         // waves' candidates at once.  (Loading the keys before the resolver would hide the one round
         // trip, but 40 registers held across it cost a workgroup per CU: 55 us instead of 25.)
         asm volatile("" ::: "memory");         // (keeps the compiler from hoisting the loads above the resolver)
         unsigned long long keys[MM_TAIL_KEYS];
         mm_tail_keys<SAME_LAUNCH>(a, my_list_count, 0, keys);
         __syncthreads();                       // every wave's key is in T.wkey
         unsigned int cnt[MM_WAVES] = {0, 0, 0, 0};
         for (unsigned long long base = 0;;) {
#pragma unroll
            for (int u = 0; u < MM_TAIL_KEYS; u++) {
#pragma unroll
               for (int w = 0; w < MM_WAVES; w++) {
                  cnt[w] += keys[u] < T.wkey[w] ? 1u : 0u;
               }
            }
            base += 4ull * MM_TAIL_KEYS;
            if (base >= T.maxcount) {
               break;
            }
            mm_tail_keys<SAME_LAUNCH>(a, my_list_count, base, keys);     // lists longer than a register pass: rare
         }
#pragma unroll
         for (int w = 0; w < MM_WAVES; w++) {
            unsigned int s = cnt[w];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
               s += (unsigned int)__shfl_xor((int)s, d);
            }
            if (lane == 0 && s) {
               atomicAdd(&T.smaller[w], s);
            }
         }
         __syncthreads();
         if (live && lane == 0) {
            const uint32_t rank = T.smaller[wave];
            const uint64_t value = verdict == 1 ? mm_report_value(a.t.g, o, a.base_offset) : MM_NO_MATCH;
            a.out[ci] = value;
            // system-scope store: written through to host memory, complete once this wave's vmcnt drains
            // (a __threadfence_system() per workgroup instead -- an L2 write-back each -- cost 15 us of a 4 GiB scan's tail)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.host_result) + MM_RESULT_HEADER_WORDS + rank, (unsigned long long)value,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            a.dev_result[MM_RESULT_HEADER_WORDS + rank] = value;
            if (verdict != 1) {
               atomicAdd(&T.holes, 1u);
            }
            if (verdict == -1) {
               mm_resolve_hand_over(a, o, ci, hi, set, dom);
            }
         }
         __syncthreads();                       // T.wkey / T.smaller are rewritten by the next round
      }
   }
   // ---- statistics, header, flag ---------------------------------------------------------------
   if (blockIdx.x == 0 && threadIdx.x == 0) {
      __hip_atomic_store(a.ctrl + MM_CTRL_T_STAMPS + 1, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
   if (lane == 0 && walked) {
      atomicAdd(&R.walked, (unsigned int)walked);
   }
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's result stores have arrived (host memory: written through)
   __syncthreads();
   if (threadIdx.x == 0) {
      if (R.walked) {
         atomicAdd(a.tiles_walked + (blockIdx.x % MM_STAT_STRIPES), (unsigned long long)R.walked);
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
      // one round trip for all the words the header is made of: lane k reads word k
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
      const unsigned long long t0 = __shfl(v, MM_CTRL_T_START), t1 = __shfl(v, MM_CTRL_T_BARRIER);
      // tuning aid (header word 1): when workgroup 0 left the barrier and finished its candidates, and now,
      // in wall-clock ticks after the last arrival at the barrier
      const unsigned long long s2 = __shfl(v, MM_CTRL_T_STAMPS), s3 = __shfl(v, MM_CTRL_T_STAMPS + 1), s4 = wall_clock64();
      unsigned long long h = 0;
      switch (lane) {
      case 0: h = ncand; break;                            // candidates = slots (~0: a list overflowed)
      case 1: h = ((s2 - t1) & 0xFFFFF) | (((s3 - t1) & 0xFFFFF) << 20) | (((s4 - t1) & 0xFFFFF) << 40); break;
      case 2: h = tiles; break;
      // [4]: flags in the low byte, above them the streaming phase's duration in wall-clock ticks
      case 4: h = (resolvable ? MM_HDR_SPARSE : 0) | (together ? 0 : MM_HDR_GAVE_UP) | ((t1 > t0 ? t1 - t0 : 0) << 8); break;
      case 5: h = mid; break;
      case 6: h = resolvable ? ncand - nomatch + 1 : 0; break;   // matches + 1 (0: not ordered here)
      default: break;                                      // [3] unused here; [7] = 0: the list length is word 0
      }
      if (lane < (int)MM_RESULT_HEADER_WORDS) {
         a.host_result[lane] = h;
         a.dev_result[lane] = h;
      }
      if (lane == 0) {
         R.walked = (unsigned int)(mid & 0xFFFFFFFFull);   // (for the decision below)
      }
   }
   __syncthreads();
   // the control block goes back to zero for the next scan -- unless the second phase follows
   // (left-overs) or the plain kernels take over (gave up / too many candidates): they need the
   // candidate counters
   const bool keep = !resolvable || R.walked != 0;
   if (keep) {
      if (threadIdx.x == 0) {
         a.ctrl[MM_CTRL_NOMATCH] = 0;
         a.ctrl[MM_CTRL_DECISION] = 0;
         a.ctrl[MM_CTRL_T_START] = 0;
         a.ctrl[MM_CTRL_T_BARRIER] = 0;
         a.ctrl[MM_CTRL_T_STAMPS] = 0;
         a.ctrl[MM_CTRL_T_STAMPS + 1] = 0;
         a.ctrl[MM_CTRL_TOTAL] = ncand;                    // what mm_resolve would have left for the later stages
      }
      for (uint32_t k = MM_CTRL_ARRIVE_BARRIER + threadIdx.x; k < MM_CTRL_WORDS; k += blockDim.x) {
         a.ctrl[k] = 0;                                    // arrival counters and release flags
      }
   }
   else {
      for (uint32_t k = threadIdx.x; k < a.ctrl_words; k += blockDim.x) {
         a.ctrl[k] = 0;
      }
   }
   __syncthreads();
   if (threadIdx.x == 0) {
      __threadfence_system();
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.host_result) + MM_HDR_FLAG_WORD, (unsigned long long)a.seq,
                         __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
   }
}

// One launch for a whole (small) scan.  (second launch bound: 4 waves per SIMD, i.e. <= 128 VGPRs)
template <int ELEM, int SHAPE>
__global__ __launch_bounds__(64 * MM_WAVES, 4) void mm_scan_fused(MmFusedArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLdsShort Wv[MM_WAVES];
   __shared__ MmResolveLds R;
   __shared__ MmTailLds T;

   if (blockIdx.x == 0 && threadIdx.x == 0) {
      __hip_atomic_store(a.ctrl + MM_CTRL_T_START, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
   mm_plan_to_lds(P, a.t.plan);                 // (before the streaming phase: it needs no candidates)
   // ---- phase A: the streaming filter ------------------------------------------------------
   if (a.ngroups) {
      if (ELEM == 1) {
         mm_stream_u8<SHAPE>(a);
      }
      else {
         mm_stream_u16<SHAPE>(a);
      }
   }
   if (a.has_edge && blockIdx.x == gridDim.x - 1) {
      if (ELEM == 1) {
         mm_edge_u8<SHAPE>(a, 0, 1);
      }
      else {
         mm_edge_u16<SHAPE>(a, 0, 1);
      }
   }
   const bool together = mm_grid_barrier(a.ctrl, gridDim.x, a.timeout_ticks);
   if (blockIdx.x == 0 && threadIdx.x == 0) {
      __hip_atomic_store(a.ctrl + MM_CTRL_T_STAMPS, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
   // ---- phase B: resolve + rank + publish ----------------------------------------------------
   mm_scan_tail_phase<true>(a, P, Wv, R, T, together);
}

// The tail as a kernel of its own, behind mm_filter_* (large ROMs: the streaming kernel keeps
// its own, leaner launch and all the wave slots; this one replaces mm_resolve, mm_rank_count and
// mm_rank_scatter -- one dependent launch instead of three).
// OCC waves per SIMD: 5 (<= 96 VGPRs) keeps the 1056 workgroups the bench ROM's candidates need resident at once
template <int OCC, bool LONG = false>
__global__ __launch_bounds__(64 * MM_WAVES, OCC) void mm_scan_tail(MmFusedArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLdsShort Wv[MM_WAVES];
   __shared__ MmResolveLds R;
   __shared__ MmTailLds T;
   mm_plan_to_lds(P, a.t.plan);
   mm_scan_tail_phase<false, LONG>(a, P, Wv, R, T, true);
}

#endif
