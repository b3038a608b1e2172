// mm_fused.h -- one launch for a whole scan: streaming filter -> grid barrier -> every candidate
// resolved AND ranked by one wave -> results published.  Included by mm_kernels.hip after
// mm_tiles.h (device code only).
//
// Why: behind the streaming kernel a scan used to run three more dependent launches (mm_resolve,
// mm_rank_count, mm_rank_scatter: 32 us on the bench ROM, most of it launch and dependency
// latency, round 1) and the host then waited on an event (~12 us).  Here the same work is one
// kernel:
//   phase A  the streaming code of mm_filter_u8 / mm_filter_u16, unchanged (mm_stream_*); the
//            ragged end of the ROM goes to the last workgroup (mm_edge_*);
//   barrier  every workgroup arrives once (one agent-scope atomic each); candidates were stored
//            write-through and are read past L1 afterwards (cdna guideline 16 R1), so no fence;
//   phase B  wave w takes candidate w: the reference's compare loop + chain membership through
//            the two short look-back windows (mm_resolve_candidate, the code of mm_resolve), and
//            its RANK among all candidates (offsets are unique: rank = number of smaller ones,
//            counted against the candidate lists staged in LDS) -- the ordering needs no verdicts,
//            so it needs no second barrier.  The result goes to slot `rank` of the published
//            block: the reported value, or a hole (~0) for a candidate that is not reported.
//            Holes are rare (on the bench ROM: none); the host drops them while copying out.
//   end      the last workgroup to finish writes the header, zeroes the control block for the
//            next scan and raises a flag word in pinned host memory -- the host polls that word
//            instead of waiting for a HIP event.
// Candidates the two windows cannot settle are handed to the second phase exactly as mm_resolve
// does (mid list; the host launches mm_resolve2 / mm_hard_resolve and the rank kernels then).
//
// A grid barrier needs every workgroup resident at once: the grid is sized well below the
// device's capacity (host side, launch_fused), every spin is bounded by a wall-clock timeout, and
// a timeout makes the kernel give up cleanly (header word 4 bit 1) -- the host then runs the
// plain kernels on the candidate lists, which are complete by then or rebuilt.  Two fused
// kernels at once could starve each other of slots, so the host never has more than one in
// flight per process (a process-wide try-lock; the loser takes the plain path).
#ifndef MM_FUSED_H
#define MM_FUSED_H

constexpr int MM_FUSED_RANK_SLICE = 1024;      // candidate keys staged in LDS per ranking pass (8 KiB)

// header word 4 of the published block
constexpr unsigned long long MM_HDR_SPARSE = 1;     // the list holds one slot per candidate, ~0 = hole
constexpr unsigned long long MM_HDR_GAVE_UP = 2;    // a barrier timed out: nothing was resolved

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
   uint32_t *mid_set;
   uint32_t *mid_slot;
   unsigned int *mid_count;
   uint32_t *flag_bits;                // always null here
   // the fused kernel's own
   unsigned long long *ctrl;           // control block (mm_internal.h)
   uint64_t *host_result;              // pinned: [8 header words][slots]
   uint64_t *dev_result;               // the same block in HBM, for the multi-GPU gather
   uint32_t max_rank;                  // slots the published block holds
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

// rank of `key` among the candidates = number of candidates with a smaller offset; the workgroup
// stages the candidate lists slice by slice in LDS (keys[]), each wave counts for its own key.
// All four waves call this together (block-wide barriers inside); `live` waves get their rank.
__device__ __forceinline__ uint32_t mm_fused_rank(const MmFusedArgs &a, const MmResolveLds &R, uint64_t *keys, uint64_t ncand,
                                                  uint64_t key, bool live, int lane)
{
   uint32_t smaller = 0;
   for (uint64_t s0 = 0; s0 < ncand; s0 += MM_FUSED_RANK_SLICE) {
      const uint32_t n = (uint32_t)(ncand - s0 < MM_FUSED_RANK_SLICE ? ncand - s0 : MM_FUSED_RANK_SLICE);
      __syncthreads();                                     // the previous slice has been consumed
      for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
         // compact number s0 + k -> its list and entry (the 64 prefix sums sit in LDS)
         const uint64_t ci = s0 + k;
         uint32_t lo = 0, hi = MM_CAND_LISTS - 1;
         while (lo < hi) {                                  // last list whose first number is <= ci
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (R.excl[mid] <= ci) {
               lo = mid;
            }
            else {
               hi = mid - 1;
            }
         }
         const unsigned long long *slot = reinterpret_cast<const unsigned long long *>(a.cand) + (uint64_t)lo * a.list_cap + (ci - R.excl[lo]);
         keys[k] = mm_load_shared(slot);
      }
      __syncthreads();
      if (live) {
         for (uint32_t k = (uint32_t)lane; k < n; k += 64) {
            smaller += keys[k] < key ? 1u : 0u;
         }
      }
   }
#pragma unroll
   for (int d = 32; d >= 1; d >>= 1) {
      smaller += (uint32_t)__shfl_xor((int)smaller, d);
   }
   return smaller;
}

// (second launch bound: 4 waves per SIMD, i.e. <= 128 VGPRs -- 4 workgroups per CU must fit)
template <int ELEM, int SHAPE>
__global__ __launch_bounds__(64 * MM_WAVES, 4) void mm_scan_fused(MmFusedArgs a)
{
   __shared__ MmPlanLds P;
   __shared__ MmWaveLdsShort Wv[MM_WAVES];
   __shared__ MmResolveLds R;
   __shared__ uint64_t keys[MM_FUSED_RANK_SLICE];
   __shared__ int last_block;
   __shared__ unsigned int holes, wrote;
   const int wave = (int)mm_uniform(threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;

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
   if (threadIdx.x == 0) {
      holes = 0;
      wrote = 0;
   }
   mm_resolve_prefix(a, R);
   __syncthreads();
   const unsigned long long excl = R.excl[lane];
   const unsigned long long ncand = R.ncand;
   const bool resolvable = together && ncand <= a.out_cap && ncand <= a.max_candidates && ncand <= a.max_rank;
   unsigned long long walked = 0;
   if (resolvable) {
      const uint64_t nwaves = (uint64_t)gridDim.x * MM_WAVES;
      // block-uniform trip count: the ranking synchronises the workgroup
      const uint64_t rounds = (ncand + nwaves - 1) / nwaves;
      for (uint64_t rd = 0; rd < rounds; rd++) {
         const uint64_t first = rd * nwaves + (uint64_t)blockIdx.x * MM_WAVES;
         if (first >= ncand) {
            break;                              // (uniform in the workgroup: none of its waves has a candidate)
         }
         const uint64_t ci = first + wave;
         const bool live = ci < ncand;
         uint64_t o = 0;
         int verdict = 0;
         int64_t hi = 0; uint32_t set = 0; uint64_t dom = 0;
         if (live) {
            o = mm_candidate(a, excl, ci);
            verdict = mm_resolve_candidate(a, P, Wv[wave], o, lane, &walked, &hi, &set, &dom);
         }
         const uint32_t rank = mm_fused_rank(a, R, keys, ncand, o, live, lane);
         if (live && lane == 0) {
            const uint64_t value = verdict == 1 ? mm_report_value(a.t.g, o, a.base_offset) : MM_NO_MATCH;
            a.out[ci] = value;
            a.host_result[MM_RESULT_HEADER_WORDS + rank] = value;
            a.dev_result[MM_RESULT_HEADER_WORDS + rank] = value;
            wrote = 1;
            if (verdict != 1) {
               atomicAdd(&holes, 1u);
            }
            if (verdict == -1) {
               mm_resolve_hand_over(a, o, ci, hi, set, dom);
            }
         }
      }
   }
   // ---- end: statistics, header, flag ----------------------------------------------------------
   if (blockIdx.x == 0 && threadIdx.x == 0) {
      __hip_atomic_store(a.ctrl + MM_CTRL_T_STAMPS + 1, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
   if (lane == 0 && walked) {
      atomicAdd(&R.walked, (unsigned int)walked);
   }
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's result stores have left
   __syncthreads();
   if (threadIdx.x == 0) {
      if (R.walked) {
         atomicAdd(a.tiles_walked + (blockIdx.x % MM_STAT_STRIPES), (unsigned long long)R.walked);
      }
      if (wrote) {
         __threadfence_system();                           // this workgroup published results: in host memory before its arrival
      }
      if (holes) {
         atomicAdd(a.ctrl + MM_CTRL_NOMATCH, (unsigned long long)holes);
      }
      last_block = mm_arrive_last(a.ctrl + MM_CTRL_ARRIVE_END, gridDim.x);
   }
   __syncthreads();
   if (!last_block) {
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
      // tuning aid (header words 1 and 3): when workgroup 0 left the barrier and finished its candidates, and now,
      // in wall-clock ticks after the last arrival at the barrier
      const unsigned long long s2 = __shfl(v, MM_CTRL_T_STAMPS), s3 = __shfl(v, MM_CTRL_T_STAMPS + 1), s4 = wall_clock64();
      unsigned long long h = 0;
      switch (lane) {
      case 0: h = ncand; break;                            // candidates = slots (~0: a list overflowed)
      case 2: h = tiles; break;
      // [4]: flags in the low byte, above them the streaming phase's duration in wall-clock ticks
      case 4: h = (resolvable ? MM_HDR_SPARSE : 0) | (together ? 0 : MM_HDR_GAVE_UP) | ((t1 > t0 ? t1 - t0 : 0) << 8); break;
      case 5: h = mid; break;
      case 6: h = resolvable ? ncand - nomatch + 1 : 0; break;   // matches + 1 (0: not ordered here)
      case 1: h = ((s2 - t1) & 0xFFFFF) | (((s3 - t1) & 0xFFFFF) << 20) | (((s4 - t1) & 0xFFFFF) << 40); break;
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

#endif
