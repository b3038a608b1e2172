// SPDX-License-Identifier: GPL-3.0-or-later
// mm_kernels.hip -- launchers and the small kernels of the gfx950 (MI355X / CDNA4) relative-search engine.
//
// The device code of the hot path lives in headers (everything a template or __forceinline__):
//   mm_filter.h   the streaming filter -- mm_filter_u8 / mm_filter_u16<SHAPE>, HBM-bound: coalesced 16 B/lane loads, two SWAR
//                 stages, survivors (CANDIDATES: positions where the reference's compare loop may report a match IF its chain
//                 visits them) into ROM-ordered buckets -- whose 62 shapes are instantiated in the units of
//                 mm_filter_shapes.hip and looked up here by table (mm_filter_shapes.h);
//   mm_tail2.h    mm_scan_tail2, one launch behind the filter: every candidate verified, resolved and ranked into the
//                 ascending list in pinned host memory and HBM; header; the flag word the host polls;
//   mm_fused.h    mm_scan_fused (ROMs of up to 4 MiB: both stages in ONE launch, a grid barrier between them) and the
//                 list-based tail mm_scan_tail;
//   mm_tiles.h    the resolvers: exact chain membership through phase maps of look-back windows (mm_resolve_candidate, and
//                 _long for keywords beyond 64 symbols), mm_resolve / mm_resolve2 / mm_hard_resolve for what they leave open;
//   mm_forward.h  mm_forward, the candidate-free forward engine, for inputs the per-candidate path does not suit.
// Here: the launch wrappers mm_capi.hip calls (choose_filter: which SWAR conditions a plan gets), the sequential
// cross-check engine mm_chain_seq, the rank kernels of the event-synchronised chain, ROM fill / gather helpers.
// Routes are switched at run time (mmh_set_route), not by environment.
//
// No MFMA anywhere: the path is integer byte comparison (BASELINE.json).
#include <atomic>
#include <cstdio>

#include "mm_filter.h"
#include "mm_filter_shapes.h"

// --------------------------------------------------------------------------
// sequential engine: one lane per domain, exact by construction
// --------------------------------------------------------------------------

struct MmSeqArgs {
   MmGeom g;
   mmh_plan_desc plan;
   uint64_t *out;
   unsigned long long *out_count;
   uint64_t out_cap;
   uint64_t base_offset;
};

__global__ __launch_bounds__(64) void mm_chain_seq(MmSeqArgs a)
{
   const uint64_t ndom = a.g.whole ? 1 : a.g.nblocks * a.g.S;
   const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
   // every lane of the wave runs the loop the same number of times so that the
   // wave-aggregated append stays convergent
   const uint64_t rounds = (ndom + stride - 1) / stride;
   for (uint64_t rd = 0; rd < rounds; rd++) {
      uint64_t dom = rd * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
      bool live = dom < ndom;
      uint64_t b = live ? dom / a.g.S : 0;
      uint32_t p = live ? (uint32_t)(dom % a.g.S) : 0;
      int64_t nv = live ? mm_domain_nv(a.g, b, p) : 0;
      uint64_t start = mm_domain_start(a.g, b, p);
      int64_t h = 0;
      while (__ballot(live && h < nv) != 0) {
         bool want = false;
         uint64_t val = 0;
         if (live && h < nv) {
            bool matched;
            int J = mm_step(a.plan, [&](int64_t k) { return mm_elem(a.g, start, k); }, h, &matched);
            if (matched) {
               want = true;
               val = a.g.whole ? (uint64_t)h : start + (uint64_t)h * a.g.S + a.base_offset;
            }
            h += J;
         }
         mm_append(a.out, a.out_count, a.out_cap, want, val);
      }
   }
}

// --------------------------------------------------------------------------
// ordering of the match list (unique keys): rank = number of smaller keys
// --------------------------------------------------------------------------
//
// Two small launches with fixed grids (the count only exists on the device):
//   mm_rank_count    block (x, s): for keys i = x*256+tid (grid-stride) count the
//                    keys of slice s that are smaller -> partials[s][i]
//   mm_rank_scatter  rank = sum of the partials; result[rank] = key, written
//                    straight into pinned host memory together with the counters
// Lists longer than max_n are ordered by the host (search_engine.cpp:193-197).

constexpr int MM_RANK_SLICES = 32;

__global__ __launch_bounds__(256) void mm_rank_count(const uint64_t *in, const unsigned long long *count, uint64_t cap,
                                                     uint32_t max_n, uint32_t *partials)
{
   __shared__ uint64_t slice[1024];
   const unsigned long long n64 = *count;
   if (n64 > cap || n64 > max_n || n64 == 0) {
      return;
   }
   const uint32_t n = (uint32_t)n64;
   const uint32_t slen = (n + MM_RANK_SLICES - 1) / MM_RANK_SLICES;      // <= max_n / 32 <= 1024
   const uint32_t s0 = blockIdx.y * slen;
   if (s0 >= n || blockIdx.x * 256u >= n) {
      // slices past the end still owe zeros to the scatter pass
      if (s0 >= n) {
         for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
            partials[blockIdx.y * max_n + i] = 0;
         }
      }
      return;
   }
   const uint32_t s1 = s0 + slen < n ? s0 + slen : n;
   for (uint32_t k = threadIdx.x; k < s1 - s0; k += 256) {
      slice[k] = in[s0 + k];
   }
   __syncthreads();
   for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
      const uint64_t key = in[i];
      uint32_t r = 0;
      const uint32_t len = s1 - s0;
#pragma unroll 8
      for (uint32_t k = 0; k < len; k++) {
         r += slice[k] < key ? 1u : 0u;
      }
      partials[blockIdx.y * max_n + i] = r;
   }
}

// `in` holds ctrl[count_index] keys; MM_NO_MATCH keys (candidates that are not on the
// chain) sort behind everything else and are dropped.  Every block counts the ones it meets
// into ctrl[MM_CTRL_NOMATCH]; the last block to finish publishes "matches + 1" as word 6 of the
// header (0 = the list was too long to be ordered here: the host sorts the slots itself).
// The block is written twice: to pinned host memory (what mmh_scan returns) and to device
// memory (dev_result: what the multi-GPU gather sends without a host round trip).
__global__ __launch_bounds__(256) void mm_rank_scatter(const uint64_t *in, unsigned long long *ctrl, int count_index,
                                                       uint64_t cap, uint32_t max_n, const uint32_t *partials,
                                                       uint64_t *host_result, uint64_t *dev_result, uint32_t ctrl_words,
                                                       uint32_t keep_leftovers)
{
   __shared__ int last_block;
   __shared__ unsigned int holes;
   const unsigned long long n64 = ctrl[count_index];
   // keep_leftovers: the scan's first phase -- when mm_resolve left candidates over, the host
   // launches the second phase (mm_resolve2, mm_hard_resolve, this ordering again), which needs
   // the control block as it is
   const bool leftovers = keep_leftovers && (ctrl[MM_CTRL_MID] & 0xFFFFFFFFull) != 0;
   if (threadIdx.x == 0) {
      holes = 0;
   }
   if (blockIdx.x == 0 && threadIdx.x < 8 && threadIdx.x != 6) {
      unsigned long long v = ctrl[threadIdx.x];            // counters travel with the results
      if (threadIdx.x == 2) {
         v = 0;
         for (int k = 0; k < MM_STAT_STRIPES; k++) {
            v += ctrl[MM_CTRL_TILES + k];
         }
      }
      if (threadIdx.x == 5) {
         v = ctrl[MM_CTRL_MID];                            // word 2 of the header carries the tile count
      }
      if (threadIdx.x == 4) {
         v = 0;                                            // flags (mm_fused.h MM_HDR_*): a dense list
      }
      if (threadIdx.x == 7) {
         v = (unsigned long long)count_index;              // which header word holds the list length
      }
      host_result[threadIdx.x] = v;
      dev_result[threadIdx.x] = v;
   }
   __syncthreads();
   const bool ordered = n64 <= cap && n64 <= max_n;
   if (ordered) {
      const uint32_t n = (uint32_t)n64;
      unsigned int mine = 0;
      for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
         uint32_t r = 0;
#pragma unroll
         for (int s = 0; s < MM_RANK_SLICES; s++) {
            r += partials[s * max_n + i];
         }
         const uint64_t key = in[i];
         if (key == MM_NO_MATCH) {
            mine++;
         }
         else {
            host_result[8 + r] = key;
            dev_result[8 + r] = key;
         }
      }
      if (mine) {
         atomicAdd(&holes, mine);
      }
   }
   // The last block to get here publishes the number of matches and leaves the control block
   // zeroed for the next scan (this is the last kernel of a scan), which saves a memset launch
   // + its stream gap per scan.
   __syncthreads();
   if (threadIdx.x == 0) {
      if (holes) {
         atomicAdd(ctrl + MM_CTRL_NOMATCH, (unsigned long long)holes);
      }
      __threadfence();
      last_block = atomicAdd(ctrl + MM_CTRL_TICKET, 1ull) == gridDim.x - 1;
   }
   __syncthreads();
   if (last_block) {
      if (threadIdx.x == 0) {
         __threadfence();
         const unsigned long long nomatch = atomicAdd(ctrl + MM_CTRL_NOMATCH, 0ull);
         const unsigned long long word6 = ordered ? n64 - nomatch + 1 : 0ull;
         host_result[6] = word6;
         dev_result[6] = word6;
      }
      __syncthreads();
      if (leftovers) {
         if (threadIdx.x == 0) {
            ctrl[MM_CTRL_TICKET] = 0;                        // the second phase's ordering counts arrivals again
            ctrl[MM_CTRL_NOMATCH] = 0;
         }
      }
      else {
         for (uint32_t k = threadIdx.x; k < ctrl_words; k += blockDim.x) {
            ctrl[k] = 0;
         }
      }
   }
}

// --------------------------------------------------------------------------
// synthetic ROM (bench / tests): splitmix64 words, SURVEY 8d
// --------------------------------------------------------------------------

__device__ __forceinline__ uint64_t mm_splitmix(uint64_t seed, uint64_t k)
{
   uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
   return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void mm_synth_fill(uint8_t *rom, uint64_t nbytes, uint64_t seed, uint64_t base_offset)
{
   // base_offset is a multiple of 16: each thread writes two consecutive 64-bit words
   const uint64_t npairs = (nbytes + 15) / 16;
   const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
   for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += stride) {
      uint64_t k = (base_offset >> 3) + 2 * i;
      uint64_t w0 = mm_splitmix(seed, k), w1 = mm_splitmix(seed, k + 1);
      if (i * 16 + 16 <= nbytes) {
         *reinterpret_cast<ulonglong2 *>(rom + i * 16) = make_ulonglong2(w0, w1);
      }
      else {
         for (uint64_t q = i * 16; q < nbytes; q++) {
            uint64_t w = (q - i * 16) < 8 ? w0 : w1;
            rom[q] = (uint8_t)(w >> (8 * (q & 7)));
         }
      }
   }
}

__global__ __launch_bounds__(256) void mm_pattern_fill(uint8_t *rom, uint64_t first, uint64_t nbytes, int value, int ramp)
{
   const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
   for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += stride) {
      rom[first + i] = (uint8_t)(ramp ? value + (int)(i & 0xFF) * ramp : value);
   }
}

// packed copy of `each` bytes at every one of n ROM offsets (bytes behind the ROM read as 0)
__global__ __launch_bounds__(256) void mm_gather(const uint8_t *rom, uint64_t nbytes, const uint64_t *offsets, uint64_t n,
                                                 uint32_t each, uint8_t *out)
{
   const uint64_t total = n * each;
   const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
   for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const uint64_t at = offsets[i / each] + i % each;
      out[i] = at < nbytes ? rom[at] : (uint8_t)0;
   }
}

// One wave that does nothing for `ticks` of the 100 MHz wall clock: holds the kernels behind it on its stream back
// (mmh_scan_submit: the second scan of a burst starts half a streaming kernel behind the first, see there).
// A long list's way to the host (round 5): n ordered slots from the device-side copy of the result block into the pinned block,
// contiguous system-scope stores by as many workgroups as the list has KiB; the workgroup that finishes last raises `seq` in
// the word behind the scan's flag.  Launched by the host when it has seen the scan's flag and the list is on the device only
// (more than MM_DIRECT_PUBLISH slots) -- in place of hipMemcpy, which takes 82-120 us for 130-200 KiB on this stack
// whatever the device is doing; nothing waits on a stream: the host polls the word.
__global__ __launch_bounds__(256) void mm_publish_list(const uint64_t *src, uint64_t *dst, uint32_t n, unsigned long long *arrive,
                                                       unsigned long long *flag, unsigned long long seq)
{
   for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst) + k, (unsigned long long)src[k], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
   }
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's slots have arrived
   __syncthreads();
   if (threadIdx.x == 0) {
      const unsigned long long before = __hip_atomic_fetch_add(arrive, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (before + 1 == gridDim.x) {
         __hip_atomic_store(arrive, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (for the next list)
         __threadfence_system();
         __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
   }
}

// --------------------------------------------------------------------------
// launch wrappers (called from mm_capi.hip)
// --------------------------------------------------------------------------

namespace mm {

// Launch geometry and limits (the measured choices; DESIGN.md section 4 has the sweeps behind them).  One of them can be
// set from outside, for tests: MMOORE_MAX_CANDIDATES, where the per-candidate path hands a scan over to the flood paths.
const Tuning &tuning()
{
   static const Tuning t = [] {
      Tuning k;
      k.filter_max_conditions = 4;
      k.filter_blocks = 256 * 6;
      k.filter_groups_per_span = 7;               // (at most 8: a span's flagged pieces are 32 bits)
      k.resolve_blocks = 4096;
      // (mm_arrive_last counts arrivals in MM_ARRIVE_LINES - 1 groups of MM_ARRIVE_FAN: more workgroups would spill into the next lines)
      k.tail_blocks = (unsigned)std::min<long>(2048, (long)MM_ARRIVE_FAN * (MM_ARRIVE_LINES - 1));
      // (scans in flight: a small tail grid beside the next scan's streaming kernel -- between 512 and 1536 workgroups the
      // sparse steady state is the same, dense searches in flight like 1024 best, r03 / r04 sweeps)
      k.lane_tail_blocks = (unsigned)std::min<long>(1024, (long)MM_ARRIVE_FAN * (MM_ARRIVE_LINES - 1));
      const char *v = getenv("MMOORE_MAX_CANDIDATES");
      k.max_candidates = (uint32_t)(v && *v ? std::max(1L, atol(v)) : 1048576L);
      k.list_candidates = 262144;
      return k;
   }();
   return t;
}

// Picks the SWAR conditions of a plan: an anchor position iA (condition 0) and up to three
// more literal positions to its left, each compared with its left neighbour (gap 1) or, over
// one wildcard, with the literal two to the left (gap 2 = bridge -2 of the plan).  Condition k
// sits s_k = iA - position positions behind the anchor; the look-back of the kernels allows
//     8-bit : s_k + gap_k <= 4                       (one dword in front of a chunk)
//     16-bit: (s,gap) of condition 1 in {(1,1),(1,2),(2,1)}   (see mm_f16_chunk)
// Among the anchors the one with the most conditions wins; ties go to the contiguous run of
// adjacent literals (compile-time shifts), then to one whose first two conditions share their gap
// (round 5: the streaming loop's cost), then to the rightmost anchor.
bool choose_filter(const mmh_plan_desc &pl, FilterChoice *fc)
{
   const int L = (int)pl.L;
   const bool u8 = pl.elem_bytes == 1;
   const uint32_t emask = u8 ? 0xFFu : 0xFFFFu;
   const uint32_t rep = u8 ? 0x01010101u : 0x00010001u;
   auto gap_of = [&](int i) -> int {
      if (i < 1 || pl.cmp_mask[i] == 0) {
         return 0;
      }
      const int g = -(int)pl.bridge[i];
      return (g == 1 || g == 2) && i - g >= 0 ? g : 0;
   };
   const int want = std::min(u8 ? 4 : 2, tuning().filter_max_conditions);
   FilterChoice best;
   best.ncond = 0;
   bool best_contiguous = false, best_uniform = false;
   for (int A = L - 1; A >= 1; --A) {
      const int g0 = gap_of(A);
      if (!g0) {
         continue;
      }
      FilterChoice c;
      c.ncond = 1; c.iA = (uint32_t)A; c.shape = 0;
      c.pos[0] = (uint32_t)A; c.shift[0] = 0; c.gap[0] = (uint32_t)g0;
      for (int i = A - 1; i >= 1 && (int)c.ncond < want; --i) {
         const int g = gap_of(i), s = A - i;
         if (!g) {
            continue;
         }
         const bool fits = u8 ? (s + g <= 4) : ((s == 1 && g <= 2) || (s == 2 && g == 1));
         if (fits) {
            c.pos[c.ncond] = (uint32_t)i; c.shift[c.ncond] = (uint32_t)s; c.gap[c.ncond] = (uint32_t)g;
            c.ncond++;
         }
      }
      bool contiguous = true;
      for (uint32_t k = 0; k < c.ncond; k++) {
         contiguous = contiguous && c.shift[k] == k && c.gap[k] == 1;
      }
      // The streaming loop tests conditions 0 and 1 on every byte (stage 1; the others on flagged pieces only).  When those
      // two have the SAME gap it needs one SWAR subtraction per dword, else two (gap-1 AND gap-2 deltas: 21 instead of 14
      // VALU operations per dword -- `ab*de`: 348 M against 215 M wave instructions per 4 GiB, a kernel bound by them,
      // profiles/r05_dense_sq_counters.txt): a condition with the anchor's gap moves up to place 1 where there is one.
      if (u8 && c.ncond >= 3 && c.gap[1] != c.gap[0]) {
         for (uint32_t k = 2; k < c.ncond; k++) {
            if (c.gap[k] == c.gap[0]) {
               std::swap(c.pos[1], c.pos[k]);
               std::swap(c.shift[1], c.shift[k]);
               std::swap(c.gap[1], c.gap[k]);
               break;
            }
         }
      }
      const bool uniform = c.ncond < 2 || c.gap[1] == c.gap[0];
      if (c.ncond > best.ncond || (c.ncond == best.ncond && ((contiguous && !best_contiguous) ||
                                                             (!best_contiguous && uniform && !best_uniform)))) {
         best = c;
         best_contiguous = contiguous;
         best_uniform = uniform;
      }
   }
   // Wide shapes (gaps up to 4, conditions up to MM_F8W_REACH places back; see MM_F8_WIDE / MM_F16_WIDE): where the shapes
   // above leave the streaming loop or the resolver short -- fewer than 3 (8-bit) / 2 (16-bit) conditions -- and a wide choice
   // has more.  Condition 1 of a wide choice is the literal next to the anchor's left (the kernels are compiled for the two
   // gaps only); ties go to equal gaps (one stream of deltas on the hot path), then to short ones, then to the rightmost anchor.
   // ... and where their two per-byte conditions have different gaps (`qz*k`, `q*vk`: two exact SWAR subtractions per dword,
   // 21 operations) while a wide choice has as many conditions: its seven-bit stage costs 14 (round 6, 4 GiB: the scan on
   // the device 0.860 -> 0.803 ms and 0.830 -> 0.798 ms, profiles/r06_wide_shapes.log)
   const bool mixed = u8 && best.ncond >= 2 && best.gap[0] != best.gap[1];
   if ((int)best.ncond < std::min(u8 ? 3 : 2, want) || mixed) {
      auto wide_gap = [&](int i) -> int {
         if (i < 1 || pl.cmp_mask[i] == 0) {
            return 0;
         }
         const int g = -(int)pl.bridge[i];
         return g >= 1 && i - g >= 0 ? g : 0;
      };
      FilterChoice wbest;
      wbest.ncond = 0;
      int wbest_cost = 0;
      for (int A = L - 1; A >= 1; --A) {
         const int g0 = wide_gap(A);
         if (g0 < 1 || g0 > 4) {
            continue;
         }
         FilterChoice c;
         c.ncond = 1; c.iA = (uint32_t)A; c.shape = 0;
         c.pos[0] = (uint32_t)A; c.shift[0] = 0; c.gap[0] = (uint32_t)g0;
         const int wwant = std::max(2, std::min(4, tuning().filter_max_conditions));
         for (int i = A - 1; i >= 1 && (int)c.ncond < wwant; --i) {
            const int g = wide_gap(i), s = A - i;
            if (!g) {
               continue;
            }
            if (u8 && c.ncond == 1 && (s != g0 || g > 4)) {
               break;                                   // (the literal to the anchor's left IS g0 places from it; its own gap is compiled in)
            }
            if (s + g <= (int)MM_F8W_REACH) {
               c.pos[c.ncond] = (uint32_t)i; c.shift[c.ncond] = (uint32_t)s; c.gap[c.ncond] = (uint32_t)g;
               c.ncond++;
            }
         }
         if (u8 && c.ncond < 2) {
            continue;                                   // (one 7-bit condition flags every piece: not a filter)
         }
         const int cost = u8 ? (c.gap[0] == c.gap[1] ? 0 : 8) + (int)(c.gap[0] + c.gap[1]) : (int)c.gap[0];
         if (c.ncond > wbest.ncond || (c.ncond == wbest.ncond && cost < wbest_cost)) {
            wbest = c;
            wbest_cost = cost;
         }
      }
      if (wbest.ncond > best.ncond || (mixed && wbest.ncond >= best.ncond)) {
         *fc = wbest;
         for (uint32_t k = 0; k < 4; k++) {
            const bool used = k < fc->ncond;
            fc->pat[k] = used ? ((uint32_t)pl.expected[fc->pos[k]] & emask) * rep : 0u;
            if (!used) {
               fc->pos[k] = 0; fc->shift[k] = 0; fc->gap[k] = 0;
            }
         }
         fc->shape = u8 ? (uint32_t)MM_F8W_SHAPE(fc->gap[0], fc->gap[1]) : (uint32_t)MM_F16W_SHAPE(fc->gap[0]);
         return true;
      }
   }
   if (best.ncond == 0) {
      fc->ncond = 0;
      return false;
   }
   *fc = best;
   uint32_t mask2 = 0;
   for (uint32_t k = 0; k < 4; k++) {
      const bool used = k < fc->ncond;
      fc->pat[k] = used ? ((uint32_t)pl.expected[fc->pos[k]] & emask) * rep : 0u;
      if (!used) {
         fc->pos[k] = 0; fc->shift[k] = 0; fc->gap[k] = 0;
      }
      else if (fc->gap[k] == 2) {
         mask2 |= 1u << k;
      }
   }
   if (u8) {
      fc->shape = fc->ncond | (best_contiguous ? 0u : (mask2 << 4) | 0x100u);
   }
   else {
      uint32_t kind = 0;
      if (fc->ncond == 2) {
         kind = fc->shift[1] == 2 ? 2u : (fc->gap[1] == 2 ? 1u : 0u);
      }
      fc->shape = fc->ncond | ((mask2 & 1u) << 4) | (kind << 5);
   }
   return true;
}

// Who runs the reference's compare loop on the SWAR survivors.  mm_resolve does (it stages the
// bytes anyway) when there are few false survivors -- 2^-24 / 2^-32 of the positions with 3 / 2
// conditions -- or when the conditions already ARE the whole pattern (short keywords: every
// survivor is a match up to the signed / modular distinction, and has to be resolved anyway).
// Otherwise the loop runs in the filter kernel: that stalls the stream, but keeps millions of
// false survivors away from the resolver.
bool filter_verifies(const mmh_plan_desc &pl, const FilterChoice &fc)
{
   // independent comparisons of the pattern: the simple path also compares position 0 with the
   // last one (bridge L-1), which the deltas in between already imply
   uint32_t compared = 0;
   for (uint32_t i = 0; i < pl.L; i++) {
      compared += (pl.cmp_mask[i] != 0 && pl.bridge[i] < 0) ? 1u : 0u;
   }
   if (fc.ncond >= compared) {
      return false;
   }
   return !((pl.elem_bytes == 1 && fc.ncond >= 3) || (pl.elem_bytes == 2 && fc.ncond >= 2));
}

// Launch geometry of the span kernels: 6 workgroups (24 waves) per CU, a wave streams spans of
// 7 groups = 28 KiB.  The kernels' <= 64 VGPRs would let 8 workgroups per CU stay resident, but
// the streaming rate does not need them -- measured on 4 GiB (u8): spans of 4 / 8 / 16 / 32 groups
// -> 0.732 / 0.695 / 0.711 / 0.712 ms; 1024 .. 4096 workgroups are within 0.5 % of each other
// (round 2, tools/geometry_times2.py + tools/pipeline_geometry.sh: 2048 x 8 -> 0.700 ms,
// 1536 x 7 -> 0.695, 1536 x 8 -> 0.705, 1280 x 7 -> 0.700, 1024 x 8 -> 0.702) -- and the two
// free slots per CU are what lets OTHER kernels run beside it: the tail kernel of the previous
// scan when two scans are in flight (per-scan time 0.717 -> 0.697 ms, below the streaming kernel's
// own duration: the next one starts while the last waves of this one drain), and the RCCL kernel
// of an overlapped gather, which would otherwise wait for this kernel's end.
// The environment knobs are for such experiments only.
static uint64_t filter_max_blocks() { return tuning().filter_blocks; }
static uint32_t filter_groups_per_span() { return tuning().filter_groups_per_span; }

// Kernel launch with optional HIP events riding on the dispatch itself (hipExtLaunchKernelGGL):
// `start` / `stop` take the kernel's begin / end time stamps without the separate marker
// packets of hipEventRecord, which cost ~6 us of stream time each between dependent kernels.
template <class Kernel, class Args>
static void launch_timed(Kernel kernel, dim3 grid, dim3 block, hipStream_t st, hipEvent_t start, hipEvent_t stop, const Args &a)
{
   if (start || stop) {
      hipExtLaunchKernelGGL(kernel, grid, block, 0, st, start, stop, 0, a);
   }
   else {
      hipLaunchKernelGGL(kernel, grid, block, 0, st, a);
   }
}

// span kernel over the whole 4 KiB groups + bounds-checked edge kernel over the ragged end:
// `start` goes to whichever runs first, `stop` to whichever runs last
static MmTileArgs tile_args(const MmGeom &g, const mmh_plan_desc &pl);

template <class Span, class Edge>
static void launch_filter_pair(Span span, Edge edge, hipStream_t st, const MmFilterArgs &a, const MmGeom &g, hipEvent_t start,
                               hipEvent_t stop)
{
   const bool have_span = a.ngroups != 0;
   const bool have_edge = a.edge_first * 16 < g.nbytes;
   if (have_span) {
      uint64_t spans = (a.ngroups + a.groups_per_span - 1) / a.groups_per_span;
      uint64_t blocks = (spans + 3) / 4;
      const uint64_t most = filter_max_blocks();
      if (blocks > most) {
         blocks = most;
      }
      launch_timed(span, dim3((unsigned)blocks), dim3(256), st, start, have_edge ? nullptr : stop, a);
   }
   if (have_edge) {
      launch_timed(edge, dim3(1), dim3(256), st, have_span ? nullptr : start, stop, a);
   }
}

// The streaming kernels of a filter choice's shape (mm_filter_shapes.h: instantiated in the mm_filter_shapes.hip units).
static const ShapeKernels *find_shape(uint32_t elem_bytes, uint32_t shape)
{
   static const ShapeKernels *const units[MM_SHAPE_UNITS] = {shape_unit_0(), shape_unit_1(), shape_unit_2(), shape_unit_3(),
                                                             shape_unit_4(), shape_unit_5(), shape_unit_6()};
   for (const ShapeKernels *unit : units) {
      for (const ShapeKernels *k = unit; k->elem_bytes; k++) {
         if (k->elem_bytes == elem_bytes && k->shape == shape) {
            return k;
         }
      }
   }
   return nullptr;
}

bool shape_known(uint32_t elem_bytes, uint32_t shape) { return find_shape(elem_bytes, shape) != nullptr; }

// Looks every streaming kernel up once: the HIP runtime loads a translation unit's code object at the first use of one of
// its kernels (1.5 - 3 ms per unit, measured inside first scans), and resolves every kernel at its first launch.
void preload_kernels()
{
   static const ShapeKernels *const units[MM_SHAPE_UNITS] = {shape_unit_0(), shape_unit_1(), shape_unit_2(), shape_unit_3(),
                                                             shape_unit_4(), shape_unit_5(), shape_unit_6()};
   hipFuncAttributes attr;
   for (const ShapeKernels *unit : units) {
      for (const ShapeKernels *k = unit; k->elem_bytes; k++) {
         (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(k->span));
         (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(k->edge));
         (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(k->fused));
      }
   }
   (void)hipGetLastError();
}

static const ShapeKernels &shape_kernels(uint32_t elem_bytes, const FilterChoice &fc)
{
   const ShapeKernels *k = find_shape(elem_bytes, fc.shape);
   if (!k) {
      // (choose_filter hands out shapes of the tables only -- tests/test_plan_and_model.py walks them; a kernel of another
      // shape would test other conditions and lose matches silently: stop instead)
      fprintf(stderr, "mmoore: no streaming kernel for %u-bit filter shape %u\n", 8 * elem_bytes, fc.shape);
      abort();
   }
   return *k;
}

// the streaming code's arguments, common to the plain and the fused kernels
template <class A>
static void fill_filter_args(A &a, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc, uint64_t *cand,
                             unsigned long long *ctrl, uint64_t cand_cap, uint32_t groups_per_span)
{
   a.t = tile_args(g, pl); a.iA = fc.iA; a.ncond = fc.ncond;
   a.verify = filter_verifies(pl, fc) ? 1u : 0u;
   for (int k = 0; k < 4; k++) {
      a.pat[k] = fc.pat[k];
      // (wide shapes: the place and the gap themselves, for the rolled second stage)
      a.sh[k] = (fc.shape & 0x200u) ? (fc.shift[k] | fc.gap[k] << 8) : 32u - 8u * fc.shift[k];
   }
   a.cand = cand; a.list_count = ctrl + MM_CTRL_LISTS; a.list_cap = cand_cap / MM_CAND_LISTS;
   a.dom_count = nullptr; a.skip_bits = nullptr;
   a.loud_bits = nullptr; a.loud_tpd = 0; a.loud_tile = 1;
   a.bcand = nullptr; a.bcount = nullptr; a.boverflow = nullptr; a.bshift = 0;
   // whole 4 KiB groups go to the span code, the ragged end to the bounds-checked one
   a.ngroups = g.nbytes / 4096;
   a.groups_per_span = groups_per_span;
   a.edge_first = a.ngroups * 256;
}

void launch_filter(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc,
                   uint64_t *cand, unsigned long long *ctrl, uint64_t cand_cap, hipEvent_t start, hipEvent_t stop,
                   unsigned int *dom_count, const uint32_t *skip_bits)
{
   MmFilterArgs a{};
   // small ROMs are cut fine, as launch_fused does: 64 workgroups fill the 64 candidate lists evenly (a 1 MiB
   // ROM in spans of 7 groups is 10 workgroups = 10 lists, and a dense pattern overflows them) and every CU works
   uint32_t gps = filter_groups_per_span();
   while (gps > 1 && (g.nbytes / 4096) / gps < (uint64_t)MM_CAND_LISTS * MM_WAVES) {
      gps >>= 1;
   }
   fill_filter_args(a, g, pl, fc, cand, ctrl, cand_cap, gps);
   a.dom_count = dom_count; a.skip_bits = skip_bits;
   const ShapeKernels &k = shape_kernels(pl.elem_bytes, fc);
   launch_filter_pair(k.span, k.edge, st, a, g, start, stop);
}

// the forward engine's pre-pass: the streaming filter over the whole ROM, survivors into the tile bitmap (zeroed by the caller)
static void launch_loud(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc, unsigned long long *ctrl,
                        uint32_t *bits, uint32_t tpd, uint32_t tile)
{
   MmFilterArgs a{};
   fill_filter_args(a, g, pl, fc, nullptr, ctrl, 0, filter_groups_per_span());
   a.loud_bits = bits; a.loud_tpd = tpd; a.loud_tile = tile;
   const ShapeKernels &k = shape_kernels(pl.elem_bytes, fc);
   launch_filter_pair(k.span, k.edge, st, a, g, nullptr, nullptr);
}

// ---- the fused scan kernel (mm_fused.h) ---------------------------------------------------------

// Workgroups of mm_scan_fused that are resident at once on this device, for sure: the grid barrier
// needs all of them.  The occupancy API can be one block per CU high at this kernel's SGPR count
// (MI355X_MICROARCH.md, correctness boundaries), so one is taken off, and never more than 5 per CU
// (16 waves per CU already stream at the full read bandwidth: 1024 x 8 measured like 2048 x 8;
// 1280 workgroups = 5120 waves resolve the bench ROM's 4223 candidates in one round).
// compute units of the current device (cached per device: the attribute query costs a few hundred nanoseconds)
static unsigned device_cus()
{
   // (mmh_scan_multi runs one host thread per device through here: relaxed atomics -- racing threads store the same value)
   static std::atomic<int> cached[64];
   int device = 0;
   if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) {
      return 256;
   }
   int have = cached[device].load(std::memory_order_relaxed);
   if (have == 0) {
      int cus = 0;
      have = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0 ? cus : 256;
      cached[device].store(have, std::memory_order_relaxed);
   }
   return (unsigned)have;
}

static unsigned fused_resident_blocks()
{
   static const unsigned blocks = [] {
      int device = 0, cus = 0, per_cu = 0;
      if (hipGetDevice(&device) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, find_shape(1, 4)->fused, 64 * MM_WAVES, 0) != hipSuccess) {
         return 0u;
      }
      per_cu = std::min(5, per_cu - 1);
      return per_cu > 0 ? (unsigned)(per_cu * cus) : 0u;
   }();
   return blocks;
}

static void fill_tail_args(MmFusedArgs &a, const ResolveBuffers &rb, uint64_t base_offset, uint32_t max_candidates,
                           uint64_t *host_result, uint64_t *dev_result, uint32_t max_rank, uint64_t seq)
{
   a.out_cap = rb.out_cap; a.out = rb.out; a.tiles_walked = rb.ctrl + MM_CTRL_TILES;
   a.base_offset = base_offset; a.max_candidates = max_candidates;
   a.mid_off = rb.mid_off; a.mid_hi = rb.mid_hi; a.mid_set = rb.mid_set; a.mid_slot = rb.mid_slot;
   a.mid_count = reinterpret_cast<unsigned int *>(rb.ctrl + MM_CTRL_MID);
   a.flag_bits = nullptr;
   a.ctrl = rb.ctrl; a.host_result = host_result; a.dev_result = dev_result; a.max_rank = max_rank;
   a.ctrl_words = (uint32_t)(ctrl_bytes() / sizeof(unsigned long long));
   a.seq = seq;
   a.timeout_ticks = 20000000;                    // 200 ms of the 100 MHz wall clock
}

// ROMs up to this size take the single-launch kernel (tools/fused_probe.py: 128 KiB .. 2 MiB 11 us
// on the device against 20; 16 MiB with candidates 32 against 27; 4 GiB 782 against 738)
static uint64_t fused_max_bytes() { return 4ull << 20; }

bool fused_applies(const MmGeom &g) { return g.nbytes <= fused_max_bytes() && fused_resident_blocks() != 0; }

bool launch_fused(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc, const ResolveBuffers &rb,
                  uint64_t base_offset, uint32_t max_candidates, uint64_t *host_result, uint64_t *dev_result, uint32_t max_rank,
                  uint64_t seq, hipEvent_t start, hipEvent_t stop)
{
   const unsigned resident = fused_resident_blocks();
   if (resident == 0) {
      return false;
   }
   MmFusedArgs a{};
   const uint64_t ngroups = g.nbytes / 4096;
   // small ROMs are cut fine so that every wave has a span (a 128 KiB ROM: 32 waves of 4 KiB instead of 4 of 32 KiB)
   uint32_t gps = filter_groups_per_span();
   while (gps > 1 && ngroups / gps < (uint64_t)resident * MM_WAVES) {
      gps >>= 1;
   }
   fill_filter_args(a, g, pl, fc, rb.cand, rb.ctrl, rb.cand_cap, gps);
   fill_tail_args(a, rb, base_offset, max_candidates, host_result, dev_result, max_rank, seq);
   a.has_edge = a.edge_first * 16 < g.nbytes ? 1u : 0u;
   const uint64_t spans = (a.ngroups + gps - 1) / gps;
   const unsigned blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((spans + MM_WAVES - 1) / MM_WAVES, resident));
   launch_timed(shape_kernels(pl.elem_bytes, fc).fused, dim3(blocks), dim3(64 * MM_WAVES), st, start, stop, a);
   return true;
}

// mm_scan_tail behind the streaming kernel: every candidate resolved, ranked and published, header and
// completion flag written -- one launch in place of mm_resolve + mm_rank_count + mm_rank_scatter
void launch_tail(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc, const ResolveBuffers &rb,
                 uint64_t base_offset, uint32_t max_candidates, uint64_t *host_result, uint64_t *dev_result, uint32_t max_rank,
                 uint64_t seq, hipEvent_t stop)
{
   MmFusedArgs a{};
   fill_filter_args(a, g, pl, fc, rb.cand, rb.ctrl, rb.cand_cap, filter_groups_per_span());
   fill_tail_args(a, rb, base_offset, max_candidates, host_result, dev_result, max_rank, seq);
   a.has_edge = 0;
   // one wave per candidate for up to 8 K of them in one round (the count is only known on the device)
   // (5 waves per SIMD measured best: 6 and 8 spill and run 4-15 us longer; 1280 .. 4096 workgroups: no difference)
   if (pl.L > MM_RESOLVER_MAX_KEYWORD) {
      launch_timed(mm_scan_tail<5, true>, dim3(tuning().tail_blocks), dim3(64 * MM_WAVES), st, nullptr, stop, a);
      return;
   }
   launch_timed(mm_scan_tail<5>, dim3(tuning().tail_blocks), dim3(64 * MM_WAVES), st, nullptr, stop, a);
}

// ---- big ROMs: bucketed candidate store + mm_scan_tail2 (mm_tail2.h) ----------------------------------------------

BucketGeom bucket_geom(uint64_t nbytes)
{
   BucketGeom b;
   b.shift = MM_MIN_BUCKET_SHIFT;
   while (((nbytes + 15) >> b.shift) >= MM_MAX_BUCKETS) {      // (+ 15: survivors in the padding of the last 16-byte chunk)
      b.shift++;
   }
   b.nb = (uint32_t)(((nbytes + 15) >> b.shift) + 1);
   return b;
}
size_t bucket_cand_bytes() { return (size_t)MM_MAX_BUCKETS * MM_BUCKET_CAP * sizeof(uint64_t); }
size_t bucket_count_bytes() { return (size_t)MM_MAX_BUCKETS * sizeof(unsigned int); }

template <class A>
static void fill_bucket_args(A &a, const MmGeom &g, const ResolveBuffers &rb)
{
   const BucketGeom b = bucket_geom(g.nbytes);
   a.bcand = rb.bcand; a.bcount = rb.bcount;
   a.boverflow = rb.ctrl + MM_CTRL_BOVERFLOW; a.bshift = b.shift;
}

void launch_filter_buckets(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc, const ResolveBuffers &rb,
                           hipEvent_t start, hipEvent_t stop)
{
   MmFilterArgs a{};
   uint32_t gps = filter_groups_per_span();
   while (gps > 1 && (g.nbytes / 4096) / gps < (uint64_t)MM_CAND_LISTS * MM_WAVES) {
      gps >>= 1;
   }
   fill_filter_args(a, g, pl, fc, rb.cand, rb.ctrl, rb.cand_cap, gps);
   fill_bucket_args(a, g, rb);
   const ShapeKernels &k = shape_kernels(pl.elem_bytes, fc);
   launch_filter_pair(k.span, k.edge, st, a, g, start, stop);
}

void launch_tail2(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc, const ResolveBuffers &rb,
                  uint64_t base_offset, uint32_t max_candidates, uint64_t *host_result, uint64_t *dev_result, uint64_t seq,
                  hipEvent_t stop, unsigned tail_blocks)
{
   MmFusedArgs a{};
   fill_filter_args(a, g, pl, fc, rb.cand, rb.ctrl, rb.cand_cap, filter_groups_per_span());
   fill_tail_args(a, rb, base_offset, max_candidates, host_result, dev_result, MM_MAX_PUBLISH, seq);
   fill_bucket_args(a, g, rb);
   a.nbuckets = bucket_geom(g.nbytes).nb;
   // (never beyond what the host re-poisons between scans, note_dirty_slots in mm_capi.hip: a direct slot above that
   // could pass for an offset after a late PCIe write)
   a.direct_limit = std::min<uint32_t>(MM_DIRECT_PUBLISH, MM_MAX_RANK_SORT);
   a.has_edge = 0;
   // Several candidates per wave (mm_resolve_sub): two for keywords of up to 16 symbols, four up to 13, eight up to 4
   // (profiles/r04_candidate_density.log; 64 / 48 more registers than one per wave: 6 / 5 waves per SIMD instead of 8,
   // which costs nothing -- with 7 the compiler spills and every row of the log was slower).
   // Test hooks (tests/test_gpu_tail_groups.py runs every width on every keyword length): MMOORE_TAIL_SUB=1: one per wave
   // (2, 4: at most that many); MMOORE_TAIL_QUAD_MAXL: the longest keyword that gets four.
   static const int sub = [] { const char *e = getenv("MMOORE_TAIL_SUB"); return e && *e ? atoi(e) : 8; }();
   static const int quad_maxl = [] { const char *e = getenv("MMOORE_TAIL_QUAD_MAXL"); const int v = e && *e ? atoi(e) : 13; return v > 13 ? 13 : v; }();
   // The grid of a scan with the device to itself: as many workgroups as are resident at once at the variant's waves per
   // SIMD (the device's CUs x occupancy; 256 CUs on the MI355X) -- with 2048 for all of them a quarter of the 6-per-SIMD variants' waves started when the
   // first ones ended and a dense search's tail took half as long again (`water`, 90 K candidates in one launch: 96 -> 61 us;
   // `and` 126 -> 73 with 1280 for the 5-per-SIMD variant; profiles/r04_candidate_density_tail_grid.log).
   const int occ = (sub >= 8 && pl.L <= 4) ? 5 : ((sub >= 2 && pl.L <= 16) || pl.L > MM_RESOLVER_MAX_KEYWORD) ? 6 : 8;
   const dim3 grid(tail_blocks ? tail_blocks : device_cus() * (unsigned)occ), block(64 * MM_WAVES);
   // (grouped from an eighth of the grid's waves on: at the bench's 4223 candidates the tail takes 24 us instead of 29 and a
   // synchronous scan 0.768 ms instead of 0.79 -- fewer walks to the buckets; any threshold between 0 and 4096 measures
   // the same, profiles/r04_tail_blocks_sweep.log.  Below that a wave's candidates would only wait for each other where
   // they fall back to the one-per-wave resolver)
   a.group_min = grid.x * MM_WAVES / 8;
   if (pl.L > MM_RESOLVER_MAX_KEYWORD) {
      launch_timed(mm_scan_tail2<6, 64, true>, grid, block, st, nullptr, stop, a);   // (keywords of 65 .. 128 symbols: two phases per lane)
   }
   else if (sub >= 8 && pl.L <= 4) {
      launch_timed(mm_scan_tail2<5, 8>, grid, block, st, nullptr, stop, a);    // (96 registers: with 80 it spills, and is slower)
   }
   else if (sub >= 4 && (int)pl.L <= quad_maxl) {
      launch_timed(mm_scan_tail2<6, 16>, grid, block, st, nullptr, stop, a);
   }
   else if (sub >= 2 && pl.L <= 16) {
      launch_timed(mm_scan_tail2<6, 32>, grid, block, st, nullptr, stop, a);
   }
   else {
      launch_timed(mm_scan_tail2<8, 64>, grid, block, st, nullptr, stop, a);
   }
}

static MmTileArgs tile_args(const MmGeom &g, const mmh_plan_desc &pl)
{
   MmTileArgs t{};
   t.g = g; t.plan = pl;
   const uint32_t D = pl.L - 1;
   t.inv_d = ((1u << 20) + D - 1) / D;
   t.inv_d32 = (uint32_t)((1ull << 32) / D) + 1u;
   t.block_shift = ~0u;
   if (g.block_bytes && (g.block_bytes & (g.block_bytes - 1)) == 0) {
      t.block_shift = (uint32_t)__builtin_ctzll(g.block_bytes);
   }
   t.skip_bloom = t.skip_bloom2 = t.skip_bloom3 = 0;
   for (uint32_t k = 0; k < pl.n_skip; k++) {
      const uint32_t d = (uint32_t)pl.skip_diff[k];
      t.skip_bloom |= 1ull << (d & 63);
      t.skip_bloom2 |= 1ull << ((d >> 6) & 63);
      t.skip_bloom3 |= 1ull << ((d >> 12) & 63);
   }
   return t;
}

void launch_resolve(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const ResolveBuffers &rb,
                    uint64_t base_offset, uint32_t max_candidates, uint32_t *flag_bits)
{
   MmResolveArgs a{};
   a.t = tile_args(g, pl);
   a.cand = rb.cand; a.list_count = rb.ctrl + MM_CTRL_LISTS; a.list_cap = rb.cand_cap / MM_CAND_LISTS;
   a.total_out = rb.ctrl + MM_CTRL_TOTAL;
   a.out = rb.out; a.out_cap = rb.out_cap; a.tiles_walked = rb.ctrl + MM_CTRL_TILES;
   a.base_offset = base_offset; a.max_candidates = max_candidates;
   a.mid_off = rb.mid_off; a.mid_hi = rb.mid_hi; a.mid_set = rb.mid_set; a.mid_slot = rb.mid_slot;
   a.mid_count = reinterpret_cast<unsigned int *>(rb.ctrl + MM_CTRL_MID);
   a.flag_bits = flag_bits;
   if (pl.L > MM_RESOLVER_MAX_KEYWORD) {
      hipLaunchKernelGGL(mm_resolve_long, dim3(tuning().resolve_blocks), dim3(64 * MM_WAVES), 0, st, a);
      return;
   }
   hipLaunchKernelGGL(mm_resolve, dim3(tuning().resolve_blocks), dim3(64 * MM_WAVES), 0, st, a);
}

// Second phase of a scan, launched only when mm_resolve left candidates over (the host sees
// the count in the published header): mm_resolve2, then mm_hard_resolve for what even that
// could not settle.  Two always-launched no-op kernels cost ~5 us each per scan.
void launch_leftovers(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const ResolveBuffers &rb, uint64_t base_offset)
{
   MmResolve2Args m{};
   m.t = tile_args(g, pl);
   m.mid_off = rb.mid_off; m.mid_hi = rb.mid_hi; m.mid_set = rb.mid_set; m.mid_slot = rb.mid_slot;
   m.mid_count = reinterpret_cast<unsigned int *>(rb.ctrl + MM_CTRL_MID);
   m.out = rb.out; m.tiles_walked = rb.ctrl + MM_CTRL_TILES; m.base_offset = base_offset;
   m.hard_off = rb.hard_off; m.hard_hi = rb.hard_hi; m.hard_set = rb.hard_set; m.hard_slot = rb.hard_slot;
   m.hard_count = reinterpret_cast<unsigned int *>(rb.ctrl + MM_CTRL_HARD);
   hipLaunchKernelGGL(mm_resolve2, dim3(256), dim3(64 * MM_WAVES), 0, st, m);

   MmHardArgs h{};
   h.t = m.t;
   h.hard_off = rb.hard_off; h.hard_hi = rb.hard_hi; h.hard_set = rb.hard_set; h.hard_slot = rb.hard_slot;
   h.hard_count = reinterpret_cast<unsigned int *>(rb.ctrl + MM_CTRL_HARD);
   h.overflow = reinterpret_cast<unsigned int *>(rb.ctrl + MM_CTRL_HARD) + 1;
   h.done = reinterpret_cast<unsigned int *>(rb.ctrl + MM_CTRL_DONE);
   h.scratch = rb.scratch;
   h.out = rb.out; h.tiles_walked = rb.ctrl + MM_CTRL_TILES;
   h.base_offset = base_offset;
   hipLaunchKernelGGL(mm_hard_resolve, dim3(MM_HARD_PARTS, MM_HARD_CAP), dim3(64 * MM_WAVES), 0, st, h);
}

size_t hard_scratch_bytes() { return (size_t)MM_HARD_CAP * MM_HARD_MAX_TILES * MM_RMAXD; }
size_t hard_cap() { return MM_HARD_CAP; }
size_t mid_cap() { return MM_MID_CAP; }
size_t ctrl_bytes() { return MM_CTRL_WORDS * sizeof(uint64_t); }
static_assert(MM_CTRL_DONE * sizeof(uint64_t) + MM_HARD_CAP * sizeof(unsigned int) <= MM_CTRL_ARRIVE_BARRIER * sizeof(uint64_t), "ctrl layout");
size_t rank_partials_bytes(uint32_t max_n) { return (size_t)MM_RANK_SLICES * max_n * sizeof(uint32_t); }

DenseGeom dense_geom(const MmGeom &g, uint64_t listed_domains)
{
   DenseGeom d;
   d.ndom = listed_domains ? listed_domains : (g.whole ? 1 : g.nblocks * g.S);
   int64_t most = 0;
   for (uint32_t p = 0; p < (g.whole ? 1u : g.S); p++) {
      int64_t nv = mm_domain_nv(g, 0, p);                 // block 0 is the largest kind of block
      most = nv > most ? nv : most;
   }
   d.tpd = (uint32_t)((most + MM_FWD_TILE - 1) / MM_FWD_TILE);
   // A wave maps the tiles of its batch one after the other, and a tile full of matches takes it ~60 us (dependent LDS
   // chains with nobody to hide them): a flood a MiB wide in a small input -- the split pipeline's "forward engine on the
   // flooded blocks", a few flagged domains -- is 32 batches of 16 tiles = a millisecond on 32 waves while the device
   // idles.  Inputs that do not fill the device go in batches of 4 (round 6: 1.3 -> 0.5 ms for 5 MiB around a ramp).
   d.batch = d.ndom * d.tpd < 48u * 1024u ? 4u : (uint32_t)MM_FWD_BATCH;
   d.bpd = (d.tpd + d.batch - 1) / d.batch;
   // [ticket, pad][one look-back word per batch], then one map per batch
   d.status_bytes = (((size_t)d.ndom * d.bpd + 2) * sizeof(unsigned long long) + 255) & ~(size_t)255;
   // ... the pre-pass's tile bitmap (whole-ROM passes only; two words of slack: a batch's bits are read as two words) ...
   d.loud_bytes = listed_domains ? 0 : ((((size_t)d.ndom * d.tpd + 31) / 32 + 2) * sizeof(uint32_t) + 255) & ~(size_t)255;
   d.maps_bytes = d.status_bytes + d.loud_bytes + (size_t)d.ndom * d.bpd * (g.L > MM_MAXD ? MMH_MAX_KEYWORD : MM_MAXD);
   return d;
}

// the single-pass forward engine (mm_forward.h); db.maps holds [ticket + look-back words][batch maps]
static void launch_forward(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const DenseGeom &dg, const DenseBuffers &db,
                           uint64_t base_offset, const uint32_t *dom_list)
{
   MmForwardArgs a{};
   a.t = tile_args(g, pl);
   a.ndom = dg.ndom; a.dom_list = dom_list; a.tpd = dg.tpd; a.bpd = dg.bpd; a.batch = dg.batch;
   const uint64_t nbatches = dg.ndom * dg.bpd;
   a.ticket = reinterpret_cast<unsigned long long *>(db.maps);
   a.status = a.ticket + 2;
   a.agg = db.maps + dg.status_bytes + dg.loud_bytes; // [batches][MM_MAXD or MMH_MAX_KEYWORD] (dense_geom sized it)
   a.out = db.out; a.list_count = db.ctrl + MM_CTRL_LISTS; a.list_cap = db.out_cap / MM_CAND_LISTS;
   a.base_offset = base_offset;
   // the table-driven jump path: 8-bit elements, first compare against an element 1..4 to the left
   a.fast = 0; a.i1 = a.g1 = a.has2 = a.i2 = a.g2 = 0;

   int i1 = (int)pl.L - 1;
   while (i1 >= 0 && pl.cmp_mask[i1] == 0) {
      i1--;
   }
   if (pl.elem_bytes == 1 && i1 >= 1 && pl.bridge[i1] <= -1 && pl.bridge[i1] >= -4 && i1 + pl.bridge[i1] >= 0) {
      a.fast = 1; a.i1 = (uint32_t)i1; a.g1 = (uint32_t)(-pl.bridge[i1]);
      int i2 = i1 - 1;
      while (i2 >= 0 && pl.cmp_mask[i2] == 0) {
         i2--;
      }
      if (i2 >= 1 && pl.bridge[i2] < 0 && i2 + pl.bridge[i2] >= 0) {
         a.has2 = 1; a.i2 = (uint32_t)i2; a.g2 = (uint32_t)(-pl.bridge[i2]);
      }
   }
   // Which tiles have something to report (the sweep of mm_forward.h): 8-bit searches on the table path find out themselves, batch
   // by batch, with the streaming filter's first two conditions; 16-bit searches get a bitmap from the streaming filter itself,
   // run over the whole ROM before the forward kernel (whole-ROM passes only).  Keywords without a SWAR test: no sweep.
   a.loud = nullptr;
   a.loud_shape = 0; a.loud_iA = 0; a.loud_pat[0] = a.loud_pat[1] = 0; a.loud_sh1 = 0;
   FilterChoice fc;
   const bool sweeps = choose_filter(pl, &fc);
   if (sweeps && a.fast && !(fc.shape & 0x200u)) {          // (its batch sweep knows the one-dword shapes only)
      const uint32_t nc = fc.ncond < 2 ? fc.ncond : 2;
      uint32_t mask2 = fc.gap[0] == 2 ? 1u : 0u;
      if (nc == 2) {
         mask2 |= fc.gap[1] == 2 ? 2u : 0u;
         a.loud_pat[1] = fc.pat[1];
         a.loud_sh1 = 32u - 8u * fc.shift[1];
      }
      a.loud_shape = 0x100u | (mask2 << 4) | nc;
      a.loud_iA = fc.iA;
      a.loud_pat[0] = fc.pat[0];
   }
   // (the sweep lives on chains that merge.  8-bit: one delta in 25 is in the skip table and moves its chain to another phase;
   // 16-bit: practically none is, every jump is the default one and the phases never mix -- unless the wildcard loop caps the
   // jumps (wst): measured on 1 GiB, 16-bit: wildcard keyword 4.98 -> 1.04 ms, plain keyword 5.79 -> 6.45 with a pre-pass that
   // bought nothing.  So: only with capped jumps.)
   bool capped = false;
   for (uint32_t i = 0; i < pl.L; i++) {
      capped = capped || pl.wst[i] != 255;
   }
   const bool prepass = sweeps && !a.fast && capped && !dom_list && dg.loud_bytes != 0;
   // plain 16-bit keywords: quiet tiles (every jump the default one) are recognised from their bytes and never mapped
   // (mm_fwd_exceptional16): the first compare must be against the left neighbour, on the signed path, no capped jumps
   a.quiet16 = 0;
   a.q16_bloom[0] = a.q16_bloom[1] = a.q16_bloom[2] = 0;
   if (pl.elem_bytes == 2 && !a.fast && !capped && i1 >= 1 && pl.bridge[i1] == -1 && pl.cmp_mask[i1] == 0xFFFFFFFFu &&
       pl.default_skip == (int32_t)pl.L - 1) {
      a.quiet16 = 1;
      a.i1 = (uint32_t)i1;
      auto add = [&](int32_t d) {
         const uint32_t x = (uint32_t)d;
         a.q16_bloom[0] |= 1u << (x & 31);
         a.q16_bloom[1] |= 1u << ((x >> 5) & 31);
         a.q16_bloom[2] |= 1u << ((x >> 10) & 31);
      };
      add(pl.expected[i1]);
      for (uint32_t k = 0; k < pl.n_skip; k++) {
         add(pl.skip_diff[k]);
      }
   }
   {
      // A workgroup's block of batches (mm_forward.h): four -- one per wave, as round 2 handed them out.  Bigger blocks are
      // slower, and badly so (1 GiB, wildcard keyword: 4 / 8 / 16 / 32 batches -> 0.31 / 0.67 / 0.94 / 1.19 ms): the
      // workgroups then work 4 x 32 KiB at a time out of regions 8 x ... 32 x 32 KiB apart, and at any moment the reads of
      // the whole grid land on a fraction of the memory channels.
      a.chunk = 4u;
   }
   (void)hipMemsetAsync(db.maps, 0, dg.status_bytes + (prepass ? dg.loud_bytes : 0), st);
   if (prepass) {
      uint32_t *bits = reinterpret_cast<uint32_t *>(db.maps + dg.status_bytes);
      launch_loud(st, g, pl, fc, db.ctrl, bits, dg.tpd, (uint32_t)MM_FWD_TILE);
      a.loud = bits;
   }
   int device = 0, cus = 256;
   (void)hipGetDevice(&device);
   (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
   const unsigned blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nbatches + MM_WAVES - 1) / MM_WAVES, (uint64_t)cus * 6));
   const bool wide = pl.L > MM_MAXD;                  // more phases than the narrow maps hold
   if (pl.elem_bytes == 1) {
      if (wide) {
         hipLaunchKernelGGL((mm_forward<1, MMH_MAX_KEYWORD>), dim3(blocks), dim3(64 * MM_WAVES), 0, st, a);
      }
      else {
         hipLaunchKernelGGL((mm_forward<1, MM_MAXD>), dim3(blocks), dim3(64 * MM_WAVES), 0, st, a);
      }
   }
   else if (wide) {
      hipLaunchKernelGGL((mm_forward<2, MMH_MAX_KEYWORD>), dim3(blocks), dim3(64 * MM_WAVES), 0, st, a);
   }
   else {
      hipLaunchKernelGGL((mm_forward<2, MM_MAXD>), dim3(blocks), dim3(64 * MM_WAVES), 0, st, a);
   }
}

void launch_dense(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const DenseGeom &dg, const DenseBuffers &db,
                  uint64_t base_offset, const uint32_t *dom_list)
{
   launch_forward(st, g, pl, dg, db, base_offset, dom_list);
}

void launch_chain_seq(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, uint64_t *out,
                      unsigned long long *out_count, uint64_t out_cap, uint64_t base_offset)
{
   MmSeqArgs a{};
   a.g = g; a.plan = pl; a.out = out; a.out_count = out_count; a.out_cap = out_cap; a.base_offset = base_offset;
   uint64_t ndom = g.whole ? 1 : g.nblocks * g.S;
   uint64_t blocks = (ndom + 63) / 64;
   if (blocks > 4096) {
      blocks = 4096;
   }
   hipLaunchKernelGGL(mm_chain_seq, dim3((unsigned)blocks), dim3(64), 0, st, a);
}

void launch_rank_sort(hipStream_t st, const uint64_t *in, unsigned long long *ctrl, int count_index, uint64_t cap,
                      uint32_t max_n, uint32_t *partials, uint64_t *host_result, uint64_t *dev_result, hipEvent_t stop,
                      bool keep_leftovers)
{
   const uint32_t keep = keep_leftovers ? 1u : 0u;
   hipLaunchKernelGGL(mm_rank_count, dim3(16, MM_RANK_SLICES), dim3(256), 0, st, in, ctrl + count_index, cap, max_n,
                      partials);
   const uint32_t ctrl_words = (uint32_t)(ctrl_bytes() / sizeof(unsigned long long));
   if (stop) {
      hipExtLaunchKernelGGL(mm_rank_scatter, dim3(64), dim3(256), 0, st, nullptr, stop, 0, in, ctrl, count_index, cap, max_n,
                            partials, host_result, dev_result, ctrl_words, keep);
   }
   else {
      hipLaunchKernelGGL(mm_rank_scatter, dim3(64), dim3(256), 0, st, in, ctrl, count_index, cap, max_n, partials,
                         host_result, dev_result, ctrl_words, keep);
   }
}

void launch_publish_list(hipStream_t st, const uint64_t *src, uint64_t *dst, uint32_t n, unsigned long long *arrive,
                         unsigned long long *flag, unsigned long long seq)
{
   const unsigned blocks = std::max(1u, std::min(256u, (n + 1023u) / 1024u));
   hipLaunchKernelGGL(mm_publish_list, dim3(blocks), dim3(256), 0, st, src, dst, n, arrive, flag, seq);
}

void launch_synth(hipStream_t st, uint8_t *rom, uint64_t nbytes, uint64_t seed, uint64_t base_offset)
{
   hipLaunchKernelGGL(mm_synth_fill, dim3(2048), dim3(256), 0, st, rom, nbytes, seed, base_offset);
}

// The forward engine's MM_CAND_LISTS output lists, one behind the other in `packed` (what the ordering takes): block (l, part)
// copies its share of list l to where the lists in front of it end.  One launch instead of a copy per list.
__global__ __launch_bounds__(256) void mm_pack_lists(const uint64_t *lists, uint64_t list_cap, const unsigned long long *count, uint64_t *packed)
{
   const uint32_t l = blockIdx.x;
   uint64_t before = 0;
   for (uint32_t k = 0; k < l; k++) {
      const uint64_t n = count[k * MM_LIST_STRIDE];
      before += n < list_cap ? n : list_cap;
   }
   uint64_t n = count[l * MM_LIST_STRIDE];
   n = n < list_cap ? n : list_cap;                 // (an overflowed list: the caller sizes the lists anew and scans again)
   for (uint64_t i = (uint64_t)blockIdx.y * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.y * 256) {
      packed[before + i] = lists[(uint64_t)l * list_cap + i];
   }
}

void launch_pack_lists(hipStream_t st, const uint64_t *lists, uint64_t list_cap, const unsigned long long *list_count, uint64_t *packed)
{
   hipLaunchKernelGGL(mm_pack_lists, dim3(MM_CAND_LISTS, 16), dim3(256), 0, st, lists, list_cap, list_count, packed);
}

void launch_gather(hipStream_t st, const uint8_t *rom, uint64_t nbytes, const uint64_t *offsets, uint64_t n, uint32_t each,
                   uint8_t *out)
{
   const uint64_t total = n * each;
   const unsigned blocks = (unsigned)std::min<uint64_t>((total + 255) / 256, 4096);
   if (blocks) {
      hipLaunchKernelGGL(mm_gather, dim3(blocks), dim3(256), 0, st, rom, nbytes, offsets, n, each, out);
   }
}

void launch_pattern_fill(hipStream_t st, uint8_t *rom, uint64_t first, uint64_t nbytes, int value, int ramp)
{
   hipLaunchKernelGGL(mm_pattern_fill, dim3(256), dim3(256), 0, st, rom, first, nbytes, value, ramp);
}

} // namespace mm
