// SPDX-License-Identifier: GPL-3.0-or-later
// mm_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the relative-search engine.
//
// Pipeline of one scan (one stream; the host polls a word in pinned memory for the end):
//
//   mm_filter_u8 / mm_filter_u16   HBM-bound streaming pass over the whole ROM.
//        Coalesced 16 B/lane loads; two stages: the first two SWAR conditions
//        (pattern deltas) on every byte position, 4 positions per 32-bit VALU
//        op; pieces with a stage-1 hit get all conditions (up to 4) from a
//        re-read; the survivors -- CANDIDATES: positions where the reference's
//        compare loop may report a match IF its chain visits them -- are
//        appended wave-aggregated to 64 lists.
//   mm_scan_tail  (mm_fused.h, mm_tiles.h)   one wavefront per candidate.  The
//        reference is not a complete matcher: it only tests the positions its
//        skip chain visits.  The wave verifies the compare loop
//        at the candidate and decides "is it on the chain of its domain" exactly,
//        by pulling the set of acceptable chain phases (phase = position mod
//        (L-1)) back through the phase maps of one or two short windows until
//        the set is empty, full, or the domain start (phase 0) is reached; it
//        also counts the candidates with a smaller offset -- its place in the
//        ascending list -- and writes the verdict there, in pinned host memory
//        and in HBM (for the multi-GPU gather).  The last workgroup adds the
//        header and raises the scan's sequence number.
//   ROMs of up to 4 MiB: mm_scan_fused runs both stages in ONE launch (grid barrier).
//   Second phase, only when candidates are left over (host decides from the published
//   counters): mm_resolve2 -> mm_hard_resolve (mm_tiles.h) -> mm_rank_count / mm_rank_scatter.
//   The lanes of mmh_scan_submit run the same two kernels; MMOORE_FUSED=0 brings back the round-1
//   chain mm_filter -> mm_resolve -> mm_rank_count -> mm_rank_scatter everywhere.
//
// Other engines: mm_forward (mm_forward.h), the candidate-free forward engine, for
// inputs the per-candidate path does not suit and for keywords beyond 32 symbols;
// mm_chain_seq, one lane per domain walking the chain literally, as an on-device
// cross-check.
//
// No MFMA anywhere: the path is integer byte comparison (BASELINE.json).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <utility>

#include "mm_internal.h"
#include "mm_kernels.h"

// --------------------------------------------------------------------------
// small helpers
// --------------------------------------------------------------------------

__device__ __forceinline__ uint32_t mm_alignbit(uint32_t hi, uint32_t lo, uint32_t shift)
{
   return __builtin_amdgcn_alignbit(hi, lo, shift);   // ({hi,lo} >> shift)[31:0]
}

// per-byte (a - b) mod 256 on four packed bytes
__device__ __forceinline__ uint32_t mm_bytesub(uint32_t a, uint32_t b)
{
   uint32_t t = (a | 0x80808080u) - (b & 0x7F7F7F7Fu);
   return t ^ (~(a ^ b) & 0x80808080u);
}

// bit 7 of a byte set when (at least) that byte or a lower one is zero; used as a
// superset filter, survivors are verified exactly
__device__ __forceinline__ uint32_t mm_haszero8(uint32_t v)
{
   return (v - 0x01010101u) & ~v & 0x80808080u;
}

__device__ __forceinline__ uint32_t mm_haszero16(uint32_t v)
{
   return (v - 0x00010001u) & ~v & 0x80008000u;
}

__device__ __forceinline__ uint32_t mm_sub16x2(uint32_t a, uint32_t b)
{
   typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
   u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
   return __builtin_bit_cast(uint32_t, (u16x2)(x - y));             // v_pk_sub_u16
}

__device__ __forceinline__ uint32_t mm_bswap16x2(uint32_t v)
{
   return __builtin_amdgcn_perm(v, v, 0x02030001u);                 // swap bytes inside each half
}

// element j of the domain that starts at byte `start`
__device__ __forceinline__ int mm_elem(const MmGeom &g, uint64_t start, int64_t j)
{
   const uint8_t *p = g.rom + start + (uint64_t)j * g.S;
   if (g.S == 1) {
      return p[0];
   }
   int lo = p[0], hi = p[1];
   return g.big_endian ? (lo << 8 | hi) : (hi << 8 | lo);
}

__device__ __forceinline__ int mm_skip_sparse(const mmh_plan_desc &pl, int d)
{
   int s = pl.default_skip;
   for (uint32_t k = 0; k < pl.n_skip; k++) {
      if (pl.skip_diff[k] == d) {
         s = pl.skip_val[k];
      }
   }
   return s;
}

// The reference's compare loop at alignment j (unified form of
// monkey_moore.cpp:347-407 and :449-543).  Returns the jump; *matched says
// whether the loop reported a match.
template <class Reader>
__device__ __forceinline__ int mm_step(const mmh_plan_desc &pl, Reader rd, int64_t j, bool *matched)
{
   for (int i = (int)pl.L - 1; i >= 0; --i) {
      int c = rd(j + i);
      int p = rd(j + i + pl.bridge[i]);
      int d = c - p;
      if (((uint32_t)(d ^ pl.expected[i]) & pl.cmp_mask[i]) != 0) {
         int s = mm_skip_sparse(pl, d);
         s = s < 1 ? 1 : s;
         int w = pl.wst[i];
         *matched = false;
         return s < w ? s : w;
      }
   }
   *matched = true;
   return (int)pl.match_jump;
}

// which domain a byte offset belongs to; returns false when the offset is not a
// valid alignment of any domain (tail of the file, 16-bit odd boundary, ...)
__device__ __forceinline__ bool mm_locate(const MmGeom &g, uint64_t o, uint64_t *b, uint32_t *p, int64_t *j)
{
   if (g.whole) {
      if (o % g.S) {
         return false;
      }
      *b = 0; *p = 0; *j = (int64_t)(o / g.S);
   }
   else {
      if (o >= g.nbytes) {
         return false;                          // a SWAR survivor in the padding behind the ROM: block o / B does not exist
      }
      uint64_t blk = o / g.block_bytes;
      uint64_t r = o - blk * g.block_bytes;
      *b = blk; *p = (uint32_t)(r % g.S); *j = (int64_t)(r / g.S);
   }
   return *j < mm_domain_nv(g, *b, *p);
}

// first active lane's value (wave uniform)
__device__ __forceinline__ uint64_t mm_uniform64_k(uint64_t v)
{
   return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
          ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}

// A word another workgroup of the SAME launch reads (the fused scan kernel resolves candidates in
// the launch that found them): stored write-through (sc1), cdna guideline 16 R1.  Candidates are
// a few thousand stores per scan, so the plain kernels use it as well.
__device__ __forceinline__ void mm_store_shared(uint64_t *p, uint64_t v)
{
   __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave-aggregated append (ballot + prefix popcount -> one atomic per wave)
__device__ __forceinline__ void mm_append(uint64_t *list, unsigned long long *count, uint64_t cap, bool want, uint64_t value)
{
   unsigned long long mask = __ballot(want);
   if (mask == 0) {
      return;
   }
   unsigned lane = __lane_id();
   unsigned long long base = 0;
   int leader = __ffsll((long long)mask) - 1;
   if ((int)lane == leader) {
      base = atomicAdd(count, (unsigned long long)__popcll(mask));
   }
   base = __shfl(base, leader);
   if (want) {
      unsigned long long slot = base + __popcll(mask & ((1ull << lane) - 1));
      if (slot < cap) {
         mm_store_shared(list + slot, value);
      }
   }
}

// verify the full match predicate at byte offset o and append it as a candidate
__device__ __forceinline__ bool mm_is_candidate(const MmGeom &g, const mmh_plan_desc &pl, int64_t o)
{
   if (o < 0) {
      return false;
   }
   uint64_t b; uint32_t p; int64_t j;
   if (!mm_locate(g, (uint64_t)o, &b, &p, &j)) {
      return false;
   }
   uint64_t start = mm_domain_start(g, b, p);
   bool matched;
   mm_step(pl, [&](int64_t k) { return mm_elem(g, start, k); }, j, &matched);
   return matched;
}

// --------------------------------------------------------------------------
// streaming filter, 8-bit elements
// --------------------------------------------------------------------------
//
// One 16-byte chunk per lane (lane l of a wave reads bytes [16l, 16l+16) of a
// 1 KiB piece -> fully coalesced dwordx4).  For every byte t the SWAR code
// evaluates up to four necessary conditions for a match anchored at t (starting
// at t - iA), condition k being
//     (x[t-s_k] - x[t-s_k-g_k]) mod 256 == pat[k]
// the delta of keyword position iA-s_k against the literal g_k (1, or 2 over a
// wildcard) to its left; signed equality on the simple path implies modular
// equality, so this is a superset on both reference paths.  On random bytes
// 2^(-8 n) of the positions pass n conditions.  All four on every dword cost 21
// VALU instructions per dword and made the kernel VALU-limited; the streaming
// loop therefore tests two (14 per dword) and hands the ~1.6 % of the pieces
// with a hit to a rolled second stage that tests all of them.

// wave-uniform constants shared by the tile kernels (kernel arguments -> SGPRs)
struct MmTileArgs {
   MmGeom g;
   mmh_plan_desc plan;
   uint32_t inv_d;          // ceil(2^20 / D): x / D == (x * inv_d) >> 20 for x < 4096
   uint32_t inv_d32;        // floor(2^32 / D) + 1: quotient estimate for 32-bit x, at most one too big
   uint32_t block_shift;    // log2(block_bytes) when that is a power of two, else ~0
   uint64_t skip_bloom;     // bit (d & 63) set for every listed bad-character diff
   uint64_t skip_bloom2;    // ... bit ((d >> 6) & 63) ...
   uint64_t skip_bloom3;    // ... and bit ((d >> 12) & 63): a 16-bit delta that is in none of the lists passes all three once in ~1000
};

// Arguments of the streaming code.  The span / edge kernels take this struct; the fused scan kernel
// (mm_fused.h) takes a struct with the same member names (the streaming code is templated on it).
struct MmFilterArgs {
   MmTileArgs t;           // geometry + plan (the tile constants are only used by the fused kernel's resolver)
   uint32_t pat[4];        // condition k, replicated over the SWAR lanes
   uint32_t sh[4];         // 8-bit shapes with run-time shifts: v_alignbit amount 32 - 8 s_k of condition k
   uint32_t iA;            // keyword index of the element whose delta is pat[0]
   uint32_t ncond;
   uint32_t verify;        // 1: run the full compare loop on survivors here; 0: leave it to mm_resolve

   uint64_t *cand;         // MM_CAND_LISTS lists of candidate byte offsets, list_cap entries each
   unsigned long long *list_count;   // their counters, MM_LIST_STRIDE words apart (mm_internal.h)
   uint64_t list_cap;
   // candidate floods (engine mode, see run_candidate_floods in mm_capi.hip); both null in a normal scan
   unsigned int *dom_count;          // count pass: candidates per domain instead of the lists
   const uint32_t *skip_bits;        // filtered pass: candidates of flagged domains are dropped
   // the forward engine's pre-pass (mm_forward.h): no lists at all -- every survivor sets the bit of its forward-engine tile
   // (tile (b S + p) loud_tpd + j / loud_tile of the bitmap: "a position of this tile may pass the compare loop"); else null
   uint32_t *loud_bits;
   uint32_t loud_tpd, loud_tile;
   // bucketed store (mm_internal.h MM_BUCKET_*; null: the lists above): big ROMs, read by mm_scan_tail2
   uint64_t *bcand;                  // [nb][MM_BUCKET_CAP] candidate byte offsets
   unsigned int *bcount;             // [nb] members of every bucket
   unsigned long long *boverflow;    // appends that found their bucket full
   uint32_t bshift;                  // log2 of a bucket's width in bytes

   uint64_t ngroups;       // span kernel: number of whole 4 KiB groups it covers
   uint32_t groups_per_span;
   uint64_t edge_first;    // edge kernel: 16-byte chunks [edge_first, nchunks)
};

// a workgroup always appends to the same list: no two lists share an atomic address
template <class A>
__device__ __forceinline__ void mm_cand_append(const A &a, bool want, uint64_t off)
{
   const uint32_t c = blockIdx.x & (MM_CAND_LISTS - 1);
   mm_append(a.cand + (uint64_t)c * a.list_cap, a.list_count + c * MM_LIST_STRIDE, a.list_cap, want, off);
}

// Where the survivors go: the candidate lists -- or, in the two passes that deal with candidate
// floods, a per-domain count (one atomic per wave and domain) / the lists minus the flagged domains.
template <class A>
__device__ __forceinline__ void mm_cand_emit(const A &a, bool want, uint64_t off)
{
   if (a.loud_bits) {                                      // wave uniform
      uint64_t b = 0; uint32_t p = 0; int64_t j = 0;
      if (want && mm_locate(a.t.g, off, &b, &p, &j)) {
         const uint64_t tile = (a.t.g.whole ? 0 : (b * a.t.g.S + p) * a.loud_tpd) + (uint64_t)j / a.loud_tile;
         atomicOr(a.loud_bits + (tile >> 5), 1u << (tile & 31));
      }
      return;
   }
   if (a.dom_count || a.skip_bits) {                       // wave uniform
      uint64_t b = 0; uint32_t p = 0; int64_t j = 0;
      const bool located = want && mm_locate(a.t.g, off, &b, &p, &j);
      const uint32_t dom = located ? (uint32_t)(b * a.t.g.S + p) : 0u;
      if (a.dom_count) {
         unsigned long long todo = __ballot(located);
         while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t d = (uint32_t)__shfl((int)dom, leader);
            const unsigned long long same = __ballot(located && dom == d);
            if ((int)__lane_id() == leader) {
               atomicAdd(a.dom_count + d, (unsigned int)__popcll(same));
            }
            todo &= ~same;
         }
         return;
      }
      want = located && ((a.skip_bits[dom >> 5] >> (dom & 31)) & 1u) == 0;
   }
   mm_cand_append(a, want, off);
}

// All survivors of a piece at once: every lane brings `cnt` candidates; one atomic for the wave
// reserves their slots and returns this lane's first one (the caller stores while slot < cap).
// A padding run that matches the keyword wholesale has 16 survivors per lane and piece: one
// round trip per piece instead of sixteen.
template <class A>
__device__ __forceinline__ uint64_t *mm_cand_reserve(const A &a, uint32_t cnt, uint32_t *first, uint32_t *room)
{
   // returns (wave uniform) the address of the wave's first reserved slot; *first = this lane's
   // first slot relative to it, *room = how many of the wave's slots exist at all (list capacity)
   const uint32_t c = blockIdx.x & (MM_CAND_LISTS - 1);
   uint32_t incl = cnt;
#pragma unroll
   for (int d = 1; d < 64; d <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, d);
      incl += (int)__lane_id() >= d ? v : 0u;
   }
   const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
   unsigned long long base = 0;
   if (__lane_id() == 0) {
      base = atomicAdd(a.list_count + c * MM_LIST_STRIDE, (unsigned long long)total);
   }
   const uint64_t b = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base) |
                      ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32);
   *first = incl - cnt;
   *room = b >= a.list_cap ? 0u : (uint32_t)((a.list_cap - b) > 0xFFFFFFFFull ? 0xFFFFFFFFull : (a.list_cap - b));
   return a.cand + (uint64_t)c * a.list_cap + b;
}

// Bucketed store: all survivors of the wave's piece (1 KiB, aligned, starting at byte piece0: one bucket) at once.
// Every lane brings `cnt` candidates; returns (wave uniform) the address of the wave's first reserved slot,
// *first = this lane's first slot relative to it, *room = how many of the wave's slots exist (bucket capacity).
template <class A>
__device__ __forceinline__ uint64_t *mm_bucket_reserve(const A &a, uint64_t piece0, uint32_t cnt, uint32_t *first, uint32_t *room)
{
   uint32_t incl = cnt;
#pragma unroll
   for (int d = 1; d < 64; d <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, d);
      incl += (int)__lane_id() >= d ? v : 0u;
   }
   const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
   *first = incl - cnt;
   *room = 0;
   if (total == 0) {
      return a.bcand;
   }
   const uint64_t b = piece0 >> a.bshift;
   unsigned int base = 0;
   if (__lane_id() == 0) {
      base = atomicAdd(a.bcount + b, total);
      // (no counter per super-bucket here: 256 of them are 8 cache lines, and 64 K fire-and-forget atomics on 8 lines
      // cost the streaming kernel 65 us -- mm_scan_tail2's workgroups sum the bucket counters instead)
      if (base + total > MM_BUCKET_CAP) {
         atomicAdd(a.boverflow, 1ull);
      }
   }
   base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
   *room = base >= MM_BUCKET_CAP ? 0u : MM_BUCKET_CAP - base;
   return a.bcand + b * MM_BUCKET_CAP + base;
}

// ---- the wave's survivor queue (bucketed scans, round 4) ---------------------------------------------------------------
// mm_bucket_reserve costs the streaming wave one RETURNING atomic per flagged piece -- a round trip to L2 of a microsecond
// or two during which the wave streams nothing.  Harmless at one candidate per MiB; in the script of a ROM, where a common
// word sits every few hundred bytes, every piece of a span is flagged: 28 round trips on a span that streams in 28 us, and
// the kernel is as slow as its slowest round (profiles/r03_candidate_density.log: 0.72 -> 0.80-0.82 ms at 90 K candidates).
// So survivors are parked in LDS -- MM_QCAP entries per wave, candidate offsets in the order they were found -- and their
// buckets are reserved later, between two spans once the queue is half full and at the wave's end: one atomic per RUN of
// entries of the same bucket (a span lies in one or two buckets of a big ROM), all runs of 64 entries with one wave
// instruction (mm_queue_flush).  Survivors that do not fit (a flood) take the old way.
// (Round 4 emptied the queue behind every span.  A wave's stores and atomics count on vmcnt like its loads, in order: the next
// span's first group cannot be waited for before they are done, and with a two-condition keyword -- `th*s`, any three-symbol
// word: 2^-16 of random positions survive, 65 K per 4 GiB -- the waves that met five or six survivors ended the kernel late:
// 0.77-0.82 ms per 4 GiB of random bytes against 0.73-0.75 now, tools/filter_ab.py --config THIS, profiles/r05_dense_ab.log.)
constexpr uint32_t MM_QCAP = 128;                 // entries per wave: 1 KiB of LDS (4 KiB per workgroup)
constexpr int MM_FILTER_WAVES = 4;
// Queue AND fill count live in LDS, the wave's number is worked out where it is needed: nothing of this is held in a
// register across the streaming loop (the 16-bit kernel sits at 71 VGPRs = 7 waves per SIMD and at the SGPR limit: two more
// live scalars spilled into VGPR lanes and cost it a wave per SIMD).
struct MmSurvivorQueue {
   uint64_t entry[MM_FILTER_WAVES][MM_QCAP];
   uint32_t count[MM_FILTER_WAVES];
};

// this wave's queue (computed on the spot: the asm keeps the compiler from hoisting it out of the streaming loop)
__device__ __forceinline__ int mm_queue_wave()
{
   uint32_t t = threadIdx.x;
   asm volatile("" : "+v"(t));
   return __builtin_amdgcn_readfirstlane((int)(t >> 6)) & (MM_FILTER_WAVES - 1);
}

// queue entries [0, n) to their buckets.  key_bytes: candidate offset + key_bytes = the position the filter keyed on, which is
// what orders candidates into buckets (monotone in the offset, and inside the piece the wave read it from -- 16-bit odd
// stream: at most one byte in front of it, which still keeps bucket order = offset order).
// 64 entries per step, one per lane: the lanes at the head of a RUN of entries of one bucket reserve the run's slots -- all
// runs of the step with ONE wave instruction, one round trip to L2 however many buckets the entries belong to (round 5:
// the queue is no longer emptied behind every span, see mm_stream_u8, and holds the survivors of spans far apart).
template <class A>
__device__ __forceinline__ void mm_queue_flush(const A &a, MmSurvivorQueue &Q, uint32_t key_bytes, uint32_t at_least = 1)
{
   const int w = mm_queue_wave();
   const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)Q.count[w]);
   if (n < at_least) {
      return;                                                              // (at_least >= 1: nothing parked, nothing to do)
   }
   Q.count[w] = 0;
   const uint64_t *q = Q.entry[w];
   const uint32_t lane = __lane_id();
   for (uint32_t pos = 0; pos < n; pos += 64) {                            // wave uniform
      const uint32_t valid = n - pos < 64u ? n - pos : 64u;
      const bool in = lane < valid;
      const uint64_t e = in ? q[pos + lane] : 0ull;
      const uint32_t b = in ? (uint32_t)((e + key_bytes) >> a.bshift) : 0xFFFFFFFFu;
      const uint32_t b_prev = (uint32_t)__builtin_amdgcn_update_dpp((int)~b, (int)b, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
      const unsigned long long heads = __ballot(in && b != b_prev);       // (lane 0: its `old` is ~b -- always a head)
      const unsigned long long upto = heads & ((2ull << lane) - 1ull);    // the heads at or below this lane
      const unsigned long long above = heads & ~((2ull << lane) - 1ull);
      const uint32_t head = in ? 63u - (uint32_t)__builtin_clzll(upto) : lane;
      const uint32_t next = above ? (uint32_t)__builtin_ctzll(above) : valid;
      unsigned int base = 0;
      if (in && head == lane) {
         const uint32_t run = next - lane;
         base = atomicAdd(a.bcount + b, run);
         if (base + run > MM_BUCKET_CAP) {
            atomicAdd(a.boverflow, 1ull);
         }
      }
      base = (unsigned int)__shfl((int)base, (int)head);
      const uint32_t slot = base + (lane - head);
      if (in && slot < MM_BUCKET_CAP) {
         mm_store_shared(a.bcand + (uint64_t)b * MM_BUCKET_CAP + slot, e);
      }
   }
}

// exclusive prefix of cnt over the wave's lanes; *total = the wave's sum (wave uniform)
__device__ __forceinline__ uint32_t mm_wave_prefix(uint32_t cnt, uint32_t *total)
{
   uint32_t incl = cnt;
#pragma unroll
   for (int d = 1; d < 64; d <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, d);
      incl += (int)__lane_id() >= d ? v : 0u;
   }
   *total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
   return incl - cnt;
}

__device__ __forceinline__ uint4 mm_load_chunk(const uint8_t *rom, uint64_t nbytes, uint64_t byte0)
{
   if (byte0 + 16 <= nbytes) {
      return *reinterpret_cast<const uint4 *>(rom + byte0);
   }
   uint32_t w[4] = {0, 0, 0, 0};
   for (uint64_t k = byte0; k < nbytes; k++) {
      w[(k - byte0) >> 2] |= (uint32_t)rom[k] << (8 * ((k - byte0) & 3));
   }
   return make_uint4(w[0], w[1], w[2], w[3]);
}

// SHAPE of an 8-bit filter: bits 0-2 = number of conditions NC (1..4); bits 4-7 = MASK2, bit k
// set when condition k compares over a wildcard (gap 2: x[t-s] - x[t-s-2]) instead of with the
// left neighbour (gap 1); bit 8 = the byte shifts s_k come from the kernel arguments (any
// 1..3 with s_k + gap_k <= 4) instead of being k.  SHAPE = NC is the contiguous run of
// adjacent literals (no wildcard between the last NC+1 keyword symbols used).
#define MM_F8_NC(shape) ((shape) & 7)
#define MM_F8_MASK2(shape) (((shape) >> 4) & 15)
#define MM_F8_RT(shape) (((shape) >> 8) & 1)

// One dword: a byte of the result is zero where all conditions of SHAPE hold for the match
// anchored at that byte; p1 / p2 = gap-1 / gap-2 byte deltas of the previous dword
template <int SHAPE>
__device__ __forceinline__ uint32_t mm_f8_z(uint32_t w, uint32_t wprev, uint32_t &p1, uint32_t &p2,
                                            const uint32_t (&pat)[4], const uint32_t (&sh)[4])
{
   constexpr int NC = MM_F8_NC(SHAPE), M2 = MM_F8_MASK2(SHAPE);
   constexpr bool RT = MM_F8_RT(SHAPE) != 0;
   const uint32_t d1 = M2 != (1 << NC) - 1 ? mm_bytesub(w, mm_alignbit(w, wprev, 24)) : 0u;
   const uint32_t d2 = M2 != 0 ? mm_bytesub(w, mm_alignbit(w, wprev, 16)) : 0u;
   uint32_t z = ((M2 & 1) ? d2 : d1) ^ pat[0];
#pragma unroll
   for (int k = 1; k < NC; k++) {
      const bool two = ((M2 >> k) & 1) != 0;
      const uint32_t amount = RT ? sh[k] : (uint32_t)(32 - 8 * k);
      z |= mm_alignbit(two ? d2 : d1, two ? p2 : p1, amount) ^ pat[k];
   }
   p1 = d1;
   p2 = d2;
   return z;
}

// hit flags (bit 7 of a byte) of one dword
template <int SHAPE>
__device__ __forceinline__ uint32_t mm_f8_hits(uint32_t w, uint32_t wprev, uint32_t &p1, uint32_t &p2,
                                               const uint32_t (&pat)[4], const uint32_t (&sh)[4])
{
   return mm_haszero8(mm_f8_z<SHAPE>(w, wprev, p1, p2, pat, sh));
}

// The first two conditions of a shape: what the streaming loop tests on every byte.  They let
// 2^-16 of random positions through, i.e. some lane of a wave sees a hit in ~1.6 % of its
// 1 KiB pieces; only then are all conditions evaluated (mm_f8_chunk) -- 14 instead of 21 VALU
// operations per dword on the hot path, which is what separates the filter from the pure-read
// bandwidth of the chip.
#define MM_F8_STAGE1(shape) (((shape) & 0x130) | (MM_F8_NC(shape) < 2 ? MM_F8_NC(shape) : 2))

// non-zero when some byte of the chunk passes the stage-1 conditions
template <int SHAPE>
__device__ __forceinline__ uint32_t mm_f8_chunk_any(const uint4 &w, uint32_t back, const uint32_t (&pat)[4],
                                                    const uint32_t (&sh)[4])
{
   constexpr int S1 = MM_F8_STAGE1(SHAPE);
   uint32_t p1, p2;
   if constexpr (S1 == 2) {
      // The contiguous two-condition stage (BASELINE's keywords): of the previous dword's deltas only the LAST byte's is ever
      // looked at (v_alignbit ... 24), and that one is a single byte subtract -- round 5: 1 VALU instruction instead of the 6
      // of a whole SWAR subtract, 5 of the ~62 a 16-byte chunk costs (the upper bytes of p1 are never read: left undefined)
      asm("v_sub_u32_sdwa %0, %1, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_2" : "=v"(p1) : "v"(back));
      p2 = 0;
   }
   else {
      p1 = mm_bytesub(back, back << 8);
      p2 = mm_bytesub(back, back << 16);
   }
   const uint32_t z0 = mm_f8_z<S1>(w.x, back, p1, p2, pat, sh);
   const uint32_t z1 = mm_f8_z<S1>(w.y, w.x, p1, p2, pat, sh);
   const uint32_t z2 = mm_f8_z<S1>(w.z, w.y, p1, p2, pat, sh);
   const uint32_t z3 = mm_f8_z<S1>(w.w, w.z, p1, p2, pat, sh);
   // mm_haszero8 on all four, with the final mask shared
   return (((z0 - 0x01010101u) & ~z0) | ((z1 - 0x01010101u) & ~z1) | ((z2 - 0x01010101u) & ~z2) | ((z3 - 0x01010101u) & ~z3)) &
          0x80808080u;
}

// hit flags of a 16-byte chunk; `back` = the dword in front of it
template <int SHAPE>
__device__ __forceinline__ uint32_t mm_f8_chunk(const uint4 &w, uint32_t back, const uint32_t (&pat)[4],
                                                const uint32_t (&sh)[4], uint32_t (&h)[4])
{
   uint32_t p1 = mm_bytesub(back, back << 8);       // gap-1 deltas of bytes -3..-1 (byte -4's is not needed)
   uint32_t p2 = mm_bytesub(back, back << 16);      // gap-2 deltas of bytes -2..-1 (s + gap <= 4)
   h[0] = mm_f8_hits<SHAPE>(w.x, back, p1, p2, pat, sh);
   h[1] = mm_f8_hits<SHAPE>(w.y, w.x, p1, p2, pat, sh);
   h[2] = mm_f8_hits<SHAPE>(w.z, w.y, p1, p2, pat, sh);
   h[3] = mm_f8_hits<SHAPE>(w.w, w.z, p1, p2, pat, sh);
   return h[0] | h[1] | h[2] | h[3];
}

// ---- WIDE 8-bit shapes (round 6): runs of two and three wildcards ---------------------------------------------------
// The reference's wildcard loop compares every literal with the literal before it, however many wildcards lie between
// (monkey_moore.cpp:425-546; its own vectors: `But**er`, `**に*行きますか`, tests/test_monkey_moore.cpp:194-246).  The shapes
// above know gaps of 1 and 2 inside ONE dword of look-back (s + gap <= 4): `qz**mb` kept one condition, `q**k**x` none.
// A wide shape keeps, on every byte, conditions 0 and 1 of
//     (x[t - s_k] - x[t - s_k - g_k]) mod 256 == pat[k],      g_k in 1..4,   s_0 = 0,  s_1 = g_0
// (condition 1 is the literal next to the anchor's left: its place follows from the anchor's gap, so (g_0, g_1) is all a
// kernel is compiled for: 16 instantiations), over TWO dwords of look-back (s_1 + g_1 <= 8: the second one is lane l-1's
// w.z through one more DPP move).  Stage 1 compares the LOW SEVEN BITS of the deltas only -- (a | 0x80) - (b & 0x7F) per
// byte cannot borrow, and its low seven bits are those of a - b: 3 VALU operations instead of the 6 of an exact SWAR
// subtraction, 14 per dword for two conditions with different gaps where the exact form costs 20 -- and lets 2^-14 instead
// of 2^-16 of random positions through; stage 2 is exact.
// Stage 2 (flagged pieces only) takes ALL conditions of the choice, any s_k and g_k with s_k + g_k <= MM_F8W_REACH, with
// run-time places: condition k is a SWAR subtraction of two UNALIGNED 16-byte loads of the piece, x[.. - s_k] and
// x[.. - s_k - g_k] (L2 / TCP hits: the piece was streamed a moment ago) -- no register shuffling, so one rolled copy
// serves every shape.  sh[k] of the arguments carries s_k | g_k << 8 for these shapes.
#define MM_F8_WIDE(shape) (((shape) >> 9) & 1)
#define MM_F8W_G0(shape) ((((shape) >> 4) & 3) + 1)
#define MM_F8W_G1(shape) ((((shape) >> 6) & 3) + 1)
#define MM_F8W_SHAPE(g0, g1) (0x200 | 2 | (((g0) - 1) << 4) | (((g1) - 1) << 6))
#define MM_F8W_BACK2(shape) (MM_F8W_G0(shape) + MM_F8W_G1(shape) > 4)      // stage 1 looks at the dword two in front
constexpr uint32_t MM_F8W_REACH = 16;

// the dword K bytes in front of `cur` (K = 0..8) out of cur and the two dwords before it
template <int K>
__device__ __forceinline__ uint32_t mm_bytes_back(uint32_t cur, uint32_t prev, uint32_t prev2)
{
   static_assert(K >= 0 && K <= 8, "two dwords of look-back");
   if constexpr (K == 0) {
      return cur;
   }
   else if constexpr (K < 4) {
      return mm_alignbit(cur, prev, 32 - 8 * K);
   }
   else if constexpr (K == 4) {
      return prev;
   }
   else if constexpr (K < 8) {
      return mm_alignbit(prev, prev2, 32 - 8 * (K - 4));
   }
   else {
      return prev2;
   }
}

// per byte: low seven bits = those of (a - b); bit 7 is not meaningful
__device__ __forceinline__ uint32_t mm_bytesub7(uint32_t a, uint32_t b)
{
   return (a | 0x80808080u) - (b & 0x7F7F7F7Fu);
}

// non-zero when some byte of the chunk passes the two stage-1 conditions on their low seven bits;
// back / back2 = the dwords one / two in front of the chunk
template <int SHAPE>
__device__ __forceinline__ uint32_t mm_f8w_chunk_any(const uint4 &w, uint32_t back, uint32_t back2, const uint32_t (&pat)[4])
{
   constexpr int G0 = MM_F8W_G0(SHAPE), G1 = MM_F8W_G1(SHAPE);
   const uint32_t r[6] = {back2, back, w.x, w.y, w.z, w.w};
   uint32_t acc = 0x80808080u;                                      // bit 7 of a byte stays set while no dword had a hit there
   if constexpr (G0 == G1) {
      // one stream of deltas: condition 1 is condition 0's delta G0 bytes earlier
      uint32_t dprev = mm_bytesub7(back, mm_bytes_back<G0>(back, back2, 0u));
#pragma unroll
      for (int j = 2; j < 6; j++) {
         const uint32_t d = mm_bytesub7(r[j], mm_bytes_back<G0>(r[j], r[j - 1], 0u));
         const uint32_t z = (d ^ pat[0]) | (mm_bytes_back<G0>(d, dprev, 0u) ^ pat[1]);
         acc &= (z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;                    // bit 7 set where the seven bits are not all zero
         dprev = d;
      }
   }
   else {
#pragma unroll
      for (int j = 2; j < 6; j++) {
         const uint32_t xa = mm_bytes_back<G0>(r[j], r[j - 1], r[j - 2]);
         const uint32_t xb = mm_bytes_back<G0 + G1>(r[j], r[j - 1], r[j - 2]);
         const uint32_t z = (mm_bytesub7(r[j], xa) ^ pat[0]) | (mm_bytesub7(xa, xb) ^ pat[1]);
         acc &= (z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
      }
   }
   return ~acc & 0x80808080u;
}

// 16 bytes of the ROM from any byte address, zeros where the ROM is not (in front of it, behind it)
__device__ __forceinline__ uint4 mm_load16_at(const uint8_t *rom, uint64_t nbytes, int64_t at)
{
   if (at >= 0 && (uint64_t)at + 16 <= nbytes) {
      struct __attribute__((packed, aligned(1))) Unaligned { uint32_t x, y, z, w; };
      const Unaligned v = *reinterpret_cast<const Unaligned *>(rom + at);   // global_load_dwordx4 at a byte address
      return make_uint4(v.x, v.y, v.z, v.w);
   }
   uint32_t w[4] = {0, 0, 0, 0};
   for (int k = 0; k < 16; k++) {
      const int64_t i = at + k;
      if (i >= 0 && (uint64_t)i < nbytes) {
         w[k >> 2] |= (uint32_t)rom[i] << (8 * (k & 3));
      }
   }
   return make_uint4(w[0], w[1], w[2], w[3]);
}

// exact hit flags of the 16-byte chunk at byte0 for ALL conditions of a wide choice (run-time places)
template <class A>
__device__ __forceinline__ uint32_t mm_f8w_chunk(const A &a, uint64_t byte0, uint32_t (&h)[4])
{
   uint32_t z[4] = {0, 0, 0, 0};
   for (uint32_t k = 0; k < a.ncond; k++) {                         // wave uniform
      const uint32_t s = a.sh[k] & 0xFFu, g = a.sh[k] >> 8;
      const uint4 x = mm_load16_at(a.t.g.rom, a.t.g.nbytes, (int64_t)byte0 - (int64_t)s);
      const uint4 y = mm_load16_at(a.t.g.rom, a.t.g.nbytes, (int64_t)byte0 - (int64_t)(s + g));
      z[0] |= mm_bytesub(x.x, y.x) ^ a.pat[k];
      z[1] |= mm_bytesub(x.y, y.y) ^ a.pat[k];
      z[2] |= mm_bytesub(x.z, y.z) ^ a.pat[k];
      z[3] |= mm_bytesub(x.w, y.w) ^ a.pat[k];
   }
#pragma unroll
   for (int j = 0; j < 4; j++) {
      h[j] = mm_haszero8(z[j]);
   }
   return h[0] | h[1] | h[2] | h[3];
}

__device__ __forceinline__ uint32_t mm_f8_pack(const uint32_t (&h)[4])
{
   return (h[0] >> 7) | (h[1] >> 6) | (h[2] >> 5) | (h[3] >> 4);   // bit 8*b + k <-> dword k, byte b
}

// verify the survivors flagged in `bits` (bit 8*b + k of this lane's chunk at byte `chunk0`)
// and append the real candidates; the whole wave takes part (ballots inside)
template <class A>
__device__ __forceinline__ void mm_f8_survivors(const A &a, uint64_t chunk0, uint32_t bits, MmSurvivorQueue *Q = nullptr)
{
   const bool head = __ballot(bits != 0 && chunk0 < MMH_MAX_KEYWORD) != 0;   // a survivor may lie in front of the anchor
   if (a.bcount) {
      // bucketed store: what must be checked per survivor (in front of the ROM / the compare loop) first, lane by
      // lane, then everything that is left in one go
      if (a.verify || head) {
         uint32_t keep = 0;
         for (uint32_t todo = bits; todo; todo &= todo - 1) {
            const int bit = __ffs((int)todo) - 1;
            const int64_t o = (int64_t)(chunk0 + 4 * (bit & 3) + (bit >> 3)) - (int64_t)a.iA;
            if (a.verify ? mm_is_candidate(a.t.g, a.t.plan, o) : (o >= 0)) {
               keep |= 1u << bit;
            }
         }
         bits = keep;
      }
      if (Q) {
         // the streaming kernel: park them in the wave's queue (no round trip to L2 here)
         uint32_t total;
         uint32_t at = mm_wave_prefix((uint32_t)__popc(bits), &total);
         if (total == 0) {
            return;
         }
         const int w = mm_queue_wave();
         const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)Q->count[w]);
         if (len + total <= MM_QCAP) {
            at += len;
            while (bits) {
               const int bit = __ffs((int)bits) - 1;
               bits &= bits - 1;
               Q->entry[w][at++] = chunk0 + 4 * (bit & 3) + (bit >> 3) - a.iA;
            }
            Q->count[w] = len + total;
            return;
         }
         // (the queue is full -- a span with more than MM_QCAP survivors is a flood: straight to the bucket)
      }
      uint32_t slot, room;
      uint64_t *list = mm_bucket_reserve(a, mm_uniform64_k(chunk0), (uint32_t)__popc(bits), &slot, &room);
      while (bits) {
         const int bit = __ffs((int)bits) - 1;
         bits &= bits - 1;
         if (slot < room) {
            mm_store_shared(list + slot, chunk0 + 4 * (bit & 3) + (bit >> 3) - a.iA);
         }
         slot++;
      }
      return;
   }
   bool batch = !a.verify && !a.dom_count && !a.skip_bits && !a.loud_bits && !head;
   if (!a.verify && !head && !batch && !a.t.g.whole && !a.loud_bits) {
      // Flood handling passes, 8-bit: a piece lies in ONE block (= domain) nearly always.  Then
      // the count pass adds the piece's survivors with one atomic and the filtered pass drops
      // or keeps them wholesale, instead of locating every survivor on its own.
      const uint64_t blk = (chunk0 - a.iA) / a.t.g.block_bytes;
      const uint64_t blk_last = (chunk0 + 15 - a.iA) / a.t.g.block_bytes;
      const uint64_t lead = mm_uniform64_k(blk);
      if (__ballot(bits != 0 && (blk != lead || blk_last != lead)) == 0) {
         if (a.dom_count) {
            uint32_t n = (uint32_t)__popc(bits);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
               n += (uint32_t)__shfl_xor((int)n, d);
            }
            if (__lane_id() == 0 && n) {
               atomicAdd(a.dom_count + lead, n);
            }
            return;
         }
         if ((a.skip_bits[lead >> 5] >> (lead & 31)) & 1u) {
            return;                              // a flagged domain: the forward engine has it
         }
         batch = true;
      }
   }
   // the usual case (the resolver verifies, not the ROM's first bytes): everything in one go
   if (batch) {
      uint32_t slot, room;
      uint64_t *list = mm_cand_reserve(a, (uint32_t)__popc(bits), &slot, &room);
      while (bits) {
         const int bit = __ffs((int)bits) - 1;
         bits &= bits - 1;
         if (slot < room) {
            mm_store_shared(list + slot, chunk0 + 4 * (bit & 3) + (bit >> 3) - a.iA);
         }
         slot++;
      }
      return;
   }
   while (__ballot(bits != 0) != 0) {
      bool want = false;
      uint64_t off = 0;
      if (bits) {
         const int bit = __ffs((int)bits) - 1;
         bits &= bits - 1;
         const int64_t t = (int64_t)(chunk0 + 4 * (bit & 3) + (bit >> 3));
         const int64_t o = t - (int64_t)a.iA;
         // With enough SWAR conditions practically every survivor is a real candidate and the
         // resolver (which stages the bytes anyway) verifies it; running the dependent-load
         // compare loop here would stall this wave's stream for microseconds per survivor.
         if (a.verify ? mm_is_candidate(a.t.g, a.t.plan, o) : (o >= 0)) {
            want = true;
            off = (uint64_t)o;
         }
      }
      mm_cand_emit(a, want, off);
   }
}


// Which span a wave works on next: round i hands span i*nwaves + ((wave + 2731 i) mod nwaves) to this wave -- neighbouring
// waves stream neighbouring spans, and anything periodic in the ROM (candidates at every 32 MiB, say) lands on a different
// wave every round instead of piling up on a few.  (Round 3 also had spans drawn through tickets, for the scans-in-flight
// case where workgroups start up to 0.7 ms apart: measured worse in every mix -- the ticket's round trip at every span
// boundary and the loss of "neighbouring waves stream neighbouring spans" cost more than the balance gains, 0.742 against
// 0.705 ms per 4 GiB, profiles/r03_lane_gate_and_span_tickets.log -- and its cursor held 16 SGPRs across the streaming
// loop of a kernel that sits at the SGPR limit: removed in round 4.)  32-bit throughout: a span is >= 4 KiB.
struct MmSpanCursor {
   uint32_t round, nwaves, wave, nspans;
};

template <class A>
__device__ __forceinline__ MmSpanCursor mm_span_cursor(const A &, uint64_t wave, uint64_t nwaves, uint64_t nspans)
{
   MmSpanCursor c;
   c.round = 0; c.nwaves = (uint32_t)nwaves; c.wave = (uint32_t)wave; c.nspans = (uint32_t)nspans;
   return c;
}

// the next span of this wave (wave uniform), false when there is none left
__device__ __forceinline__ bool mm_next_span(MmSpanCursor &c, uint64_t *span)
{
   for (; (uint64_t)c.round * c.nwaves < c.nspans; c.round++) {
      const uint64_t s = (uint64_t)c.round * c.nwaves + (c.wave + c.round * 2731u) % c.nwaves;
      if (s < c.nspans) {
         c.round++;
         *span = s;
         return true;
      }
   }
   return false;
}

// bounds-checked version for the ragged end of the ROM (everything behind the last
// whole 4 KiB group): one chunk per lane per iteration, look-back by a second load
// (nblocks workgroups take part, this one is number `block` of them)
template <int SHAPE, class A>
__device__ __forceinline__ void mm_edge_u8(const A &a, uint32_t block, uint32_t nblocks)
{
   const uint64_t nchunks = (a.t.g.nbytes + 15) / 16;
   const uint64_t stride = (uint64_t)nblocks * blockDim.x;
   // same trip count in every lane: the ballots must see whole waves
   const uint64_t iters = (nchunks - a.edge_first + stride - 1) / stride;
   for (uint64_t it = 0; it < iters; it++) {
      const uint64_t c = a.edge_first + it * stride + (uint64_t)block * blockDim.x + threadIdx.x;
      uint32_t h[4] = {0, 0, 0, 0};
      uint32_t any = 0;
      if (c < nchunks) {
         if constexpr (MM_F8_WIDE(SHAPE)) {
            any = mm_f8w_chunk(a, c * 16, h);
         }
         else {
            const uint4 w = mm_load_chunk(a.t.g.rom, a.t.g.nbytes, c * 16);
            const uint32_t back = c ? *reinterpret_cast<const uint32_t *>(a.t.g.rom + c * 16 - 4) : 0u;
            any = mm_f8_chunk<SHAPE>(w, back, a.pat, a.sh, h);
         }
      }
      if (__ballot(any != 0) != 0) {
         mm_f8_survivors(a, c * 16, mm_f8_pack(h));
      }
   }
}

// The four dwordx4 loads of one 4 KiB group: scalar base (the group's first byte: wave uniform, handed to the compiler as
// two opaque SGPRs so that it cannot fold the lane's share into a loop-invariant 64-bit VGPR pointer) + the lane's 32-bit
// byte offset -- global_load_dwordx4 v, v_offset, s[base] offset:k*1024.  With the pointer in VGPRs the 16-bit kernel
// needed 73 registers (72 is where the seventh wave per SIMD fits) and an add of 64 bits per group.
__device__ __forceinline__ void mm_load_group(uint4 (&w)[4], const uint8_t *rom, uint64_t group, uint32_t lane16)
{
   const uint64_t base = reinterpret_cast<uint64_t>(rom) + group * 4096;
   uint32_t lo = (uint32_t)base, hi = (uint32_t)(base >> 32);
   asm volatile("" : "+s"(lo), "+s"(hi));
   // (a GLOBAL pointer: rebuilt from integers it would be a generic one, and round 4's kernels streamed through flat_load --
   // which counts on vmcnt AND lgkmcnt, so that the compiler drained every load in flight, s_waitcnt vmcnt(0) lgkmcnt(0),
   // once per ring turn instead of waiting for the oldest group only (vmcnt(8)): the 1.5 % the streaming kernel lost
   // between rounds 3 and 4, profiles/r05_filter_ab.log)
   typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
   typedef const __attribute__((address_space(1))) uint8_t *global_bytes;
   typedef const __attribute__((address_space(1))) u32x4 *global_u32x4;
   const global_bytes sb = reinterpret_cast<global_bytes>(((uint64_t)hi << 32) | lo);
#pragma unroll
   for (int k = 0; k < 4; k++) {
      const u32x4 v = *reinterpret_cast<global_u32x4>(sb + lane16 + 1024u * k);
      w[k] = make_uint4(v.x, v.y, v.z, v.w);
   }
}

// The hot kernel.  The ROM is cut into 4 KiB groups; a wave owns SPANS of
// consecutive groups and streams through them, 4 x dwordx4 per lane per group
// (each wave instruction = 1 KiB contiguous), with the loads of the next TWO
// groups in flight while a group is processed (8 KiB per wave; depth 1 leaves
// ~4 % of the read bandwidth on the table).  The dword in front of a lane's
// chunk comes from the neighbouring lane (DPP wave_shr:1), lane 0 takes it from
// lane 63 of the previous piece (v_readlane) -- no second memory access.
template <int SHAPE>
__global__ __launch_bounds__(256) void mm_filter_u8_edge(MmFilterArgs a)
{
   mm_edge_u8<SHAPE>(a, blockIdx.x, gridDim.x);
}

template <int SHAPE, class A>
__device__ __forceinline__ void mm_stream_u8(const A &a)
{
   constexpr int DEPTH = 2;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = __builtin_amdgcn_readfirstlane((uint32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
   const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
   const uint64_t gps = a.groups_per_span;
   const uint4 *rom4 = reinterpret_cast<const uint4 *>(a.t.g.rom);
   const uint32_t *rom1 = reinterpret_cast<const uint32_t *>(a.t.g.rom);

   const uint64_t nspans = (a.ngroups + gps - 1) / gps;
   MmSpanCursor cursor = mm_span_cursor(a, wave, nwaves, nspans);
   // The stage-1 constants in VECTOR registers: on gfx950 a two-source integer instruction with both sources in VGPRs
   // issues a wave64 in ~2.4 cycles, the same instruction with an SGPR source in ~4.3 (tools/valu_probe.hip,
   // profiles/r05_valu_probe.log) -- and the streaming loop's VALU is busy 65 % of the time (r05_filter_sq_counters.txt).
   uint32_t vpat[4] = {a.pat[0], a.pat[1], a.pat[2], a.pat[3]};
   asm volatile("" : "+v"(vpat[0]), "+v"(vpat[1]));
   // the wave's survivor queue (bucketed scans only: see mm_queue_flush)
   __shared__ MmSurvivorQueue Q;
   if ((threadIdx.x & 63) == 0) {
      Q.count[threadIdx.x >> 6] = 0;                               // (wave-private: no barrier)
   }
   uint64_t span;
   while (mm_next_span(cursor, &span)) {
      const uint64_t g0 = span * gps;
      const uint64_t g1 = g0 + gps < a.ngroups ? g0 + gps : a.ngroups;
      uint32_t carry = g0 ? rom1[g0 * 1024 - 1] : 0u;          // dword in front of the span (wave uniform)
      uint32_t carry2 = 0;                                     // wide shapes: the dword in front of that one
      if constexpr (MM_F8_WIDE(SHAPE) && MM_F8W_BACK2(SHAPE)) {
         carry2 = g0 ? rom1[g0 * 1024 - 2] : 0u;
      }
      uint4 w[DEPTH + 1][4];
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
         const uint64_t gg = g0 + d < g1 ? g0 + d : g1 - 1;
         mm_load_group(w[d], a.t.g.rom, gg, lane * 16u);
      }
      for (uint64_t g = g0; g < g1; g += DEPTH + 1) {
         uint32_t pending = 0;                                  // bit 4*s + u: piece u of group g+s has stage-1 hits
#pragma unroll
         for (int s = 0; s <= DEPTH; s++) {
            // ring slot s holds group g+s; refill the slot that is DEPTH groups ahead
            constexpr int RING = DEPTH + 1;
            const int slot_new = (s + DEPTH) % RING;
            const uint64_t gn = g + s + DEPTH < g1 ? g + s + DEPTH : g1 - 1;
            mm_load_group(w[slot_new], a.t.g.rom, gn, lane * 16u);
            if (g + s < g1) {
#pragma unroll
               for (int u = 0; u < 4; u++) {
                  // last dword of the previous 16 bytes: lane l-1's w.w; lane 0 keeps c
                  const uint32_t c = u == 0 ? carry : __builtin_amdgcn_readlane(w[s][u - 1].w, 63);
                  const uint32_t back = __builtin_amdgcn_update_dpp(c, w[s][u].w, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                  uint32_t any;
                  if constexpr (MM_F8_WIDE(SHAPE)) {
                     uint32_t back2 = 0;
                     if constexpr (MM_F8W_BACK2(SHAPE)) {
                        const uint32_t c2 = u == 0 ? carry2 : __builtin_amdgcn_readlane(w[s][u - 1].z, 63);
                        back2 = __builtin_amdgcn_update_dpp(c2, w[s][u].z, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                     }
                     any = mm_f8w_chunk_any<SHAPE>(w[s][u], back, back2, vpat);
                  }
                  else {
                     any = mm_f8_chunk_any<SHAPE>(w[s][u], back, vpat, a.sh);
                  }
                  pending |= __ballot(any != 0) != 0 ? 1u << (4 * s + u) : 0u;
               }
               carry = __builtin_amdgcn_readlane(w[s][3].w, 63);
               if constexpr (MM_F8_WIDE(SHAPE) && MM_F8W_BACK2(SHAPE)) {
                  carry2 = __builtin_amdgcn_readlane(w[s][3].z, 63);
               }
            }
         }
         // Rare (~1.6 % of the pieces on random bytes): all conditions on the flagged pieces.  One
         // rolled copy for the whole ring, working from a re-read of the piece (an L2 hit) rather
         // than from the ring registers: indexing those with a run-time piece number would move the
         // ring to scratch memory, and unrolling the survivor code 12 times costs 25 VGPRs.
         while (pending) {
            const uint32_t bit = (uint32_t)__builtin_ctz(pending);
            pending &= pending - 1;
            const uint64_t byte0 = (g + (bit >> 2)) * 4096 + (uint64_t)(bit & 3) * 1024 + lane * 16;
            uint32_t h[4];
            uint32_t some;
            if constexpr (MM_F8_WIDE(SHAPE)) {
               some = mm_f8w_chunk(a, byte0, h);
            }
            else {
               const uint4 wu = *reinterpret_cast<const uint4 *>(a.t.g.rom + byte0);
               const uint32_t back = byte0 ? *reinterpret_cast<const uint32_t *>(a.t.g.rom + byte0 - 4) : 0u;
               some = mm_f8_chunk<SHAPE>(wu, back, a.pat, a.sh, h);
            }
            if (__ballot(some != 0) != 0) {
               mm_f8_survivors(a, byte0, mm_f8_pack(h), a.bcount ? &Q : nullptr);
            }
         }
      }
      // The parked survivors to their buckets once the queue is half full (the ring registers are dead here; else: one LDS
      // read).  Round 4 emptied it behind every span: a returning atomic per span with a survivor, a microsecond or two in
      // which the wave has no load in flight -- with a two-condition keyword (2^-16 of random positions survive: two or three
      // per wave) the waves that meet six of them end the kernel late.
      if (a.bcount) {
         mm_queue_flush(a, Q, a.iA, MM_QCAP / 2);
      }
   }
   if (a.bcount) {
      mm_queue_flush(a, Q, a.iA);
   }
}

// --------------------------------------------------------------------------
// streaming filter, 16-bit elements (both byte alignments from one read)
// --------------------------------------------------------------------------
//
// For every byte position u the element e(u) = 16 bits at bytes u,u+1 (file
// endianness; big endian = v_perm_b32 on the register).  Conditions
//     (e(u)   - e(u-2)) mod 65536 == pat[0]        delta of keyword position iA
//     (e(u-2) - e(u-4)) mod 65536 == pat[1]        delta of position iA-1 (NCOND == 2)
// candidate start o = u - 2*iA.  Even and odd u are two SWAR streams (2 positions per
// 32-bit op, v_pk_sub_u16) over the same registers: the ROM is read once for both of the
// reference's byte alignments (search_engine.cpp:129-147 copies and scans the block twice).
// The odd stream with two conditions looks 12 bytes back: r[0..2] are the three dwords in
// front of the chunk, r[3..6] the chunk.

// SHAPE of a 16-bit filter: bits 0-2 = number of conditions NC (1..2); bit 4 = condition 0
// compares over a wildcard (gap 2: e(u) - e(u-4)); bits 5-6 = kind of condition 1:
//     0  e(u-2) - e(u-4)   keyword position iA-1, gap 1 (the contiguous shape above)
//     1  e(u-2) - e(u-6)   position iA-1, gap 2 (position iA-2 is a wildcard)
//     2  e(u-4) - e(u-6)   position iA-2, gap 1 (position iA-1 is a wildcard; condition 0 has gap 2)
// All of them stay inside the 12 bytes of look-back of the odd stream.
#define MM_F16_NC(shape) ((shape) & 7)
#define MM_F16_G0(shape) (((shape) >> 4) & 1)
#define MM_F16_K1(shape) (((shape) >> 5) & 3)

template <int SHAPE>
__device__ __forceinline__ uint32_t mm_f16_chunk(const uint32_t (&r)[7], bool be, const uint32_t (&pat)[4],
                                                 uint32_t (&he)[4], uint32_t (&ho)[4])
{
   constexpr int NC = MM_F16_NC(SHAPE), K1 = MM_F16_K1(SHAPE);
   constexpr bool G0 = MM_F16_G0(SHAPE) != 0;
   constexpr bool NEED1 = !G0 || (NC >= 2 && K1 != 1), NEED2 = G0 || (NC >= 2 && K1 == 1);
   uint32_t ev[7], od[7], de[7], dd[7], de2[7], dd2[7];
#pragma unroll
   for (int k = 0; k < 7; k++) {
      ev[k] = be ? mm_bswap16x2(r[k]) : r[k];                       // elements at bytes 4k, 4k+2
   }
#pragma unroll
   for (int k = 1; k < 7; k++) {
      const uint32_t o = mm_alignbit(r[k], r[k - 1], 24);           // bytes 4k-1 .. 4k+2
      od[k] = be ? mm_bswap16x2(o) : o;                              // elements at bytes 4k-1, 4k+1
      de[k] = NEED1 ? mm_sub16x2(ev[k], mm_alignbit(ev[k], ev[k - 1], 16)) : 0u;   // even deltas, gap 1
      de2[k] = NEED2 ? mm_sub16x2(ev[k], ev[k - 1]) : 0u;                           // gap 2
   }
#pragma unroll
   for (int k = 2; k < 7; k++) {
      dd[k] = NEED1 ? mm_sub16x2(od[k], mm_alignbit(od[k], od[k - 1], 16)) : 0u;   // odd deltas, gap 1
      dd2[k] = NEED2 ? mm_sub16x2(od[k], od[k - 1]) : 0u;                           // gap 2
   }
   uint32_t any = 0;
#pragma unroll
   for (int j = 0; j < 4; j++) {
      const int k = j + 3;
      uint32_t ze = (G0 ? de2[k] : de[k]) ^ pat[0];
      uint32_t zo = (G0 ? dd2[k] : dd[k]) ^ pat[0];
      if (NC >= 2) {
         if (K1 == 0) {
            ze |= mm_alignbit(de[k], de[k - 1], 16) ^ pat[1];
            zo |= mm_alignbit(dd[k], dd[k - 1], 16) ^ pat[1];
         }
         else if (K1 == 1) {
            ze |= mm_alignbit(de2[k], de2[k - 1], 16) ^ pat[1];
            zo |= mm_alignbit(dd2[k], dd2[k - 1], 16) ^ pat[1];
         }
         else {
            ze |= de[k - 1] ^ pat[1];
            zo |= dd[k - 1] ^ pat[1];
         }
      }
      he[j] = mm_haszero16(ze);
      ho[j] = mm_haszero16(zo);
      any |= he[j] | ho[j];
   }
   return any;
}

// Stage 1 of the 16-bit streaming loop: condition 0 only (2^-16 of random positions pass, so
// some lane of a wave sees a hit in ~1.6 % of its 1 KiB pieces); mm_f16_chunk runs on those.
#define MM_F16_STAGE1(shape) (((shape) & 0x10) | 1)

// non-zero when some position of the chunk (either byte alignment) passes condition 0
template <int SHAPE>
__device__ __forceinline__ uint32_t mm_f16_chunk_any(const uint32_t (&r)[7], bool be, const uint32_t (&pat)[4])
{
   constexpr bool G0 = MM_F16_G0(SHAPE) != 0;
   // gap 1 needs the element one place (2 bytes) back, gap 2 the one a whole dword back
   uint32_t ev[7], od[7];
#pragma unroll
   for (int k = 2; k < 7; k++) {
      ev[k] = be ? mm_bswap16x2(r[k]) : r[k];
      const uint32_t o = mm_alignbit(r[k], r[k - 1], 24);
      od[k] = be ? mm_bswap16x2(o) : o;
   }
   uint32_t acc = 0;
#pragma unroll
   for (int k = 3; k < 7; k++) {
      const uint32_t ze = mm_sub16x2(ev[k], G0 ? ev[k - 1] : mm_alignbit(ev[k], ev[k - 1], 16)) ^ pat[0];
      const uint32_t zo = mm_sub16x2(od[k], G0 ? od[k - 1] : mm_alignbit(od[k], od[k - 1], 16)) ^ pat[0];
      acc |= ((ze - 0x00010001u) & ~ze) | ((zo - 0x00010001u) & ~zo);
   }
   return acc & 0x80008000u;
}

// ---- WIDE 16-bit shapes (round 6): bit 9 set, bits 4-5 = gap of condition 0 minus one (1..4 elements: up to three
// wildcards between the anchor and the literal before it).  Stage 1 is condition 0 on every byte position, exact, as above:
// the element G0 places back is a quarter / half / three quarters / a whole of the dwords in front (r[0..2] reach 12 bytes
// back: enough for the odd stream at gap 4).  Stage 2 takes every condition of the choice with run-time places, each a
// v_pk_sub_u16 of two UNALIGNED 16-byte loads per byte alignment (see the wide 8-bit shapes); sh[k] = s_k | g_k << 8.
#define MM_F16_WIDE(shape) (((shape) >> 9) & 1)
#define MM_F16W_G0(shape) ((((shape) >> 4) & 3) + 1)
#define MM_F16W_SHAPE(g0) (0x200 | 1 | (((g0) - 1) << 4))

template <int SHAPE>
__device__ __forceinline__ uint32_t mm_f16w_chunk_any(const uint32_t (&r)[7], bool be, const uint32_t (&pat)[4])
{
   constexpr int G0 = MM_F16W_G0(SHAPE);
   uint32_t ev[7], od[7];
#pragma unroll
   for (int k = 1; k < 7; k++) {
      ev[k] = be ? mm_bswap16x2(r[k]) : r[k];
      const uint32_t o = mm_alignbit(r[k], r[k - 1], 24);
      od[k] = be ? mm_bswap16x2(o) : o;
   }
   // the elements G0 places in front of those of dword k: 2 G0 bytes back
   auto back = [](const uint32_t (&e)[7], int k) {
      return G0 == 1 ? mm_alignbit(e[k], e[k - 1], 16) : G0 == 2 ? e[k - 1] : G0 == 3 ? mm_alignbit(e[k - 1], e[k - 2], 16) : e[k - 2];
   };
   uint32_t acc = 0;
#pragma unroll
   for (int k = 3; k < 7; k++) {
      const uint32_t ze = mm_sub16x2(ev[k], back(ev, k)) ^ pat[0];
      const uint32_t zo = mm_sub16x2(od[k], back(od, k)) ^ pat[0];
      acc |= ((ze - 0x00010001u) & ~ze) | ((zo - 0x00010001u) & ~zo);
   }
   return acc & 0x80008000u;
}

// exact hit flags of the 16-byte chunk at byte0, both byte alignments, for ALL conditions of a wide choice
template <class A>
__device__ __forceinline__ uint32_t mm_f16w_chunk(const A &a, uint64_t byte0, bool be, uint32_t (&he)[4], uint32_t (&ho)[4])
{
   uint32_t ze[4] = {0, 0, 0, 0}, zo[4] = {0, 0, 0, 0};
   for (uint32_t k = 0; k < a.ncond; k++) {                         // wave uniform
      const int64_t s2 = 2 * (int64_t)(a.sh[k] & 0xFFu), g2 = 2 * (int64_t)(a.sh[k] >> 8);
      const uint4 xe = mm_load16_at(a.t.g.rom, a.t.g.nbytes, (int64_t)byte0 - s2);
      const uint4 ye = mm_load16_at(a.t.g.rom, a.t.g.nbytes, (int64_t)byte0 - s2 - g2);
      const uint4 xo = mm_load16_at(a.t.g.rom, a.t.g.nbytes, (int64_t)byte0 - 1 - s2);     // the odd stream starts a byte earlier
      const uint4 yo = mm_load16_at(a.t.g.rom, a.t.g.nbytes, (int64_t)byte0 - 1 - s2 - g2);
      const uint32_t x[8] = {xe.x, xe.y, xe.z, xe.w, xo.x, xo.y, xo.z, xo.w};
      const uint32_t y[8] = {ye.x, ye.y, ye.z, ye.w, yo.x, yo.y, yo.z, yo.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
         ze[j] |= mm_sub16x2(be ? mm_bswap16x2(x[j]) : x[j], be ? mm_bswap16x2(y[j]) : y[j]) ^ a.pat[k];
         zo[j] |= mm_sub16x2(be ? mm_bswap16x2(x[4 + j]) : x[4 + j], be ? mm_bswap16x2(y[4 + j]) : y[4 + j]) ^ a.pat[k];
      }
   }
   uint32_t any = 0;
#pragma unroll
   for (int j = 0; j < 4; j++) {
      he[j] = mm_haszero16(ze[j]);
      ho[j] = mm_haszero16(zo[j]);
      any |= he[j] | ho[j];
   }
   return any;
}

// 16 flags of a chunk: bit 4*j + 2*odd + half
__device__ __forceinline__ uint32_t mm_f16_pack(const uint32_t (&he)[4], const uint32_t (&ho)[4])
{
   uint32_t bits = 0;
#pragma unroll
   for (int j = 0; j < 4; j++) {
      const uint32_t e = ((he[j] >> 15) & 1u) | ((he[j] >> 30) & 2u);
      const uint32_t o = ((ho[j] >> 15) & 1u) | ((ho[j] >> 30) & 2u);
      bits |= (e | (o << 2)) << (4 * j);
   }
   return bits;
}

template <class A>
__device__ __forceinline__ void mm_f16_survivors(const A &a, uint64_t chunk0, uint32_t bits, MmSurvivorQueue *Q = nullptr)
{
   if (a.bcount) {
      // bucketed store (see mm_f8_survivors).  The odd stream's first position of a chunk starts one byte in front of it,
      // but no candidate of chunk c lies behind one of chunk c + 1: buckets (whole pieces) stay in offset order.
      auto offset_of = [&](int bit) {
         const int j = bit >> 2, odd = (bit >> 1) & 1, half = bit & 1;
         return (int64_t)(chunk0 + 4 * j + 2 * half) - odd - 2 * (int64_t)a.iA;
      };
      if (a.verify || __ballot(bits != 0 && chunk0 < 2 * MMH_MAX_KEYWORD + 2) != 0) {
         uint32_t keep = 0;
         for (uint32_t todo = bits; todo; todo &= todo - 1) {
            const int bit = __ffs((int)todo) - 1;
            const int64_t o = offset_of(bit);
            if (a.verify ? mm_is_candidate(a.t.g, a.t.plan, o) : (o >= 0)) {
               keep |= 1u << bit;
            }
         }
         bits = keep;
      }
      if (Q) {
         uint32_t total;
         uint32_t at = mm_wave_prefix((uint32_t)__popc(bits), &total);
         if (total == 0) {
            return;
         }
         const int w = mm_queue_wave();
         const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)Q->count[w]);
         if (len + total <= MM_QCAP) {
            at += len;
            while (bits) {
               const int bit = __ffs((int)bits) - 1;
               bits &= bits - 1;
               Q->entry[w][at++] = (uint64_t)offset_of(bit);
            }
            Q->count[w] = len + total;
            return;
         }
      }
      uint32_t slot, room;
      uint64_t *list = mm_bucket_reserve(a, mm_uniform64_k(chunk0), (uint32_t)__popc(bits), &slot, &room);
      while (bits) {
         const int bit = __ffs((int)bits) - 1;
         bits &= bits - 1;
         if (slot < room) {
            mm_store_shared(list + slot, (uint64_t)offset_of(bit));
         }
         slot++;
      }
      return;
   }
   if (!a.verify && !a.dom_count && !a.skip_bits && !a.loud_bits && __ballot(bits != 0 && chunk0 < 2 * MMH_MAX_KEYWORD + 2) == 0) {
      uint32_t slot, room;
      uint64_t *list = mm_cand_reserve(a, (uint32_t)__popc(bits), &slot, &room);
      while (bits) {
         const int bit = __ffs((int)bits) - 1;
         bits &= bits - 1;
         const int j = bit >> 2, odd = (bit >> 1) & 1, half = bit & 1;
         if (slot < room) {
            mm_store_shared(list + slot, chunk0 + 4 * j + 2 * half - odd - 2 * (uint64_t)a.iA);
         }
         slot++;
      }
      return;
   }
   while (__ballot(bits != 0) != 0) {
      bool want = false;
      uint64_t off = 0;
      if (bits) {
         const int bit = __ffs((int)bits) - 1;
         bits &= bits - 1;
         const int j = bit >> 2, odd = (bit >> 1) & 1, half = bit & 1;
         const int64_t u = (int64_t)(chunk0 + 4 * j + 2 * half) - odd;
         const int64_t o = u - 2 * (int64_t)a.iA;
         if (a.verify ? mm_is_candidate(a.t.g, a.t.plan, o) : (o >= 0)) {
            want = true;
            off = (uint64_t)o;
         }
      }
      mm_cand_emit(a, want, off);
   }
}

// bounds-checked version for the ragged end of the ROM
template <int SHAPE, class A>
__device__ __forceinline__ void mm_edge_u16(const A &a, uint32_t block, uint32_t nblocks)
{
   const uint64_t nchunks = (a.t.g.nbytes + 15) / 16;
   const uint64_t stride = (uint64_t)nblocks * blockDim.x;
   const bool be = a.t.g.big_endian != 0;
   const uint64_t iters = (nchunks - a.edge_first + stride - 1) / stride;
   const uint32_t *rom1 = reinterpret_cast<const uint32_t *>(a.t.g.rom);
   for (uint64_t it = 0; it < iters; it++) {
      const uint64_t c = a.edge_first + it * stride + (uint64_t)block * blockDim.x + threadIdx.x;
      uint32_t he[4] = {0, 0, 0, 0}, ho[4] = {0, 0, 0, 0};
      uint32_t any = 0;
      if (c < nchunks) {
         if constexpr (MM_F16_WIDE(SHAPE)) {
            any = mm_f16w_chunk(a, c * 16, be, he, ho);
         }
         else {
            const uint4 w = mm_load_chunk(a.t.g.rom, a.t.g.nbytes, c * 16);
            uint32_t r[7] = {0, 0, 0, w.x, w.y, w.z, w.w};
            if (c) {
               r[0] = rom1[c * 4 - 3]; r[1] = rom1[c * 4 - 2]; r[2] = rom1[c * 4 - 1];
            }
            any = mm_f16_chunk<SHAPE>(r, be, a.pat, he, ho);
         }
      }
      if (__ballot(any != 0) != 0) {
         mm_f16_survivors(a, c * 16, mm_f16_pack(he, ho));
      }
   }
}

// span kernel: same streaming structure as mm_filter_u8 (4 KiB groups, two groups of
// look-ahead, look-back through DPP wave_shr:1 / v_readlane instead of a second load)
template <int SHAPE>
__global__ __launch_bounds__(256) void mm_filter_u16_edge(MmFilterArgs a)
{
   mm_edge_u16<SHAPE>(a, blockIdx.x, gridDim.x);
}

template <int SHAPE, class A>
__device__ __forceinline__ void mm_stream_u16(const A &a)
{
   constexpr int DEPTH = 2;
   const uint32_t lane = threadIdx.x & 63;
   const bool be = a.t.g.big_endian != 0;
   const uint64_t wave = __builtin_amdgcn_readfirstlane((uint32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
   const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
   const uint64_t gps = a.groups_per_span;
   const uint4 *rom4 = reinterpret_cast<const uint4 *>(a.t.g.rom);
   const uint32_t *rom1 = reinterpret_cast<const uint32_t *>(a.t.g.rom);

   const uint64_t nspans = (a.ngroups + gps - 1) / gps;
   MmSpanCursor cursor = mm_span_cursor(a, wave, nwaves, nspans);
   // (the stage-1 constant in a VECTOR register: an SGPR source halves a VALU instruction's issue rate on gfx950, see mm_stream_u8)
   uint32_t vpat[4] = {a.pat[0], a.pat[1], a.pat[2], a.pat[3]};
   asm volatile("" : "+v"(vpat[0]));
   __shared__ MmSurvivorQueue Q;                                   // the waves' survivor queues (mm_queue_flush)
   if ((threadIdx.x & 63) == 0) {
      Q.count[threadIdx.x >> 6] = 0;
   }
   uint64_t span;
   while (mm_next_span(cursor, &span)) {
      const uint64_t g0 = span * gps;
      const uint64_t g1 = g0 + gps < a.ngroups ? g0 + gps : a.ngroups;
      // the three dwords in front of the span (wave uniform)
      uint32_t c1 = 0, c2 = 0, c3 = 0;
      if (g0) {
         c1 = rom1[g0 * 1024 - 3]; c2 = rom1[g0 * 1024 - 2]; c3 = rom1[g0 * 1024 - 1];
      }
      uint4 w[DEPTH + 1][4];
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
         const uint64_t gg = g0 + d < g1 ? g0 + d : g1 - 1;
         mm_load_group(w[d], a.t.g.rom, gg, lane * 16u);
      }
      uint32_t flagged = 0;                                     // bit 4*(g - g0) + u: piece u of group g has stage-1 hits
      for (uint64_t g = g0; g < g1; g += DEPTH + 1) {
#pragma unroll
         for (int s = 0; s <= DEPTH; s++) {
            constexpr int RING = DEPTH + 1;
            const int slot_new = (s + DEPTH) % RING;
            const uint64_t gn = g + s + DEPTH < g1 ? g + s + DEPTH : g1 - 1;
            mm_load_group(w[slot_new], a.t.g.rom, gn, lane * 16u);
            if (g + s < g1) {
               const uint32_t first_bit = 4u * (uint32_t)(g + s - g0);
#pragma unroll
               for (int u = 0; u < 4; u++) {
                  // dwords -3..-1 of this chunk: lane l-1's y, z, w; lane 0 keeps the carries
                  const uint32_t k1 = u == 0 ? c1 : __builtin_amdgcn_readlane(w[s][u - 1].y, 63);
                  const uint32_t k2 = u == 0 ? c2 : __builtin_amdgcn_readlane(w[s][u - 1].z, 63);
                  const uint32_t k3 = u == 0 ? c3 : __builtin_amdgcn_readlane(w[s][u - 1].w, 63);
                  uint32_t r[7];
                  r[0] = __builtin_amdgcn_update_dpp(k1, w[s][u].y, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                  r[1] = __builtin_amdgcn_update_dpp(k2, w[s][u].z, 0x138, 0xf, 0xf, false);
                  r[2] = __builtin_amdgcn_update_dpp(k3, w[s][u].w, 0x138, 0xf, 0xf, false);
                  r[3] = w[s][u].x; r[4] = w[s][u].y; r[5] = w[s][u].z; r[6] = w[s][u].w;
                  uint32_t any;
                  if constexpr (MM_F16_WIDE(SHAPE)) {
                     any = mm_f16w_chunk_any<SHAPE>(r, be, vpat);
                  }
                  else {
                     any = mm_f16_chunk_any<SHAPE>(r, be, vpat);
                  }
                  flagged |= __ballot(any != 0) != 0 ? 1u << (first_bit + u) : 0u;
               }
               c1 = __builtin_amdgcn_readlane(w[s][3].y, 63);
               c2 = __builtin_amdgcn_readlane(w[s][3].z, 63);
               c3 = __builtin_amdgcn_readlane(w[s][3].w, 63);
            }
         }
      }
      // Rare (~1.6 % of the pieces on random bytes): all conditions on the flagged pieces, behind the SPAN -- the ring
      // registers are dead by then, so this code costs the streaming loop no registers: 71 VGPRs = 7 waves per SIMD
      // instead of 73 = 6 with the bucketed store's code inside the loop (the next scan's streaming kernel then
      // finds no slot beside this one).  From a re-read of the piece, like mm_stream_u8, whose 69 VGPRs leave the code
      // where it was: the later re-read misses L2 more often, which costs a scan with 64 K candidates 20 us.
      // (A span has at most 8 groups: 32 flag bits.)
      while (flagged) {
         const uint32_t bit = (uint32_t)__builtin_ctz(flagged);
         flagged &= flagged - 1;
         const uint64_t c = (g0 + (bit >> 2)) * 256 + (uint64_t)(bit & 3) * 64 + lane;     // 16-byte chunk number
         uint32_t he[4], ho[4];
         uint32_t some;
         if constexpr (MM_F16_WIDE(SHAPE)) {
            some = mm_f16w_chunk(a, c * 16, be, he, ho);
         }
         else {
            const uint4 wu = rom4[c];
            uint32_t r[7] = {0, 0, 0, wu.x, wu.y, wu.z, wu.w};
            if (c) {
               r[0] = rom1[c * 4 - 3]; r[1] = rom1[c * 4 - 2]; r[2] = rom1[c * 4 - 1];
            }
            some = mm_f16_chunk<SHAPE>(r, be, a.pat, he, ho);
         }
         if (__ballot(some != 0) != 0) {
            mm_f16_survivors(a, c * 16, mm_f16_pack(he, ho), a.bcount ? &Q : nullptr);
         }
      }
      if (a.bcount) {
         mm_queue_flush(a, Q, 2 * a.iA, MM_QCAP / 2);             // (once it is half full: see mm_stream_u8)
      }
   }
   if (a.bcount) {
      mm_queue_flush(a, Q, 2 * a.iA);
   }
}

template <int SHAPE>
__global__ __launch_bounds__(256) void mm_filter_u8(MmFilterArgs a)
{
   mm_stream_u8<SHAPE>(a);
}

template <int SHAPE>
__global__ __launch_bounds__(256) void mm_filter_u16(MmFilterArgs a)
{
   mm_stream_u16<SHAPE>(a);
}

#include "mm_tiles.h"
#include "mm_forward.h"
#include "mm_fused.h"
#include "mm_tail2.h"

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

__global__ __launch_bounds__(64) void mm_gate(unsigned long long ticks)
{
   const unsigned long long t0 = wall_clock64();
   while (wall_clock64() - t0 < ticks) {
      __builtin_amdgcn_s_sleep(64);
   }
}

// --------------------------------------------------------------------------
// launch wrappers (called from mm_capi.hip)
// --------------------------------------------------------------------------

namespace mm {

// Experiment knobs, read once per process.  They exist for the tuning probes under tools/
// (launch geometry sweeps, condition-count sweeps); production runs leave them unset.
const Tuning &tuning()
{
   static const Tuning t = [] {
      auto number = [](const char *name, long fallback) {
         const char *v = getenv(name);
         return v && *v ? std::max(1L, atol(v)) : fallback;
      };
      Tuning k;
      k.filter_max_conditions = (int)number("MMOORE_FILTER_MAXCOND", 4);
      k.filter_blocks = (uint64_t)number("MMOORE_FILTER_BLOCKS", 256 * 6);
      k.filter_groups_per_span = (uint32_t)std::min<long>(number("MMOORE_FILTER_GPS", 7), 8);   // (a span's flagged pieces are 32 bits)
      k.resolve_blocks = (unsigned)number("MMOORE_RESOLVE_BLOCKS", 4096);
      // (mm_arrive_last counts arrivals in MM_ARRIVE_LINES - 1 groups of MM_ARRIVE_FAN: more workgroups would spill into the next lines)
      k.tail_blocks = (unsigned)std::min<long>(number("MMOORE_TAIL_BLOCKS", 2048), (long)MM_ARRIVE_FAN * (MM_ARRIVE_LINES - 1));
      // (scans in flight: a small tail grid beside the next scan's streaming kernel -- 512 workgroups: 0.687-0.689 ms per 4 GiB
      // scan in the steady state against 0.695-0.696 with 2048, profiles/r03_lane_gate_and_span_tickets.log)
      // (round 4, grouped candidates: between 512 and 1536 workgroups the sparse steady state is the same 0.689-0.691 ms, and
      // dense searches in flight like 1024 best -- `water` 0.775 -> 0.757 ms, `th*s` 0.934 -> 0.91; 2048 costs the split
      // pipeline's synchronous caller 0.09 ms: profiles/r04_lane_tail_blocks_sweep.log, r04_candidate_density_lane_tail.log)
      k.lane_tail_blocks = (unsigned)std::min<long>(number("MMOORE_LANE_TAIL_BLOCKS", 1024), (long)MM_ARRIVE_FAN * (MM_ARRIVE_LINES - 1));
      k.max_candidates = (uint32_t)number("MMOORE_MAX_CANDIDATES", 1048576);
      k.list_candidates = (uint32_t)number("MMOORE_LIST_CANDIDATES", 262144);
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
   if ((int)best.ncond < std::min(u8 ? 3 : 2, want) && !(getenv("MMOORE_FILTER_WIDE") && *getenv("MMOORE_FILTER_WIDE") == '0')) {
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
      if (wbest.ncond > best.ncond) {
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

// Calls f(integral_constant<ELEM>, integral_constant<SHAPE>) for the kernel shape a plan's filter
// choice selects: the contiguous 8-bit shapes 1..4 (compile-time shifts), the run-time-shift
// 8-bit shapes (one per number of conditions and gap mask), the five 16-bit shapes.
template <int V> using mm_int = std::integral_constant<int, V>;

template <int NC, class F, int... M2>
static bool with_shape_u8_masks(uint32_t mask2, std::integer_sequence<int, M2...>, F &f)
{
   return ((mask2 == (uint32_t)M2 ? (f(mm_int<1>(), mm_int<NC | (M2 << 4) | 0x100>()), true) : false) || ...);
}

template <class F, int... G>
static bool with_shape_u8_wide(uint32_t shape, std::integer_sequence<int, G...>, F &f)
{
   return ((shape == (uint32_t)MM_F8W_SHAPE(G / 4 + 1, G % 4 + 1) ? (f(mm_int<1>(), mm_int<MM_F8W_SHAPE(G / 4 + 1, G % 4 + 1)>()), true) : false) || ...);
}

template <class F, int... G>
static bool with_shape_u16_wide(uint32_t shape, std::integer_sequence<int, G...>, F &f)
{
   return ((shape == (uint32_t)MM_F16W_SHAPE(G + 1) ? (f(mm_int<2>(), mm_int<MM_F16W_SHAPE(G + 1)>()), true) : false) || ...);
}

template <class F>
static void with_shape(uint32_t elem_bytes, const FilterChoice &fc, F &&f)
{
   const uint32_t shape = fc.shape;
   if (shape & 0x200u) {
      // wide shapes: 16 pairs of gaps (8-bit), 4 gaps (16-bit)
      if (elem_bytes == 1) {
         with_shape_u8_wide(shape, std::make_integer_sequence<int, 16>(), f);
      }
      else {
         with_shape_u16_wide(shape, std::make_integer_sequence<int, 4>(), f);
      }
      return;
   }
   if (elem_bytes == 1) {
      const uint32_t mask2 = MM_F8_MASK2(shape);
      if (!MM_F8_RT(shape)) {
         switch (fc.ncond) {
         case 4: f(mm_int<1>(), mm_int<4>()); break;
         case 3: f(mm_int<1>(), mm_int<3>()); break;
         case 2: f(mm_int<1>(), mm_int<2>()); break;
         default: f(mm_int<1>(), mm_int<1>()); break;
         }
      }
      else {
         switch (fc.ncond) {
         case 4: with_shape_u8_masks<4>(mask2, std::make_integer_sequence<int, 16>(), f); break;
         case 3: with_shape_u8_masks<3>(mask2, std::make_integer_sequence<int, 8>(), f); break;
         case 2: with_shape_u8_masks<2>(mask2, std::make_integer_sequence<int, 4>(), f); break;
         default: with_shape_u8_masks<1>(mask2, std::make_integer_sequence<int, 2>(), f); break;
         }
      }
   }
   else {
      switch (shape) {
      case 1 | 16: f(mm_int<2>(), mm_int<1 | 16>()); break;
      case 2: f(mm_int<2>(), mm_int<2>()); break;
      case 2 | 32: f(mm_int<2>(), mm_int<2 | 32>()); break;
      case 2 | 16 | 64: f(mm_int<2>(), mm_int<2 | 16 | 64>()); break;
      default:
         // shape 1; (a gap-2 anchor with an adjacent second condition cannot occur: position iA-1 would be a wildcard)
         f(mm_int<2>(), mm_int<1>());
         break;
      }
   }
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
   with_shape(pl.elem_bytes, fc, [&](auto elem, auto shape) {
      constexpr int SHAPE = decltype(shape)::value;
      if constexpr (decltype(elem)::value == 1) {
         launch_filter_pair(mm_filter_u8<SHAPE>, mm_filter_u8_edge<SHAPE>, st, a, g, start, stop);
      }
      else {
         launch_filter_pair(mm_filter_u16<SHAPE>, mm_filter_u16_edge<SHAPE>, st, a, g, start, stop);
      }
   });
}

// the forward engine's pre-pass: the streaming filter over the whole ROM, survivors into the tile bitmap (zeroed by the caller)
static void launch_loud(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc, unsigned long long *ctrl,
                        uint32_t *bits, uint32_t tpd, uint32_t tile)
{
   MmFilterArgs a{};
   fill_filter_args(a, g, pl, fc, nullptr, ctrl, 0, filter_groups_per_span());
   a.loud_bits = bits; a.loud_tpd = tpd; a.loud_tile = tile;
   with_shape(pl.elem_bytes, fc, [&](auto elem, auto shape) {
      constexpr int SHAPE = decltype(shape)::value;
      if constexpr (decltype(elem)::value == 1) {
         launch_filter_pair(mm_filter_u8<SHAPE>, mm_filter_u8_edge<SHAPE>, st, a, g, nullptr, nullptr);
      }
      else {
         launch_filter_pair(mm_filter_u16<SHAPE>, mm_filter_u16_edge<SHAPE>, st, a, g, nullptr, nullptr);
      }
   });
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
   static int cached[64] = {};
   int device = 0;
   if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) {
      return 256;
   }
   if (cached[device] == 0) {
      int cus = 0;
      cached[device] = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0 ? cus : 256;
   }
   return (unsigned)cached[device];
}

static unsigned fused_resident_blocks()
{
   static const unsigned blocks = [] {
      int device = 0, cus = 0, per_cu = 0;
      if (hipGetDevice(&device) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mm_scan_fused<1, 4>, 64 * MM_WAVES, 0) != hipSuccess) {
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
   {
      // (tests: MMOORE_FUSED_TIMEOUT_TICKS=1 makes every grid barrier give up -- the route behind it, finish_pipeline's
      // hand-over to the plain kernels, is otherwise only taken when something keeps workgroups out for 200 ms)
      static const long forced = [] { const char *e = getenv("MMOORE_FUSED_TIMEOUT_TICKS"); return e && *e ? atol(e) : 0L; }();
      if (forced > 0) {
         a.timeout_ticks = (uint64_t)forced;
      }
   }
}

// ROMs up to this size take the single-launch kernel (tools/fused_probe.py: 128 KiB .. 2 MiB 11 us
// on the device against 20; 16 MiB with candidates 32 against 27; 4 GiB 782 against 738)
static uint64_t fused_max_bytes()
{
   static const uint64_t v = [] {
      const char *e = getenv("MMOORE_FUSED_MAX_MIB");
      return (uint64_t)(e && *e ? atol(e) : 4) << 20;
   }();
   return v;
}

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
   with_shape(pl.elem_bytes, fc, [&](auto elem, auto shape) {
      launch_timed(mm_scan_fused<decltype(elem)::value, decltype(shape)::value>, dim3(blocks), dim3(64 * MM_WAVES), st, start, stop, a);
   });
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
   with_shape(pl.elem_bytes, fc, [&](auto elem, auto shape) {
      constexpr int SHAPE = decltype(shape)::value;
      if constexpr (decltype(elem)::value == 1) {
         launch_filter_pair(mm_filter_u8<SHAPE>, mm_filter_u8_edge<SHAPE>, st, a, g, start, stop);
      }
      else {
         launch_filter_pair(mm_filter_u16<SHAPE>, mm_filter_u16_edge<SHAPE>, st, a, g, start, stop);
      }
   });
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
   static const uint32_t direct_limit = [] {
      const char *e = getenv("MMOORE_DIRECT_PUBLISH");
      return std::min<uint32_t>((uint32_t)(e && *e ? atol(e) : MM_DIRECT_PUBLISH), MM_MAX_RANK_SORT);
   }();
   a.direct_limit = direct_limit;
   a.has_edge = 0;
   // Several candidates per wave (mm_resolve_sub): two for keywords of up to 16 symbols, four up to 13, eight up to 4
   // (profiles/r04_candidate_density.log; 64 / 48 more registers than one per wave: 6 / 5 waves per SIMD instead of 8,
   // which costs nothing -- with 7 the compiler spills and every row of the log was slower).
   // MMOORE_TAIL_SUB=1: one, as before (2, 4: at most that many); MMOORE_TAIL_QUAD_MAXL: the longest keyword that gets four
   static const int sub = [] { const char *e = getenv("MMOORE_TAIL_SUB"); return e && *e ? atoi(e) : 8; }();
   static const int quad_maxl = [] { const char *e = getenv("MMOORE_TAIL_QUAD_MAXL"); const int v = e && *e ? atoi(e) : 13; return v > 13 ? 13 : v; }();
   // The grid of a scan with the device to itself: as many workgroups as are resident at once at the variant's waves per
   // SIMD (the device's CUs x occupancy; 256 CUs on the MI355X) -- with 2048 for all of them a quarter of the 6-per-SIMD variants' waves started when the
   // first ones ended and a dense search's tail took half as long again (`water`, 90 K candidates in one launch: 96 -> 61 us;
   // `and` 126 -> 73 with 1280 for the 5-per-SIMD variant; profiles/r04_candidate_density_tail_grid.log).
   // MMOORE_TAIL_BLOCKS overrides.
   const int occ = (sub >= 8 && pl.L <= 4) ? 5 : (sub >= 2 && pl.L <= 16) ? 6 : 8;
   static const bool grid_set = [] { const char *e = getenv("MMOORE_TAIL_BLOCKS"); return e && *e; }();
   const dim3 grid(tail_blocks ? tail_blocks : grid_set ? tuning().tail_blocks : device_cus() * (unsigned)occ), block(64 * MM_WAVES);
   static const long group_min = [] { const char *e = getenv("MMOORE_TAIL_GROUP_MIN"); return e && *e ? atol(e) : -1L; }();
   // (grouped from an eighth of the grid's waves on: at the bench's 4223 candidates the tail takes 24 us instead of 29 and a
   // synchronous scan 0.768 ms instead of 0.79 -- fewer walks to the buckets; any threshold between 0 and 4096 measures
   // the same, profiles/r04_tail_blocks_sweep.log.  Below that a wave's candidates would only wait for each other where
   // they fall back to the one-per-wave resolver)
   a.group_min = group_min >= 0 ? (uint32_t)group_min : grid.x * MM_WAVES / 8;
   if (sub >= 8 && pl.L <= 4) {
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
   d.bpd = (d.tpd + MM_FWD_BATCH - 1) / MM_FWD_BATCH;
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
   a.ndom = dg.ndom; a.dom_list = dom_list; a.tpd = dg.tpd; a.bpd = dg.bpd;
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
   if (pl.elem_bytes == 1 && i1 >= 1 && pl.bridge[i1] <= -1 && pl.bridge[i1] >= -4 && i1 + pl.bridge[i1] >= 0 &&
       !(getenv("MMOORE_FORWARD_SLOW") && *getenv("MMOORE_FORWARD_SLOW") == '1')) {
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
   const bool sweeps = choose_filter(pl, &fc) && !(getenv("MMOORE_FORWARD_SWEEP") && *getenv("MMOORE_FORWARD_SWEEP") == '0');
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
       pl.default_skip == (int32_t)pl.L - 1 && !(getenv("MMOORE_FORWARD_QUIET16") && *getenv("MMOORE_FORWARD_QUIET16") == '0')) {
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
      // the whole grid land on a fraction of the memory channels.  (MMOORE_FWD_CHUNK: that experiment.)
      const char *e = getenv("MMOORE_FWD_CHUNK");
      a.chunk = e && atoi(e) >= 4 ? (uint32_t)atoi(e) : 4u;
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

void launch_gate(hipStream_t st, double ms)
{
   hipLaunchKernelGGL(mm_gate, dim3(1), dim3(64), 0, st, (unsigned long long)(ms * 1e5));
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
