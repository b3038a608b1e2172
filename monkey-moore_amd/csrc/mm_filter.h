// SPDX-License-Identifier: GPL-3.0-or-later
// mm_filter.h -- the streaming filter's device code (8- and 16-bit SWAR stages, survivor queue, span and edge kernels)
// and, included at its end, the resolvers, the forward engine, the single-launch scan and the tail kernels.  Everything in
// here is a template or __forceinline__: mm_kernels.hip (launchers, the small kernels) and the mm_filter_shapes.hip
// units (the streaming kernels of one group of filter shapes each, so that they compile side by side) both include it.
#pragma once

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
__device__ __forceinline__ uint4 mm_load16_unaligned(const uint8_t *p)
{
   struct __attribute__((packed, aligned(1))) Unaligned { uint32_t x, y, z, w; };
   const Unaligned v = *reinterpret_cast<const Unaligned *>(p);        // global_load_dwordx4 at a byte address
   return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint4 mm_load16_at(const uint8_t *rom, uint64_t nbytes, int64_t at)
{
   if (at >= 0 && (uint64_t)at + 16 <= nbytes) {
      return mm_load16_unaligned(rom + at);
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

// per dword: zero bytes where x - y == pat (mod 256), OR-ed into z
__device__ __forceinline__ void mm_f8w_test(uint32_t (&z)[4], const uint4 &x, const uint4 &y, uint32_t pat)
{
   z[0] |= mm_bytesub(x.x, y.x) ^ pat;
   z[1] |= mm_bytesub(x.y, y.y) ^ pat;
   z[2] |= mm_bytesub(x.z, y.z) ^ pat;
   z[3] |= mm_bytesub(x.w, y.w) ^ pat;
}

// Exact hit flags of the 16-byte chunk at byte0 for ALL conditions of a wide choice.  Conditions 0 and 1 first, from three
// loads issued together (their places are the shape's); three flagged pieces in four were flagged by the seven-bit test
// alone and end here.  Then conditions 2 and 3 (run-time places), four loads issued together.  A piece whose loads would
// leave the ROM (its first bytes; the ragged end) takes the checked loads, the whole wave with it.
template <int SHAPE, class A>
__device__ __forceinline__ uint32_t mm_f8w_chunk(const A &a, uint64_t byte0, bool live, uint32_t (&h)[4])
{
   constexpr int G0 = MM_F8W_G0(SHAPE), G1 = MM_F8W_G1(SHAPE);
   const uint8_t *rom = a.t.g.rom;
   const uint64_t nbytes = a.t.g.nbytes;
   const bool inside = byte0 >= MM_F8W_REACH && byte0 + 16 <= nbytes;
   if (__ballot(live) == 0) {
      h[0] = h[1] = h[2] = h[3] = 0;
      return 0;
   }
   const bool fast = __ballot(live && !inside) == 0;                 // wave uniform
   const uint64_t at = live ? byte0 : 0;                                // (idle lanes of the edge kernel: any address inside)
   uint32_t z[4] = {0, 0, 0, 0};
   if (fast) {
      const uint64_t b = at < MM_F8W_REACH ? MM_F8W_REACH : at;
      const uint4 x0 = mm_load16_unaligned(rom + b), x1 = mm_load16_unaligned(rom + b - G0), x2 = mm_load16_unaligned(rom + b - G0 - G1);
      mm_f8w_test(z, x0, x1, a.pat[0]);
      mm_f8w_test(z, x1, x2, a.pat[1]);
   }
   else {
      const uint4 x0 = mm_load16_at(rom, nbytes, (int64_t)at), x1 = mm_load16_at(rom, nbytes, (int64_t)at - G0);
      const uint4 x2 = mm_load16_at(rom, nbytes, (int64_t)at - G0 - G1);
      mm_f8w_test(z, x0, x1, a.pat[0]);
      mm_f8w_test(z, x1, x2, a.pat[1]);
   }
   uint32_t any = live ? mm_haszero8(z[0]) | mm_haszero8(z[1]) | mm_haszero8(z[2]) | mm_haszero8(z[3]) : 0u;
   if (a.ncond > 2 && __ballot(any != 0) != 0) {                        // wave uniform
      // (a choice of three conditions: the fourth is condition 2 once more)
      const uint32_t k3 = a.ncond > 3 ? 3u : 2u;
      const uint32_t s2 = a.sh[2] & 0xFFu, t2 = s2 + (a.sh[2] >> 8), s3 = a.sh[k3] & 0xFFu, t3 = s3 + (a.sh[k3] >> 8);
      if (fast) {
         const uint64_t b = at < MM_F8W_REACH ? MM_F8W_REACH : at;
         const uint4 x = mm_load16_unaligned(rom + b - s2), y = mm_load16_unaligned(rom + b - t2);
         const uint4 u = mm_load16_unaligned(rom + b - s3), v = mm_load16_unaligned(rom + b - t3);
         mm_f8w_test(z, x, y, a.pat[2]);
         mm_f8w_test(z, u, v, a.pat[k3]);
      }
      else {
         const uint4 x = mm_load16_at(rom, nbytes, (int64_t)at - s2), y = mm_load16_at(rom, nbytes, (int64_t)at - t2);
         const uint4 u = mm_load16_at(rom, nbytes, (int64_t)at - s3), v = mm_load16_at(rom, nbytes, (int64_t)at - t3);
         mm_f8w_test(z, x, y, a.pat[2]);
         mm_f8w_test(z, u, v, a.pat[k3]);
      }
   }
#pragma unroll
   for (int j = 0; j < 4; j++) {
      h[j] = live ? mm_haszero8(z[j]) : 0u;
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
      if constexpr (MM_F8_WIDE(SHAPE)) {
         any = mm_f8w_chunk<SHAPE>(a, c * 16, c < nchunks, h);          // (the whole wave: ballots inside)
      }
      else if (c < nchunks) {
         const uint4 w = mm_load_chunk(a.t.g.rom, a.t.g.nbytes, c * 16);
         const uint32_t back = c ? *reinterpret_cast<const uint32_t *>(a.t.g.rom + c * 16 - 4) : 0u;
         any = mm_f8_chunk<SHAPE>(w, back, a.pat, a.sh, h);
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
      uint32_t flagged = 0;                                     // wide shapes: bit 4*(g - g0) + u, the whole span's (<= 8 groups)
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
         if constexpr (MM_F8_WIDE(SHAPE)) {
            flagged |= pending << (4u * (uint32_t)(g - g0));
            pending = 0;
         }
         while (pending) {
            const uint32_t bit = (uint32_t)__builtin_ctz(pending);
            pending &= pending - 1;
            const uint64_t byte0 = (g + (bit >> 2)) * 4096 + (uint64_t)(bit & 3) * 1024 + lane * 16;
            const uint4 wu = *reinterpret_cast<const uint4 *>(a.t.g.rom + byte0);
            const uint32_t back = byte0 ? *reinterpret_cast<const uint32_t *>(a.t.g.rom + byte0 - 4) : 0u;
            uint32_t h[4];
            if (__ballot(mm_f8_chunk<SHAPE>(wu, back, a.pat, a.sh, h) != 0) != 0) {
               mm_f8_survivors(a, byte0, mm_f8_pack(h), a.bcount ? &Q : nullptr);
            }
         }
      }
      // Wide shapes: the flagged pieces (6 % of them on random bytes: stage 1 tests seven bits) behind the SPAN, like the
      // 16-bit kernel -- the ring registers are dead here, and the second stage's loads (up to seven of 16 bytes, issued
      // together) need them.
      if constexpr (MM_F8_WIDE(SHAPE)) {
         while (flagged) {
            const uint32_t bit = (uint32_t)__builtin_ctz(flagged);
            flagged &= flagged - 1;
            const uint64_t byte0 = (g0 + (bit >> 2)) * 4096 + (uint64_t)(bit & 3) * 1024 + lane * 16;
            uint32_t h[4];
            if (__ballot(mm_f8w_chunk<SHAPE>(a, byte0, true, h) != 0) != 0) {
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

// one condition, both byte alignments: zero halves where e(.. - s) - e(.. - s - g) == pat (mod 65536), OR-ed into ze / zo.
// CHECKED: loads that may leave the ROM (zeros there).
template <bool CHECKED>
__device__ __forceinline__ void mm_f16w_test(uint32_t (&ze)[4], uint32_t (&zo)[4], const uint8_t *rom, uint64_t nbytes, uint64_t at,
                                             uint32_t sg, uint32_t pat, bool be)
{
   const int64_t s2 = 2 * (int64_t)(sg & 0xFFu), t2 = s2 + 2 * (int64_t)(sg >> 8);
   uint4 xe, ye, xo, yo;                                                  // (the odd stream starts a byte earlier)
   if constexpr (CHECKED) {
      xe = mm_load16_at(rom, nbytes, (int64_t)at - s2); ye = mm_load16_at(rom, nbytes, (int64_t)at - t2);
      xo = mm_load16_at(rom, nbytes, (int64_t)at - 1 - s2); yo = mm_load16_at(rom, nbytes, (int64_t)at - 1 - t2);
   }
   else {
      xe = mm_load16_unaligned(rom + at - s2); ye = mm_load16_unaligned(rom + at - t2);
      xo = mm_load16_unaligned(rom + at - 1 - s2); yo = mm_load16_unaligned(rom + at - 1 - t2);
   }
   const uint32_t x[8] = {xe.x, xe.y, xe.z, xe.w, xo.x, xo.y, xo.z, xo.w};
   const uint32_t y[8] = {ye.x, ye.y, ye.z, ye.w, yo.x, yo.y, yo.z, yo.w};
#pragma unroll
   for (int j = 0; j < 4; j++) {
      ze[j] |= mm_sub16x2(be ? mm_bswap16x2(x[j]) : x[j], be ? mm_bswap16x2(y[j]) : y[j]) ^ pat;
      zo[j] |= mm_sub16x2(be ? mm_bswap16x2(x[4 + j]) : x[4 + j], be ? mm_bswap16x2(y[4 + j]) : y[4 + j]) ^ pat;
   }
}

// Exact hit flags of the 16-byte chunk at byte0, both byte alignments, for ALL conditions of a wide choice: conditions 0 and
// 1 from eight loads issued together, then -- where a position passed both -- conditions 2 and 3.  A piece whose loads would
// leave the ROM takes the checked loads, the whole wave with it.
template <class A>
__device__ __forceinline__ uint32_t mm_f16w_chunk(const A &a, uint64_t byte0, bool live, bool be, uint32_t (&he)[4], uint32_t (&ho)[4])
{
#pragma unroll
   for (int j = 0; j < 4; j++) {
      he[j] = ho[j] = 0;
   }
   if (__ballot(live) == 0) {
      return 0;
   }
   const uint8_t *rom = a.t.g.rom;
   const uint64_t nbytes = a.t.g.nbytes;
   constexpr uint64_t REACH = 2 * MM_F8W_REACH + 1;                          // bytes in front of the chunk a load may start at
   const bool inside = byte0 >= REACH && byte0 + 16 <= nbytes;
   const bool fast = __ballot(live && !inside) == 0;                      // wave uniform
   const uint64_t at = live ? byte0 : 0;
   const uint64_t b = at < REACH ? REACH : at;                               // (idle lanes on the fast way: any address inside)
   uint32_t ze[4] = {0, 0, 0, 0}, zo[4] = {0, 0, 0, 0};
   const uint32_t k1 = a.ncond > 1 ? 1u : 0u;
   if (fast) {
      mm_f16w_test<false>(ze, zo, rom, nbytes, b, a.sh[0], a.pat[0], be);
      mm_f16w_test<false>(ze, zo, rom, nbytes, b, a.sh[k1], a.pat[k1], be);
   }
   else {
      mm_f16w_test<true>(ze, zo, rom, nbytes, at, a.sh[0], a.pat[0], be);
      mm_f16w_test<true>(ze, zo, rom, nbytes, at, a.sh[k1], a.pat[k1], be);
   }
   uint32_t any = 0;
#pragma unroll
   for (int j = 0; j < 4; j++) {
      any |= mm_haszero16(ze[j]) | mm_haszero16(zo[j]);
   }
   if (a.ncond > 2 && __ballot(live && any != 0) != 0) {                  // wave uniform
      const uint32_t k3 = a.ncond > 3 ? 3u : 2u;
      if (fast) {
         mm_f16w_test<false>(ze, zo, rom, nbytes, b, a.sh[2], a.pat[2], be);
         mm_f16w_test<false>(ze, zo, rom, nbytes, b, a.sh[k3], a.pat[k3], be);
      }
      else {
         mm_f16w_test<true>(ze, zo, rom, nbytes, at, a.sh[2], a.pat[2], be);
         mm_f16w_test<true>(ze, zo, rom, nbytes, at, a.sh[k3], a.pat[k3], be);
      }
   }
   any = 0;
#pragma unroll
   for (int j = 0; j < 4; j++) {
      he[j] = live ? mm_haszero16(ze[j]) : 0u;
      ho[j] = live ? mm_haszero16(zo[j]) : 0u;
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
      if constexpr (MM_F16_WIDE(SHAPE)) {
         any = mm_f16w_chunk(a, c * 16, c < nchunks, be, he, ho);            // (the whole wave: ballots inside)
      }
      else if (c < nchunks) {
         const uint4 w = mm_load_chunk(a.t.g.rom, a.t.g.nbytes, c * 16);
         uint32_t r[7] = {0, 0, 0, w.x, w.y, w.z, w.w};
         if (c) {
            r[0] = rom1[c * 4 - 3]; r[1] = rom1[c * 4 - 2]; r[2] = rom1[c * 4 - 1];
         }
         any = mm_f16_chunk<SHAPE>(r, be, a.pat, he, ho);
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
            some = mm_f16w_chunk(a, c * 16, true, be, he, ho);
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
