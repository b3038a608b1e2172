// SPDX-License-Identifier: GPL-3.0-or-later
// mm_internal.h -- declarations shared by the host plan builder, the C ABI and
// the HIP kernels.  Not installed; the public surface is include/mmoore_hip.h.
#ifndef MM_INTERNAL_H
#define MM_INTERNAL_H

#include "mmoore_hip.h"

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
// printf-style setter for the thread-local error string behind mmh_last_error()
void mmh_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
#ifdef __cplusplus
}
#endif

// ---- geometry of a scan: how the ROM splits into independent chain domains --
//
// A "domain" is one run of the reference's sequential chain:
//   whole-buffer mode (block_bytes == 0): the single domain [0, nbytes/S) elements
//   engine mode: one domain per (block b, byte alignment p in [0,S)), block b =
//   bytes [b*B, b*B + min(B + (L-1)*S, N - b*B)), search_engine.cpp:218-253,:129-141.
// Alignment START offsets of block b all lie in [b*B, (b+1)*B), so a byte
// offset belongs to at most one domain.
struct MmGeom {
   const uint8_t *rom;
   uint64_t nbytes;        // N
   uint64_t block_bytes;   // B, 0 = whole buffer
   uint64_t nblocks;       // ceil(N / B) (1 in whole-buffer mode)
   uint32_t S;             // element bytes
   uint32_t L;             // keyword length
   uint32_t big_endian;    // 16-bit elements stored big endian
   uint32_t whole;         // 1 = whole-buffer mode (results are element indices)
};

// The per-candidate machinery holds a phase set in one 64-bit word -- one bit per lane of a wave -- and stores phase maps
// of 64 bytes: keywords of up to this many symbols (D <= 63).  (32 until round 5.)  Longer keywords take the streaming
// filter and the first resolver too since round 6 (two phases per lane, csrc/mm_tiles.h mm_resolve_candidate_long), but
// neither the single-launch kernel nor the second-phase resolvers: what the look-back windows leave open goes to the
// forward engine.
constexpr uint32_t MM_RESOLVER_MAX_KEYWORD = 64;
// ... and the longest keyword of the candidate path as a whole
constexpr uint32_t MM_CANDIDATE_MAX_KEYWORD = 128;
static_assert(MM_CANDIDATE_MAX_KEYWORD == MMH_MAX_KEYWORD, "every keyword a plan holds has a candidate path");

// layout of the block a scan publishes (pinned host memory and its device-side copies):
// MM_RESULT_HEADER_WORDS counters, then the ordered matches.  Header words: [0] candidates (= result
// slots of the filter + resolver path), [1] matches appended by the sequential engine, [2] windows
// mapped, [3] hard candidates / overflow flag, [5] left-overs, [6] matches + 1 (0: not ordered on the
// device), [7] which of words 0 / 1 is the list length.
constexpr uint64_t MM_RESULT_HEADER_WORDS = 8;
constexpr uint32_t MM_MAX_RANK_SORT = 16384;      // longest list the rank kernels / the single-launch kernel order (and a gather record holds)
// slots the published block holds: the tail kernel behind the bucketed filter (mm_tail2.h) ranks a candidate with
// two loads whatever their number, so it orders far longer lists than the count-the-smaller-ones kernels above
constexpr uint32_t MM_MAX_PUBLISH = 1048576;
// ... up to this many candidates every slot is also stored straight into pinned host memory (one PCIe write each: the
// host has the list the moment the flag word changes); beyond, the slots only exist in the device-side copy and the
// host fetches them with one DMA copy (a hundred thousand 8-byte PCIe writes would take longer than the scan)
constexpr uint32_t MM_DIRECT_PUBLISH = 8192;
// the pinned block has a few more words behind the slots: the scan's last kernel raises the scan's
// sequence number in the first of them when everything is published (the host polls it)
constexpr uint64_t MM_HDR_FLAG_WORD = MM_RESULT_HEADER_WORDS + MM_MAX_PUBLISH;
constexpr uint64_t MM_RESULT_BLOCK_WORDS = MM_HDR_FLAG_WORD + 8;
// What the host leaves in every result slot of the pinned block before a scan is launched (memset 0xFE).  A kernel's
// slot stores and its flag word travel to host memory as separate PCIe writes from different compute units; should
// the flag ever be seen first, the host recognises the slot that has not landed (no offset and no hole looks like
// this) and waits for it instead of reading a stale offset.
constexpr uint64_t MM_SLOT_POISON = 0xFEFEFEFEFEFEFEFEull;

// ---- bucketed candidate store (big ROMs: streaming kernel + mm_scan_tail2) ---------------------------------------
// The ROM is cut into nb <= MM_MAX_BUCKETS buckets of 2^shift bytes (>= 4 KiB, so a wave's 1 KiB piece lies in one
// bucket); the streaming kernel appends a candidate to the bucket its anchor chunk lies in (one returning atomic per
// wave and piece on that bucket's counter, one non-returning one on the counter of its super-bucket of MM_SUPER
// buckets).  Buckets are in offset order, so a candidate's place in the ascending list is
//     candidates in the buckets before its own  +  members of its own bucket with a smaller offset
// -- a scan over the 64 super-bucket sums (once per workgroup), one over the 64 counters of its super-bucket and a ballot
// over its bucket's members, instead of comparing it with every other candidate.
// candidates a bucket holds (more: the scan starts over with the list-based kernels).  Generous on purpose -- 128 MiB per
// workspace: a common word in a ROM's script sits every few hundred bytes (profiles/r03_candidate_density.log: 'water',
// 2000 per MiB of text), and ranking a candidate against a crowded bucket (members / 64 loads) is cheap next to a second
// pass over the ROM
constexpr uint32_t MM_BUCKET_CAP = 4096;
constexpr uint32_t MM_MAX_BUCKETS = 4096;         // (every workgroup of mm_scan_tail2 sums all the counters: 16 KiB from L2)
constexpr uint32_t MM_SUPER = 64;                 // buckets per super-bucket
constexpr uint32_t MM_MIN_BUCKET_SHIFT = 12;

// ---- per-scan control block (device memory, zeroed before every scan), in u64 words --
//
// Returning atomics on ONE address serialise at ~20 ns each on MI355X even when they are
// spread out in time (measured: 4 K candidate appends cost the streaming filter +85 us),
// so the filter appends to MM_CAND_LISTS independent lists, each with its counter on a
// 128-byte line of its own; mm_resolve renumbers them compactly.
enum {
   MM_CTRL_TOTAL = 0,        // candidates over all lists (written by mm_resolve; ~0 = a list overflowed)
   MM_CTRL_APPENDED = 1,     // matches appended by mm_chain_seq
   MM_CTRL_MID = 2,          // lo 32: candidates handed from mm_resolve to mm_resolve2
   MM_CTRL_HARD = 3,         // lo 32: hard candidates, hi 32: "prefix too long" flag
   MM_CTRL_TICKET = 4,       // arrival ticket of mm_rank_scatter's blocks (the last one re-zeroes the block)
   MM_CTRL_NOMATCH = 6,      // keys of the ordered list that are "not a match" slots (counted by mm_rank_scatter's blocks)
   MM_CTRL_BOVERFLOW = 7,    // bucketed store: appends that found their bucket full (non-zero: nothing is resolved from the buckets)
   MM_CTRL_DECISION = 25,    // mm_scan_fused: outcome of its grid barrier (1 everybody arrived, 2 timed out: nothing is resolved)
   MM_CTRL_T_START = 26,     //   ... wall clock at the kernel's start / when the last workgroup left the streaming phase
   MM_CTRL_T_BARRIER = 27,
   MM_CTRL_T_STAMPS = 28,    //   ... two more stamps of workgroup 0 (tuning aid)
   MM_CTRL_TILES = 8,        // MM_STAT_STRIPES striped counters of tiles walked
   MM_STAT_STRIPES = 16,
   MM_CTRL_LISTS = 32,       // MM_CAND_LISTS list counters, MM_LIST_STRIDE words apart
   MM_CAND_LISTS = 64,
   MM_LIST_STRIDE = 16,
   MM_CTRL_DONE = MM_CTRL_LISTS + MM_CAND_LISTS * MM_LIST_STRIDE,  // unsigned int tickets of mm_hard_resolve (32 of them: 16 words)
   // mm_scan_fused's arrival counters and release flags, one 128-byte line each.  A returning atomic on
   // one address costs ~20 ns, so 1024 workgroups arriving at ONE counter would take 20 us: they arrive
   // in groups of MM_ARRIVE_FAN at a line of their own, the last of a group at the root.
   MM_ARRIVE_FAN = 32,
   MM_ARRIVE_LINES = 1 + 64,                                       // root + up to 64 groups (2048 workgroups)
   MM_CTRL_ARRIVE_BARRIER = MM_CTRL_DONE + 32,                     // (keeps the lines 128-byte aligned: 1088 = 68 * 16)
   MM_CTRL_ARRIVE_END = MM_CTRL_ARRIVE_BARRIER + MM_ARRIVE_LINES * MM_LIST_STRIDE,
   MM_CTRL_RELEASE = MM_CTRL_ARRIVE_END + MM_ARRIVE_LINES * MM_LIST_STRIDE,   // MM_CAND_LISTS release flags the waiting workgroups poll
   MM_CTRL_WORDS = MM_CTRL_RELEASE + MM_CAND_LISTS * MM_LIST_STRIDE
};

#if defined(__HIPCC__) || defined(__cplusplus)
#if defined(__HIPCC__)
#define MM_HD __host__ __device__ inline
#else
#define MM_HD inline
#endif

// number of valid alignments (chain positions) of domain (b, p); <= 0 means none
MM_HD int64_t mm_domain_nv(const MmGeom &g, uint64_t b, uint32_t p)
{
   if (g.whole) {
      return (int64_t)(g.nbytes / g.S) - (int64_t)g.L + 1;
   }
   uint64_t off = b * g.block_bytes;
   uint64_t full = g.block_bytes + (uint64_t)(g.L - 1) * g.S;
   uint64_t remaining = g.nbytes - off;
   uint64_t size = remaining < full ? remaining : full;
   uint64_t count = size / g.S;                       // search_engine.cpp:137
   if ((uint64_t)p + count * g.S > size) {            // :139-141
      count -= 1;
   }
   return (int64_t)count - (int64_t)g.L + 1;
}

// first byte of domain (b, p)
MM_HD uint64_t mm_domain_start(const MmGeom &g, uint64_t b, uint32_t p)
{
   return g.whole ? 0 : b * g.block_bytes + p;
}
#endif

#endif
