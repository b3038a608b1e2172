// SPDX-License-Identifier: GPL-3.0-or-later
// mm_kernels.h -- launch wrappers implemented in mm_kernels.hip
#ifndef MM_KERNELS_H
#define MM_KERNELS_H

#include <hip/hip_runtime.h>

#include "mm_internal.h"

namespace mm {

struct Tuning {
   int filter_max_conditions;        // 4 SWAR conditions at most
   uint64_t filter_blocks;           // 1536 workgroups: 6 of the 8 workgroup slots of a CU
   uint32_t filter_groups_per_span;  // 7 groups of 4 KiB
   unsigned resolve_blocks;          // 4096 workgroups
   unsigned tail_blocks;             // 2048 workgroups of mm_scan_tail (mm_scan_tail2: the device's CUs x the variant's waves per SIMD)
   unsigned lane_tail_blocks;        // 1024 workgroups of mm_scan_tail2 behind a scan of the submit lanes
   uint32_t max_candidates;          // 1048576 per scan: the bucketed store, csrc/mm_tail2.h (MMOORE_MAX_CANDIDATES, tests)
   uint32_t list_candidates;         // 262144 per scan: the list-based kernels
};
const Tuning &tuning();

struct FilterChoice {
   uint32_t ncond;     // 0 = no usable SWAR condition (no literal with a literal one or two to its left)
   uint32_t iA;        // keyword position whose delta is condition 0 (the anchor)
   uint32_t pat[4];    // expected delta of condition k, replicated over the SWAR lanes
   uint32_t pos[4];    // keyword position of condition k (pos[0] = iA)
   uint32_t shift[4];  // iA - pos[k]
   uint32_t gap[4];    // 1 = compared with the left neighbour, 2 = over one wildcard
   uint32_t shape;     // kernel instantiation (MM_F8_* / MM_F16_* in mm_kernels.hip)
};

bool choose_filter(const mmh_plan_desc &pl, FilterChoice *fc);
// whether a streaming kernel of that shape exists (mm_filter_shapes.h; every shape choose_filter hands out must)
bool shape_known(uint32_t elem_bytes, uint32_t shape);
// every streaming kernel looked up once (loads the units' code objects ahead of a first scan)
void preload_kernels();
// true: the filter kernel runs the full compare loop on its survivors; false: mm_resolve does
bool filter_verifies(const mmh_plan_desc &pl, const FilterChoice &fc);

// start / stop (optional): HIP events carried by the first / last filter kernel's own dispatch.
// dom_count: count pass (candidates per domain, nothing listed); skip_bits: candidates of the
// flagged domains are dropped (both for candidate floods, engine mode)
void launch_filter(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc,
                   uint64_t *cand, unsigned long long *ctrl, uint64_t cand_cap, hipEvent_t start = nullptr,
                   hipEvent_t stop = nullptr, unsigned int *dom_count = nullptr, const uint32_t *skip_bits = nullptr);

// Device buffers of one scan.  ctrl is zeroed before every scan; layout: MM_CTRL_* in
// mm_internal.h.  cand holds MM_CAND_LISTS candidate lists of cand_cap / MM_CAND_LISTS entries.
// Filter + resolver path: out[i] is the result SLOT of candidate i (reported value or
// ~0 = not a match).  Sequential path: out is an append list of ctrl[1] matches.
struct ResolveBuffers {
   uint64_t *cand;
   uint64_t cand_cap;
   uint64_t *out;
   uint64_t out_cap;
   unsigned long long *ctrl;
   uint64_t *mid_off;        // [mid_cap()] hand-over list mm_resolve -> mm_resolve2
   uint64_t *mid_hi;
   uint64_t *mid_set;                         // (mm_set_t: 64-bit phase sets)
   uint32_t *mid_slot;
   uint64_t *hard_off;       // [hard_cap()] hand-over list mm_resolve2 -> mm_hard_resolve
   uint64_t *hard_hi;
   uint64_t *hard_set;
   uint32_t *hard_slot;
   uint8_t *scratch;
   // bucketed candidate store (mm_internal.h MM_BUCKET_*), zeroed by mm_scan_tail2's last workgroup
   uint64_t *bcand = nullptr;       // [MM_MAX_BUCKETS][MM_BUCKET_CAP]
   unsigned int *bcount = nullptr;  // [MM_MAX_BUCKETS] members of every bucket
};
// geometry of the bucketed store for a ROM: 2^shift-byte buckets, nb of them
struct BucketGeom {
   uint32_t shift, nb;
};
BucketGeom bucket_geom(uint64_t nbytes);
size_t bucket_cand_bytes();
size_t bucket_count_bytes();

// flag_bits != nullptr: the flag pass -- mm_resolve alone, setting the bit of every domain that
// holds a candidate its two windows cannot settle (one bit per domain, zeroed by the caller)
void launch_resolve(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const ResolveBuffers &rb,
                    uint64_t base_offset, uint32_t max_candidates, uint32_t *flag_bits = nullptr);
// The whole first phase of a scan in ONE launch (mm_fused.h): streaming filter, grid barrier, every
// candidate resolved and ranked, results + header published to host_result (pinned) and dev_result,
// then `seq` raised in host_result[MM_HDR_FLAG_WORD].  false: not possible on this device (no launch).
bool launch_fused(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc, const ResolveBuffers &rb,
                  uint64_t base_offset, uint32_t max_candidates, uint64_t *host_result, uint64_t *dev_result, uint32_t max_rank,
                  uint64_t seq, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// does the single-launch kernel suit this ROM (small enough) and this device?
bool fused_applies(const MmGeom &g);
// Big ROMs: the streaming kernel filling the bucketed store of rb, and mm_scan_tail2 (mm_tail2.h) behind it: results,
// header and `seq` as launch_fused; tail_blocks = 0: the default grid (every workgroup resident at once: 256 CUs x the variant's occupancy)
// n ordered slots from the device-side copy of a result block into the pinned block; `seq` raised in *flag (pinned) behind them
void launch_publish_list(hipStream_t st, const uint64_t *src, uint64_t *dst, uint32_t n, unsigned long long *arrive,
                         unsigned long long *flag, unsigned long long seq);
void launch_filter_buckets(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc, const ResolveBuffers &rb,
                           hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
void launch_tail2(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc, const ResolveBuffers &rb,
                  uint64_t base_offset, uint32_t max_candidates, uint64_t *host_result, uint64_t *dev_result, uint64_t seq,
                  hipEvent_t stop = nullptr, unsigned tail_blocks = 0);
// the same tail as a kernel of its own behind launch_filter (mm_scan_tail): results, header and `seq` as launch_fused
void launch_tail(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc, const ResolveBuffers &rb,
                 uint64_t base_offset, uint32_t max_candidates, uint64_t *host_result, uint64_t *dev_result, uint32_t max_rank,
                 uint64_t seq, hipEvent_t stop = nullptr);
// the scan's second phase (only when mm_resolve left candidates over): mm_resolve2 + mm_hard_resolve
void launch_leftovers(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const ResolveBuffers &rb, uint64_t base_offset);
size_t hard_scratch_bytes();
size_t hard_cap();
size_t mid_cap();
size_t ctrl_bytes();
size_t rank_partials_bytes(uint32_t max_n);
// the candidate-free forward engine (mm_forward.h): sizes and buffers
struct DenseGeom {
   uint64_t ndom;
   uint32_t tpd;             // tiles per domain
   uint32_t batch;           // tiles per batch: MM_FWD_BATCH (16), 4 for small inputs (see dense_geom)
   uint32_t bpd;             // batches per domain
   size_t status_bytes;      // ticket + look-back words, at the start of `maps`
   size_t loud_bytes;        // ... followed by the pre-pass's tile bitmap (0 with listed domains)
   size_t maps_bytes;        // ... followed by one published map per batch
};
struct DenseBuffers {
   uint8_t *maps;
   uint64_t *out;            // MM_CAND_LISTS lists of out_cap / MM_CAND_LISTS values
   uint64_t out_cap;
   unsigned long long *ctrl; // list counters at MM_CTRL_LISTS
};
// listed_domains > 0: geometry for a forward pass over that many listed domains only
DenseGeom dense_geom(const MmGeom &g, uint64_t listed_domains = 0);
void launch_dense(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const DenseGeom &dg, const DenseBuffers &db,
                  uint64_t base_offset, const uint32_t *dom_list = nullptr);
// the MM_CAND_LISTS lists of `lists` (list_cap entries each, counters MM_LIST_STRIDE words apart) one behind the other in `packed`
void launch_pack_lists(hipStream_t st, const uint64_t *lists, uint64_t list_cap, const unsigned long long *list_count, uint64_t *packed);
void launch_chain_seq(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, uint64_t *out,
                      unsigned long long *out_count, uint64_t out_cap, uint64_t base_offset);
// orders the ctrl[count_index] keys of `in` into host_result[8..] (pinned host memory) and
// dev_result[8..] (its device-side copy, for the multi-GPU gather), publishes the counters in
// words [0..8) of both and leaves ctrl zeroed -- unless keep_leftovers is set and mm_resolve
// left candidates over (the second phase follows); see mm_rank_scatter
void launch_rank_sort(hipStream_t st, const uint64_t *in, unsigned long long *ctrl, int count_index, uint64_t cap,
                      uint32_t max_n, uint32_t *partials, uint64_t *host_result, uint64_t *dev_result,
                      hipEvent_t stop = nullptr, bool keep_leftovers = false);
// ascending order of n 64-bit keys (mm_sort.hip, rocPRIM radix sort); in and out must not overlap
size_t sort_temp_bytes(uint64_t n);
hipError_t sort_keys(hipStream_t st, const uint64_t *in, uint64_t *out, uint64_t n, void *temp, size_t temp_bytes);
// holds the stream back for `ms` milliseconds (one sleeping wave)
void launch_synth(hipStream_t st, uint8_t *rom, uint64_t nbytes, uint64_t seed, uint64_t base_offset);
void launch_gather(hipStream_t st, const uint8_t *rom, uint64_t nbytes, const uint64_t *offsets, uint64_t n, uint32_t each,
                   uint8_t *out);
void launch_pattern_fill(hipStream_t st, uint8_t *rom, uint64_t first, uint64_t nbytes, int value, int ramp);

} // namespace mm

#endif
