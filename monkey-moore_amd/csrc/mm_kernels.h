// mm_kernels.h -- launch wrappers implemented in mm_kernels.hip
#ifndef MM_KERNELS_H
#define MM_KERNELS_H

#include <hip/hip_runtime.h>

#include "mm_internal.h"

namespace mm {

struct FilterChoice {
   uint32_t ncond;   // 0 = no usable SWAR condition (pattern has no two adjacent literals)
   uint32_t iA;
   uint32_t patA;
   uint32_t patB;
};

bool choose_filter(const mmh_plan_desc &pl, FilterChoice *fc);

void launch_filter(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const FilterChoice &fc,
                   uint64_t *cand, unsigned long long *cand_count, uint64_t cand_cap);
void launch_resolve(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, const uint64_t *cand,
                    const unsigned long long *cand_count, uint64_t cand_cap, uint64_t *out,
                    unsigned long long *out_count, uint64_t out_cap, unsigned long long *tiles_walked,
                    uint64_t base_offset, uint32_t max_candidates);
void launch_chain_seq(hipStream_t st, const MmGeom &g, const mmh_plan_desc &pl, uint64_t *out,
                      unsigned long long *out_count, uint64_t out_cap, uint64_t base_offset);
void launch_rank_sort(hipStream_t st, const uint64_t *in, const unsigned long long *count, uint64_t cap,
                      uint64_t max_n, uint64_t *out);
void launch_synth(hipStream_t st, uint8_t *rom, uint64_t nbytes, uint64_t seed, uint64_t base_offset);
void launch_pattern_fill(hipStream_t st, uint8_t *rom, uint64_t first, uint64_t nbytes, int value, int ramp);

} // namespace mm

#endif
