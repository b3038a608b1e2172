// SPDX-License-Identifier: GPL-3.0-or-later
// mm_sort.hip -- ordering of LONG match lists on the device.
//
// The reference ends every search with std::sort over its results (search_engine.cpp:193-197).
// Lists of up to 16 K entries are ordered by the rank kernels of mm_kernels.hip inside the scan;
// longer ones (short keywords on big ROMs: tens of thousands to millions of matches) go through
// rocPRIM's radix sort here -- a plain library sort, kept in its own translation unit because
// of what the header costs to compile.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "mm_kernels.h"

namespace mm {

size_t sort_temp_bytes(uint64_t n)
{
   size_t bytes = 0;
   const uint64_t *in = nullptr;
   uint64_t *out = nullptr;
   (void)rocprim::radix_sort_keys(nullptr, bytes, in, out, (size_t)n, 0, 64, (hipStream_t) nullptr);
   return bytes;
}

hipError_t sort_keys(hipStream_t st, const uint64_t *in, uint64_t *out, uint64_t n, void *temp, size_t temp_bytes)
{
   return rocprim::radix_sort_keys(temp, temp_bytes, in, out, (size_t)n, 0, 64, st);
}

} // namespace mm
