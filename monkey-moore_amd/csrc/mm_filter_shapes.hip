// SPDX-License-Identifier: GPL-3.0-or-later
// mm_filter_shapes.hip -- compiled MM_SHAPE_UNITS times with -DMM_FILTER_SHAPE_UNIT=k: unit k instantiates the span
// kernel, the edge kernel and the single-launch scan of its group of filter shapes and hands them out as a table
// (mm_filter_shapes.h).  gfx950 only; the device code is mm_filter.h.
#ifndef MM_FILTER_SHAPE_UNIT
#error "compile with -DMM_FILTER_SHAPE_UNIT=0..6 (monkey-moore_amd/build.py does)"
#endif
#include "mm_filter.h"
#include "mm_filter_shapes.h"

namespace mm {

template <int SHAPE>
static constexpr ShapeKernels u8_kernels()
{
   return ShapeKernels{1u, (uint32_t)SHAPE, mm_filter_u8<SHAPE>, mm_filter_u8_edge<SHAPE>, mm_scan_fused<1, SHAPE>};
}

template <int SHAPE>
static constexpr ShapeKernels u16_kernels()
{
   return ShapeKernels{2u, (uint32_t)SHAPE, mm_filter_u16<SHAPE>, mm_filter_u16_edge<SHAPE>, mm_scan_fused<2, SHAPE>};
}

#define MM_END ShapeKernels{0u, 0u, nullptr, nullptr, nullptr}
// one-dword shapes with run-time shifts: NC conditions, gap-2 mask M2
#define MM_RT(nc, m2) u8_kernels<(nc) | ((m2) << 4) | 0x100>()
#define MM_W8(g0, g1) u8_kernels<MM_F8W_SHAPE(g0, g1)>()

#if MM_FILTER_SHAPE_UNIT == 0
// 8-bit: the contiguous shapes (compile-time shifts; BASELINE's keywords) and the run-time-shift shapes of one and two conditions
static const ShapeKernels table[] = {u8_kernels<1>(), u8_kernels<2>(), u8_kernels<3>(), u8_kernels<4>(),
                                     MM_RT(1, 0), MM_RT(1, 1), MM_RT(2, 0), MM_RT(2, 1), MM_RT(2, 2), MM_RT(2, 3), MM_END};
const ShapeKernels *shape_unit_0() { return table; }
#elif MM_FILTER_SHAPE_UNIT == 1
static const ShapeKernels table[] = {MM_RT(3, 0), MM_RT(3, 1), MM_RT(3, 2), MM_RT(3, 3), MM_RT(3, 4), MM_RT(3, 5), MM_RT(3, 6), MM_RT(3, 7), MM_END};
const ShapeKernels *shape_unit_1() { return table; }
#elif MM_FILTER_SHAPE_UNIT == 2
static const ShapeKernels table[] = {MM_RT(4, 0), MM_RT(4, 1), MM_RT(4, 2), MM_RT(4, 3), MM_RT(4, 4), MM_RT(4, 5), MM_RT(4, 6), MM_RT(4, 7), MM_END};
const ShapeKernels *shape_unit_2() { return table; }
#elif MM_FILTER_SHAPE_UNIT == 3
static const ShapeKernels table[] = {MM_RT(4, 8), MM_RT(4, 9), MM_RT(4, 10), MM_RT(4, 11), MM_RT(4, 12), MM_RT(4, 13), MM_RT(4, 14), MM_RT(4, 15), MM_END};
const ShapeKernels *shape_unit_3() { return table; }
#elif MM_FILTER_SHAPE_UNIT == 4
// 8-bit wide shapes: the anchor's gap 1 or 2
static const ShapeKernels table[] = {MM_W8(1, 1), MM_W8(1, 2), MM_W8(1, 3), MM_W8(1, 4), MM_W8(2, 1), MM_W8(2, 2), MM_W8(2, 3), MM_W8(2, 4), MM_END};
const ShapeKernels *shape_unit_4() { return table; }
#elif MM_FILTER_SHAPE_UNIT == 5
// 8-bit wide shapes: the anchor's gap 3 or 4
static const ShapeKernels table[] = {MM_W8(3, 1), MM_W8(3, 2), MM_W8(3, 3), MM_W8(3, 4), MM_W8(4, 1), MM_W8(4, 2), MM_W8(4, 3), MM_W8(4, 4), MM_END};
const ShapeKernels *shape_unit_5() { return table; }
#elif MM_FILTER_SHAPE_UNIT == 6
// 16-bit: the five one-look-back shapes and the four wide ones
static const ShapeKernels table[] = {u16_kernels<1>(), u16_kernels<1 | 16>(), u16_kernels<2>(), u16_kernels<2 | 32>(), u16_kernels<2 | 16 | 64>(),
                                     u16_kernels<MM_F16W_SHAPE(1)>(), u16_kernels<MM_F16W_SHAPE(2)>(), u16_kernels<MM_F16W_SHAPE(3)>(),
                                     u16_kernels<MM_F16W_SHAPE(4)>(), MM_END};
const ShapeKernels *shape_unit_6() { return table; }
#else
#error "MM_FILTER_SHAPE_UNIT out of range"
#endif

} // namespace mm
