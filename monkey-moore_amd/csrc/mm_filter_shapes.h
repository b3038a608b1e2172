// SPDX-License-Identifier: GPL-3.0-or-later
// mm_filter_shapes.h -- the streaming kernels of every filter shape, by table.  A shape (MM_F8_* / MM_F16_* in
// mm_filter.h) is a template argument of three kernels: the span kernel, the edge kernel and the single-launch scan.
// 60 shapes in one translation unit took hipcc four minutes; they are instantiated in MM_SHAPE_UNITS units of
// mm_filter_shapes.hip instead (compiled side by side, build.py), and the launchers of mm_kernels.hip look the kernels
// up here.
#pragma once
#include <cstdint>

struct MmFilterArgs;
struct MmFusedArgs;

namespace mm {

struct ShapeKernels {
   uint32_t elem_bytes, shape;
   void (*span)(MmFilterArgs);
   void (*edge)(MmFilterArgs);
   void (*fused)(MmFusedArgs);
};

constexpr int MM_SHAPE_UNITS = 7;
// unit k's table: its shapes, terminated by an entry with elem_bytes == 0
const ShapeKernels *shape_unit_0();
const ShapeKernels *shape_unit_1();
const ShapeKernels *shape_unit_2();
const ShapeKernels *shape_unit_3();
const ShapeKernels *shape_unit_4();
const ShapeKernels *shape_unit_5();
const ShapeKernels *shape_unit_6();

} // namespace mm
