#!/bin/bash
# Dev probe: builds libmmoore_hip.so of another revision (or of this tree with extra compiler flags) next to the shipped
# one, for A/B measurements on ONE box (tools/filter_ab.py, tools/forward_one.py CASE LIB).
#   tools/build_variant.sh TAG GIT-REV               -> tools/ab/libmmoore_hip_TAG.so   built by that revision's own build.py
#   EXTRA="-DMM_FWD_PROFILE" tools/build_variant.sh TAG   -> this tree, mm_kernels.hip compiled with EXTRA, the other objects as built
# (*.so is git-ignored; the file travels to the GPU box with gpurun)
set -e
cd "$(dirname "$0")/.."
TAG=$1; REV=$2
mkdir -p tools/ab
if [ -n "$REV" ]; then
   SRC=/tmp/mm_variant_$TAG
   rm -rf $SRC && mkdir -p $SRC
   git archive $REV monkey-moore_amd include | tar -x -C $SRC
   python3 $SRC/monkey-moore_amd/build.py > /dev/null
   cp $SRC/monkey-moore_amd/lib/libmmoore_hip.so tools/ab/libmmoore_hip_$TAG.so
else
   python3 monkey-moore_amd/build.py > /dev/null
   OBJ=monkey-moore_amd/lib/obj
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imonkey-moore_amd/csrc $EXTRA -x hip -c monkey-moore_amd/csrc/mm_kernels.hip -o /tmp/mm_kernels_$TAG.o
   hipcc --offload-arch=gfx950 -shared -fPIC /tmp/mm_kernels_$TAG.o $(ls $OBJ/*.o | grep -v "/mm_kernels.hip.o") -pthread -L${ROCM_PATH:-/opt/rocm}/lib -lrccl -o tools/ab/libmmoore_hip_$TAG.so
fi
ls -la tools/ab/libmmoore_hip_$TAG.so
