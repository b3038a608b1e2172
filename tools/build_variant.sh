#!/bin/bash
# Dev probe (round 5): builds libmmoore_hip.so of another revision of the device code next to the shipped one, for A/B
# measurements in ONE process on ONE box (tools/filter_ab.py).
#   tools/build_variant.sh TAG [GIT-REV]       -> tools/ab/libmmoore_hip_TAG.so   (no GIT-REV: the working tree)
# (*.so is git-ignored; the file travels to the GPU box with gpurun)
set -e
cd "$(dirname "$0")/.."
TAG=$1; REV=$2
SRC=/tmp/mm_variant_$TAG
rm -rf $SRC && mkdir -p $SRC
if [ -n "$REV" ]; then git archive $REV monkey-moore_amd/csrc include | tar -x -C $SRC; else cp -r monkey-moore_amd include $SRC/ 2>/dev/null || (mkdir -p $SRC/monkey-moore_amd && cp -r monkey-moore_amd/csrc $SRC/monkey-moore_amd/ && cp -r include $SRC/); fi
OBJ=$SRC/obj; mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$SRC/include -I$SRC/monkey-moore_amd/csrc"
pids=()
for u in $SRC/monkey-moore_amd/csrc/*.hip $SRC/monkey-moore_amd/csrc/*.cpp; do
   hipcc $FLAGS -x hip -c $u -o $OBJ/$(basename $u).o &
   pids+=($!)
   if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -pthread -L${ROCM_PATH:-/opt/rocm}/lib -lrccl -o tools/ab/libmmoore_hip_$TAG.so
ls -la tools/ab/libmmoore_hip_$TAG.so
