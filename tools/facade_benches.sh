#!/bin/bash
# SPDX-License-Identifier: GPL-3.0-or-later
# The facade-level logs under profiles/ (upload / file ingest included): tools/build_tools.sh and
# `make -C oracle harness` first (build container), then this through gpurun.
OUT=gpurun_out/profiles_r02
mkdir -p $OUT
python tools/fused_probe.py > $OUT/r02_fused_vs_plain_probe.log 2>&1
tools/bench_search_mi355x.bin > $OUT/r02_bench_search_cases.log 2>&1
oracle/_ref/ref_bench_search > $OUT/r02_ref_bench_search.log 2>&1
tools/bench_engine_file.bin 4096 /dev/shm 3 > $OUT/r02_engine_file_gpu.log 2>&1
MMOORE_HIP_MULTI=1 tools/bench_engine_file.bin 4096 /dev/shm 3 > $OUT/r02_engine_file_gpu_multi_path_one_gpu.log 2>&1
rm -f /dev/shm/mm_bench_*.bin
tail -2 $OUT/r02_engine_file_gpu.log; tail -1 $OUT/r02_engine_file_gpu_multi_path_one_gpu.log; head -6 $OUT/r02_fused_vs_plain_probe.log; head -4 $OUT/r02_bench_search_cases.log; head -4 $OUT/r02_ref_bench_search.log
