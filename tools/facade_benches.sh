#!/bin/bash
# SPDX-License-Identifier: GPL-3.0-or-later
# The facade-level logs under profiles/ (upload / file ingest included): tools/build_tools.sh and
# `make -C oracle harness` first (build container), then this through gpurun.
#   the reference's own benchmark cases (benchmarks/bench_search.cpp, unmodified) through the include/mmoore facade on the GPU,
#   the same cases on the compiled reference (host cores), and SearchEngine<T>::run on a 4 GiB tmpfs file
R=${1:-r06}
OUT=gpurun_out
mkdir -p $OUT
tools/bench_search_mi355x.bin > $OUT/${R}_bench_search_cases.log 2>&1
oracle/_ref/ref_bench_search > $OUT/${R}_ref_bench_search.log 2>&1
tools/bench_engine_file.bin 4096 /dev/shm 3 > $OUT/${R}_engine_file_gpu.log 2>&1
rm -f /dev/shm/mm_bench_*.bin
head -5 $OUT/${R}_bench_search_cases.log; head -5 $OUT/${R}_ref_bench_search.log; tail -2 $OUT/${R}_engine_file_gpu.log
