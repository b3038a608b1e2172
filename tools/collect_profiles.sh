#!/bin/bash
# Collects the profiles/ evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r01
# kernel trace + stats of the default bench run, FETCH_SIZE and WRITE_SIZE in separate PMC passes
# (never combined with tracing), and the un-profiled bench line.  Summaries land in gpurun_out/.
set -u
TAG=${1:-r02}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# the driver's own command line first (20 timed steps), then 400 steps
python3 "$REPO/bench.py" --steps 20 --warmup 5 > "$OUT/bench_n1_driver_shape.json" 2> "$OUT/bench_n1_driver_shape.err"
python3 "$REPO/bench.py" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --no-other-configs --no-cpu-baseline --no-strong --no-read-probe > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
# the dominant kernel with the device to itself (what roofline.kernel_ms is): one scan at a time, nothing else in the run
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_depth1" -- python3 "$REPO/bench.py" --depth 1 --no-split --no-other-depth --no-cpu-baseline --no-other-configs --no-strong --no-read-probe > "$OUT/bench_depth1_under_rocprof.json" 2> "$OUT/stats_depth1.err"
# the same for BASELINE C3 / C4 / C4BE: their streaming kernels alone (mm_filter_u16<2> has a file of its own now)
for CFG in C3 C4 C4BE; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_depth1_$CFG" -- python3 "$REPO/bench.py" --config $CFG --steps 100 --depth 1 --no-split --no-other-depth --no-cpu-baseline --no-strong --no-read-probe > "$OUT/bench_depth1_${CFG}_under_rocprof.json" 2> "$OUT/stats_depth1_$CFG.err"
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --depth 1 --no-split --no-cpu-baseline --no-other-depth --no-other-configs --no-strong --no-read-probe --prewarm-s 0.05 > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --depth 1 --no-split --no-cpu-baseline --no-other-depth --no-other-configs --no-strong --no-read-probe --prewarm-s 0.05 > /dev/null 2> "$OUT/pmc_write.err"
# BASELINE C2 / C3 / C4 at full size: stage timings, and the kernel stats that hold mm_filter_u16 (C4)
python3 "$REPO/tools/config_times.py" > "$OUT/config_times.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_configs" -- python3 "$REPO/tools/config_times.py" > /dev/null 2> "$OUT/stats_configs.err"
cd "$REPO"
python3 tools/summarize_profiles.py "$OUT" "$TAG"
