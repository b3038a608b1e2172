#!/bin/bash
# Dev probe: bench.py's scans-in-flight figure (20 and 200 timed steps from an empty pipeline) under the lane stream
# arrangements (MMOORE_LANE_MODE) and tail grids (MMOORE_LANE_TAIL_BLOCKS) of mmh_scan_submit.
cd "$(dirname "$0")/.."
for mode in 0 1 2; do
  for tb in 2048 512 256; do
    for steps in 20 200; do
      MMOORE_LANE_MODE=$mode MMOORE_LANE_TAIL_BLOCKS=$tb python3 bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null |
        python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mode $mode tail_blocks %4d steps %3d: in flight %.4f ms/step  (timed-region kernel %.4f)  synchronous %.4f ms/step  kernel alone %.4f  matches %d' % ($tb, $steps, d['ms_per_step'], d['roofline']['timed_region']['kernel_ms'], d['synchronous']['ms_per_step'], d['roofline']['kernel_ms'], d['config']['matches']))"
    done
  done
done
