#!/bin/bash
# Dev probe: bench.py's scans-in-flight figure over 20 / 50 / 200 timed steps from an empty pipeline against the gate in front of
# the second scan of a burst (MMOORE_LANE_GATE, percent of a streaming kernel's duration; 0 = none)
cd "$(dirname "$0")/.."
for gate in ${GATES:-0 30 50 70}; do
  for tb in ${TAILS:-2048 512}; do
    for steps in 20 50 200; do
      MMOORE_LANE_GATE=$gate MMOORE_LANE_TAIL_BLOCKS=$tb python3 bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null |
        python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gate %3d%% tail_blocks %4d steps %3d: in flight %.4f ms/step  synchronous %.4f ms/step  kernel alone %.4f  matches %d' % ($gate, $tb, $steps, d['ms_per_step'], d['synchronous']['ms_per_step'], d['roofline']['kernel_ms'], d['config']['matches']))"
    done
  done
done
