#!/bin/bash
# SPDX-License-Identifier: GPL-3.0-or-later
# Dev probe: the tail kernel's grid beside the next scan's streaming kernel (MMOORE_LANE_TAIL_BLOCKS, read once per
# process) against the bench's in-flight figure, at 200 steps and in the driver's shape (20 steps from an empty pipeline).
#   gpurun -- 'bash tools/lane_tail_blocks_sweep.sh'  ->  gpurun_out/lane_tail_blocks_sweep.log
OUT=gpurun_out/lane_tail_blocks_sweep.log
: > $OUT
for ROUND in 1 2 3; do for TB in ${LANE_TAIL_BLOCKS_LIST:-512 768 1024 1536}; do
   for SHAPE in "200 20" "20 5"; do
      set -- $SHAPE
      MMOORE_LANE_TAIL_BLOCKS=$TB python3 bench.py --steps $1 --warmup $2 --no-other-configs --no-cpu-baseline --no-strong --no-pmc --no-read-probe --no-end-to-end 2>/dev/null |
         python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('lane tail blocks $TB, $1 steps: in flight %.4f ms  synchronous %.4f ms  streaming kernel %.4f  behind it %.4f' % (
    d['ms_per_step'], d['synchronous']['ms_per_step'], d['stages_ms']['filter'], d['stages_ms']['resolve_order_publish']))" >> $OUT
   done
done; done
sort -s -k4,5 $OUT
