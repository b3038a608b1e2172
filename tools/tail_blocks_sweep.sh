#!/bin/bash
# SPDX-License-Identifier: GPL-3.0-or-later
# Dev probe: the tail kernel's grid behind a synchronous scan (MMOORE_TAIL_BLOCKS, read once per process) against the
# bench's one-at-a-time figures.   gpurun -- 'bash tools/tail_blocks_sweep.sh'  ->  gpurun_out/tail_blocks_sweep.log
OUT=gpurun_out/tail_blocks_sweep.log
: > $OUT
for ROUND in 1 2 3; do for GM in ${GROUP_MIN_LIST:-8192 4096 2048 1024 256 0}; do for TB in ${TAIL_BLOCKS_LIST:-2048 1024}; do
   MMOORE_TAIL_GROUP_MIN=$GM MMOORE_TAIL_BLOCKS=$TB python3 bench.py --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-strong --no-pmc --no-read-probe --no-end-to-end 2>/dev/null |
      python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('tail blocks $TB group min $GM: in flight %.4f ms  synchronous %.4f ms  streaming kernel %.4f  behind it %.4f  device %.4f' % (
    d['ms_per_step'], d['synchronous']['ms_per_step'], d['stages_ms']['filter'], d['stages_ms']['resolve_order_publish'], d['stages_ms']['device_total']))" >> $OUT
done; done; done
cat $OUT
