#!/bin/bash
# SPDX-License-Identifier: GPL-3.0-or-later
# Dev probe: bench.py with few steps (pipeline fill and drain inside the timed region).
for k in 5 10 20 50 200 400; do
python bench.py --no-cpu-baseline --steps $k --warmup 3 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('steps %4d: %.4f ms per step (%.0f GB/s); synchronous %.4f' % ($k, r['ms_per_step'], r['value'], r['synchronous']['ms_per_step']))
    elif 'rror' in l: print(l.rstrip())
"; done
