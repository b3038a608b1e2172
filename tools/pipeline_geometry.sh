#!/bin/bash
# SPDX-License-Identifier: GPL-3.0-or-later
# Dev probe: bench.py (two scans in flight + the synchronous leg) under different streaming-kernel grids
# (workgroups, groups per span) and tail-kernel grids.  WITH_CONFIGS=1 adds tools/config_times.py per grid.
# GRIDS="blocks gps tail;blocks gps tail;..."
IFS=';' read -ra CFGS <<< "${GRIDS:-2048 8 2048;1536 7 2048;1280 7 2048;1280 8 2048;1024 8 2048;1536 7 512}"
for cfg in "${CFGS[@]}"; do
set -- $cfg
echo "== blocks $1 gps $2 tail $3"
MMOORE_FILTER_BLOCKS=$1 MMOORE_FILTER_GPS=$2 MMOORE_TAIL_BLOCKS=$3 timeout 200 python bench.py --no-cpu-baseline --steps 400 --depth ${DEPTH:-3} 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('pipe %.4f filt %.4f dev %.4f | sync %.4f filt %.4f dev %.4f'%(r['ms_per_step'],r['roofline']['kernel_ms'],r['roofline']['scan_device_ms'],r['synchronous']['ms_per_step'],r['synchronous']['kernel_ms'],r['synchronous']['scan_device_ms']))
    elif 'Error' in l or 'error' in l: print(l.rstrip())
"
if [ -n "${WITH_CONFIGS:-}" ]; then
MMOORE_FILTER_BLOCKS=$1 MMOORE_FILTER_GPS=$2 MMOORE_TAIL_BLOCKS=$3 timeout 300 python tools/config_times.py 2>&1 | cut -c1-110
fi
done
