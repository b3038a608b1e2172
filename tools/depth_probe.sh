#!/bin/bash
# SPDX-License-Identifier: GPL-3.0-or-later
# Dev probe: what a late host costs with 1 / 2 / 3 tickets outstanding (bench.py --host-delay-us: the host idles
# that long after every result before it submits the next scan).
for delay in 0 100 200 400; do
for d in 1 2 3; do
python bench.py --no-cpu-baseline --no-other-depth --steps 400 --depth $d --host-delay-us $delay 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('host late by %4d us, %d ticket(s) outstanding: %.4f ms per 4 GiB scan = %.0f GB/s' % ($delay, $d, r['ms_per_step'], r['value']))
    elif 'rror' in l: print(l.rstrip())
"; done; done
