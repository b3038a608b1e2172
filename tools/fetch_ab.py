# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: 30 one-launch scans of the 4 GiB bench ROM (C2: 'relativesrch', 512 KiB blocks, MMH_ROUTE_NO_SPLIT) through ONE
build of the library -- for rocprofv3 --pmc FETCH_SIZE per build on one box (does a revision fetch more than another?):
    tools/pmc_kernels.sh FETCH_SIZE tools/fetch_ab.py [tools/ab/libmmoore_hip_TAG.so]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

mm = load_package()
if len(sys.argv) > 1:
    mm.LIB_PATH = os.path.abspath(sys.argv[1])
N, BLOCK = 4 << 30, 524288
eng = mm.Engine(0)
eng.alloc(N)
mm.synth.RomSpec(42, N, "relativesrch", 1, None, False, BLOCK).apply_device(eng)
plan = mm.plan_relative(1, "relativesrch", 0)
eng.set_route(mm.ROUTE_NO_SPLIT)
f = []
for _ in range(30):
    r = eng.scan(plan, block_bytes=BLOCK)
    f.append(eng.timings()["filter_ms"])
print("%s: %d matches, streaming kernel median %.4f ms" % (os.path.basename(mm.LIB_PATH), len(r), float(np.median(f))))
