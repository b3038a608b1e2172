# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (round 5): N synchronous scans of a BASELINE configuration, one launch of the streaming kernel each -- what
tools/filter_counters.sh puts under rocprofv3 --pmc.    python tools/sync_scans.py [C2|C3|C4|C4BE] [scans] [GiB]"""
import os
import sys

os.environ.setdefault("MMOORE_DENSE_SPLIT", "0")             # one streaming launch per scan
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

mm = load_package()
CONFIGS = {"C2": (4.0, 1, "relativesrch", 0, False), "C3": (4.0, 1, "re*ative*ear*hxy", ord("*"), False),
           "C4": (8.0, 2, "textsrch", 0, False), "C4BE": (8.0, 2, "textsrch", 0, True)}
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
scans = int(sys.argv[2]) if len(sys.argv) > 2 else 30
gib, elem, kw, wc, be = CONFIGS[name]
if len(sys.argv) > 3:
    gib = float(sys.argv[3])
nbytes = int(gib * (1 << 30))
eng = mm.Engine(0)
eng.alloc(nbytes)
mm.synth.RomSpec(42, nbytes, kw, elem, wc or None, be, 524288).apply_device(eng)
plan = mm.plan_relative(elem, kw, wc)
import numpy as np  # noqa: E402
f = []
for _ in range(scans):
    offs = eng.scan(plan, block_bytes=524288, big_endian=be)
    f.append(eng.timings()["filter_ms"])
print(name, "%.2f GiB" % gib, len(offs), "matches; streaming kernel median %.4f ms over the last %d scans" % (float(np.median(f[len(f) // 2:])), len(f) - len(f) // 2), eng.timings())
