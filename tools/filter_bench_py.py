# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: filter_ms of mmh_scan on a 4 GiB ROM under different contents / allocators."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
import numpy as np
mm = load_package()
plan = mm.plan_relative(1, "relativesrch")
n = 4 << 30

def run(eng, label, reps=300, gap=0.0):
    f, t = [], []
    for _ in range(reps):
        r = eng.scan(plan, block_bytes=524288)
        tm = eng.timings()
        f.append(tm["filter_ms"]); t.append(tm["total_ms"])
        if gap:
            time.sleep(gap)
    k = 100
    print("%-40s matches %5d filter(last %d) min %.3f avg %.3f  total avg %.3f" % (label, len(r), k, min(f[-k:]), sum(f[-k:]) / k, sum(t[-k:]) / k))

eng = mm.Engine(0)
eng.alloc(n)
eng.synth(42, 0)
run(eng, "hipMalloc, random, no plants")
spec = mm.synth.RomSpec(42, n, "relativesrch", 1, runs=False)
spec.apply_device(eng)
run(eng, "hipMalloc, plants")
spec = mm.synth.RomSpec(42, n, "relativesrch", 1, runs=True)
spec.apply_device(eng)
run(eng, "hipMalloc, plants + runs")
run(eng, "hipMalloc, plants + runs, 5 ms gaps", reps=60, gap=0.005)
eng.close()
if "--torch" in sys.argv:
    import torch
    buf = torch.empty(n + 32, dtype=torch.uint8, device="cuda:0")
    eng = mm.Engine(0)
    eng.attach(buf.data_ptr(), n)
    spec.apply_device(eng)
    run(eng, "torch buffer, own stream, plants + runs")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    run(eng, "torch buffer, torch stream, plants + runs")
