# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: the fused scan kernel against the plain kernels (MMOORE_FUSED=0): per-scan wall time,
device time, parity of every single scan's list, on the bench ROM and on small ROMs."""
import sys, os, subprocess, time
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    import numpy as np
    mm = load_package()
    eng = mm.Engine(0)
    tag = "fused" if os.environ.get("MMOORE_FUSED", "1") != "0" else "plain"
    for n, kw, block in ((4 << 30, "relativesrch", 524288), (128 << 10, "abcde", 0), (2 << 20, "abcde", 0), (16 << 20, "relativesrch", 524288),
                         (16 << 20, "abcde", 0)):
        eng.alloc(n)
        if block:
            mm.synth.RomSpec(42, n, kw, 1, None, False, block).apply_device(eng)
        else:
            eng.synth(42)                                  # the reference benchmark's shape: random bytes, nothing planted
        plan = mm.plan_relative(1, kw)
        ref = eng.scan(plan, block_bytes=block)
        for _ in range(200):
            eng.scan(plan, block_bytes=block)
        t0 = time.perf_counter(); bad = 0
        K = 300
        for _ in range(K):
            r = eng.scan(plan, block_bytes=block)
            bad += 0 if (r.size == ref.size and (r == ref).all()) else 1
        dt = (time.perf_counter() - t0) / K
        f, t = eng.timing_history(60)
        print("%s %10d B %-13s: wall %8.2f us/scan  device %8.2f us (streaming %8.2f)  matches %d  mismatching scans %d  path %d" % (
            tag, n, kw, dt * 1e6, t.mean() * 1e3, f.mean() * 1e3, len(ref), bad, eng.counters()["path"]), flush=True)
else:
    for fused in ("1", "0"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, MMOORE_FUSED=fused))
