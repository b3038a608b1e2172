# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: streaming-kernel time vs launch geometry with a workgroup slot per CU left free
(what a multi-rank communicator makes the filter do, MMOORE_FILTER_BLOCKS_COMM)."""
import sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    mm = load_package()
    eng = mm.Engine(0)
    n = 4 << 30
    eng.alloc(n); eng.synth(42)
    plan = mm.plan_relative(1, "relativesrch")
    for i in range(150):
        eng.scan(plan, block_bytes=524288)
    f, t = eng.timing_history(60)
    print("blocks %s gps %s: filter %.4f ms (min %.4f)  total %.4f ms" % (os.environ.get("MMOORE_FILTER_BLOCKS"), os.environ.get("MMOORE_FILTER_GPS"),
          sum(f) / len(f), min(f), sum(t) / len(t)), flush=True)
else:
    for blocks, gps in ((2048, 8), (1792, 8), (1792, 9), (1536, 8), (1536, 9), (1536, 7), (1280, 8), (1024, 8), (2048, 8)):
        env = dict(os.environ, MMOORE_FILTER_BLOCKS=str(blocks), MMOORE_FILTER_GPS=str(gps))
        subprocess.run([sys.executable, __file__, "child"], env=env)
