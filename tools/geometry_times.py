# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: streaming-kernel time vs launch geometry (MMOORE_FILTER_BLOCKS / MMOORE_FILTER_GPS)."""
import sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    mm = load_package()
    eng = mm.Engine(0)
    n = 4 << 30
    eng.alloc(n); eng.synth(42)
    out = []
    for elem, kw in ((1, "relativesrch"), (2, "textsrch")):
        plan = mm.plan_relative(elem, kw)
        for i in range(120):
            eng.scan(plan, block_bytes=524288)
        f, t = eng.timing_history(40)
        out.append("u%d %.4f ms %.0f GB/s" % (elem * 8, sum(f) / len(f), n / (sum(f) / len(f)) / 1e6))
    print("blocks %s gps %s: %s" % (os.environ.get("MMOORE_FILTER_BLOCKS"), os.environ.get("MMOORE_FILTER_GPS"), " | ".join(out)), flush=True)
else:
    for blocks, gps in ((2048, 8), (2048, 6), (2048, 10), (2048, 12), (2048, 7), (2048, 9), (1024, 8), (4096, 8), (1536, 8), (2048, 5), (3072, 8), (2048, 11)):
        env = dict(os.environ, MMOORE_FILTER_BLOCKS=str(blocks), MMOORE_FILTER_GPS=str(gps))
        subprocess.run([sys.executable, __file__, "child"], env=env)
