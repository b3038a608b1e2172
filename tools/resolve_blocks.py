# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: post-filter time of scans vs the resolver's grid (MMOORE_RESOLVE_BLOCKS)."""
import sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    mm = load_package()
    eng = mm.Engine(0)
    n = 4 << 30
    out = []
    eng.alloc(n)
    for elem, kw, wc, planted in ((1, "relativesrch", None, True), (1, "re*ative*ear*hxy", ord("*"), True), (1, "the", None, False)):
        if planted:
            mm.synth.RomSpec(42, n, kw, elem, wc, False).apply_device(eng)
        else:
            eng.synth(42)
        plan = mm.plan_relative(elem, kw, wc or 0)
        for i in range(100):
            r = eng.scan(plan, block_bytes=524288, cap=1 << 17)
        f, t = eng.timing_history(40)
        out.append("%s: %d cand, post-filter %.1f us" % (kw[:6], eng.counters()["candidates"], (sum(t) - sum(f)) / len(f) * 1e3))
    print("resolve blocks %s: %s" % (os.environ.get("MMOORE_RESOLVE_BLOCKS"), " | ".join(out)), flush=True)
else:
    for blocks in (4096, 2048, 1024, 512, 256, 8192):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, MMOORE_RESOLVE_BLOCKS=str(blocks)))
