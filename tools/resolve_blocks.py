"""Dev probe: post-filter time of the bench scan vs the resolver's grid (MMOORE_RESOLVE_BLOCKS)."""
import sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    mm = load_package()
    eng = mm.Engine(0)
    n = 4 << 30
    out = []
    for elem, kw, wc in ((1, "relativesrch", None), (1, "re*ative*ear*hxy", ord("*"))):
        eng.alloc(n)
        mm.synth.RomSpec(42, n, kw, elem, wc, False).apply_device(eng)
        plan = mm.plan_relative(elem, kw, wc or 0)
        for i in range(120):
            eng.scan(plan, block_bytes=524288)
        f, t = eng.timing_history(40)
        out.append("post-filter %.4f ms (total %.4f)" % ((sum(t) - sum(f)) / len(f), sum(t) / len(t)))
    print("resolve blocks %s: %s" % (os.environ.get("MMOORE_RESOLVE_BLOCKS"), " | ".join(out)), flush=True)
else:
    for blocks in (4096, 2048, 1792, 1536, 1024, 512, 8192):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, MMOORE_RESOLVE_BLOCKS=str(blocks)))
