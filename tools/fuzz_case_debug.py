# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: one case of tests/test_gpu_fuzz.py (seed, case) through every engine against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from __graft_entry__ import load_package
from _oracle import Oracle
import test_gpu_fuzz as F
mm = load_package(); orc = Oracle(); eng = mm.Engine(0)
seed, want_case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(7000 + seed)
for case in range(24):
    elem = int(rng.choice([1, 1, 2])); be = bool(elem == 2 and rng.random() < 0.5)
    mode = str(rng.choice(["plain", "plain", "wild", "wild", "case", "seq"]))
    kw, wc, seq = F._keyword(rng, mode)
    try:
        oplan = orc.plan(elem, kw, wc, seq)
    except RuntimeError:
        continue
    plan = mm.plan_relative(elem, kw, wc, seq)
    nbytes = int(rng.choice([3000, 40000, 200000, 1 << 20])) + int(rng.integers(0, 9))
    alphabet = int(rng.choice([2, 3, 5, 16, 200 if elem == 1 else 40000]))
    rom = F._rom(rng, nbytes, elem, be, kw, wc, seq, alphabet)
    block = int(rng.choice([4096, 8191, 65536, 524288]))
    if case != want_case:
        continue
    print("case", case, "kw", kw, "elem", elem, "be", be, "block", block, "nbytes", nbytes, "alphabet", alphabet)
    eng.upload(rom)
    want = orc.engine(oplan, rom, block, be)
    for e, name in ((0, "default"), (1, "sequential"), (2, "forward")):
        eng.set_engine(e)
        for cap in (1 << 12, 1 << 20):
            got = eng.scan(plan, block_bytes=block, big_endian=be, cap=cap)
            ok = got.tolist() == want.tolist()
            print("  engine %-10s cap %7d: %6d results (want %d) path %s %s" % (name, cap, len(got), len(want), eng.counters()["path"], "OK" if ok else "MISMATCH"))
            if not ok:
                g, w = set(got.tolist()), set(want.tolist())
                miss, extra = sorted(w - g), sorted(g - w)
                print("     missing %d (first %s), extra %d (first %s)" % (len(miss), miss[:6], len(extra), extra[:6]))
                if miss:
                    blocks = sorted({m // block for m in miss})
                    print("     blocks with misses:", blocks[:20], "of", -(-nbytes // block))
    eng.set_engine(0)
