# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: the forward engine alone on small crafted ROMs, against the sequential engine."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
import numpy as np
mm = load_package()
eng = mm.Engine(0)
rng = np.random.default_rng(1)
bad = 0
for trial in range(400):
    L = int(rng.integers(2, 9))
    kw = [int(c) for c in rng.integers(97, 100 + int(rng.integers(0, 20)), L)]
    wc = 0
    if trial % 3 == 0 and L > 2:
        wc = ord("*"); kw[int(rng.integers(1, L))] = wc
    try:
        plan = mm.plan_relative(1, kw, wc)
    except mm.MMError:
        continue
    n = int(rng.choice([100, 2048 + 5, 2048 * 3 + 17, 40000, 70000]))
    alpha = int(rng.choice([1, 2, 3, 200]))
    rom = (rng.integers(0, alpha, n) + 60).astype(np.uint8)
    vals = [None if (wc and c == wc) else c for c in kw]
    for _ in range(int(rng.integers(0, 30))):
        pos = int(rng.integers(0, n - L)); sh = int(rng.integers(-90, 100))
        for j, v in enumerate(vals):
            if v is not None: rom[pos + j] = v + sh
    eng.upload(rom)
    for block in (0, 524288, 4096):
        eng.set_engine(1); seq = eng.scan(plan, block_bytes=block, cap=1 << 17).tolist()
        eng.set_engine(2); fwd = eng.scan(plan, block_bytes=block, cap=1 << 17).tolist()
        if seq != fwd:
            bad += 1
            if bad <= 10:
                miss = sorted(set(seq) - set(fwd))[:6]; extra = sorted(set(fwd) - set(seq))[:6]
                print("trial", trial, "kw", "".join(chr(c) for c in kw), "n", n, "alpha", alpha, "block", block, "seq", len(seq), "fwd", len(fwd),
                      "missing", miss, [m % 2048 for m in miss], "extra", extra, [m % 2048 for m in extra], flush=True)
eng.set_engine(0)
print("mismatching (trial, block) pairs:", bad)
