# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: run the tiny engine vectors through every engine and report the first mismatches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from __graft_entry__ import load_package
from conftest import load_tiny
mm = load_package()
eng = mm.Engine(0)
search, engine = load_tiny()
bad = 0
for i, c in enumerate(engine):
    eng.upload(c["file"])
    plan = mm.plan_relative(c["elem_bytes"], c["keyword"], c["wildcard"])
    kw = dict(block_bytes=c["block_size"], big_endian=c["big_endian"])
    res = {}
    for e, name in ((0, "auto"), (1, "seq"), (2, "fwd")):
        eng.set_engine(e)
        res[name] = eng.scan(plan, **kw).tolist()
        if e == 0:
            res["path"] = eng.counters()["path"]
    eng.set_engine(0)
    if any(res[n] != c["expect"] for n in ("auto", "seq", "fwd")):
        bad += 1
        if bad <= 12:
            print(i, "L", len(c["keyword"]), "kw", "".join(chr(x) for x in c["keyword"]), "block", c["block_size"], "elem", c["elem_bytes"], "be", c["big_endian"],
                  "n", len(c["file"]), "expect", c["expect"][:8], {k: (v[:8] if isinstance(v, list) else v) for k, v in res.items()}, mm.filter_shape(plan), flush=True)
print("engine cases with a mismatch:", bad, "of", len(engine))
