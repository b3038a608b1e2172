# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (round 5): DENSE searches, two revisions of the device code A/B in one process on one box (the sparse
counterpart: tools/filter_ab.py).  The text-like ROM of tools/candidate_density.py, 4 GiB; per keyword every variant scans
in turn -- one launch over the whole ROM (MMH_ROUTE_NO_SPLIT): the streaming kernel with its rare path busy (flagged
pieces, survivor queue, bucket appends) and the tail kernel over tens to hundreds of thousands of candidates.

    tools/build_variant.sh head HEAD ; python tools/dense_ab.py [--rounds 40] [TAG ...]     -> profiles/r05_dense_ab.log"""
import argparse
import ctypes as C
import glob
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
os.environ.setdefault("MM_DENSITY_PIECES", "16")
from __graft_entry__ import load_package  # noqa: E402
from filter_ab import Variant, BLOCK  # noqa: E402

PIECE = 256 << 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("tags", nargs="*")
    args = ap.parse_args()
    argv, sys.argv = sys.argv, sys.argv[:1]
    mm = load_package()
    npieces = int(os.environ["MM_DENSITY_PIECES"])
    nbytes = npieces * PIECE
    eng = mm.Engine(0)
    hip = C.CDLL("libamdhip64.so")
    rom_ptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(rom_ptr), C.c_size_t(nbytes + 64)) == 0
    ptr = rom_ptr.value
    eng.attach(ptr, nbytes)                                                # (every variant attaches to the same bytes)
    spec_cd = importlib.util.spec_from_file_location("cd", os.path.join(HERE, "candidate_density.py"))
    src = open(spec_cd.origin).read().split("eng = mm.Engine(0)")[0]      # its ROM builders only
    ns = {"__file__": spec_cd.origin}
    exec(compile(src, spec_cd.origin, "exec"), ns)
    sys.argv = argv
    rng = np.random.default_rng(2026)
    for per_mib in (1, 4, 16, 64, 256, 4096):                              # (the generator's state after that probe's random ROMs)
        ns["plant"](ns["random_piece"](rng), rng, "relativesrch", per_mib)
    rom = ns["text_like_piece"](rng)
    for k in range(npieces):
        eng.poke(k * PIECE, rom)
    eng.scan(mm.plan_relative(1, "water", 0), block_bytes=BLOCK, cap=1 << 20)
    paths = {os.path.basename(p)[len("libmmoore_hip_"):-3]: p for p in sorted(glob.glob(os.path.join(ROOT, "tools", "ab", "libmmoore_hip_*.so")))}
    paths["cur"] = mm.LIB_PATH
    tags = args.tags or [t for t in paths if t in ("head", "cur")]
    variants = []
    for tag in tags:
        v = Variant(tag, paths[tag], mm)
        v.out = np.zeros(1 << 20, np.uint64)
        v.attach(ptr, nbytes)
        variants.append(v)
    print("# dense A/B: text-like ROM, %d GiB, one launch per scan, %d rounds, variants: %s" % (nbytes >> 30, args.rounds, " ".join(tags)), flush=True)
    for kw in ("relativesrch", "water", "c*ke", "and", "th*s"):
        wc = ord("*") if "*" in kw else 0
        for v in variants:
            v.set_plan(1, kw, wc)
            for _ in range(5):
                v.scan(False)
        filt = {v.tag: [] for v in variants}
        tot = {v.tag: [] for v in variants}
        counts = {}
        for r in range(args.rounds):
            for v in (variants if r % 2 == 0 else variants[::-1]):
                f, t, n = v.scan(False)
                filt[v.tag].append(f)
                tot[v.tag].append(t)
                counts[v.tag] = n
        assert len(set(counts.values())) == 1, counts
        bf, bt = float(np.median(filt[tags[0]])), float(np.median(tot[tags[0]]))
        for tag in tags:
            f, t = float(np.median(filt[tag])), float(np.median(tot[tag]))
            print("'%s' %-6s streaming kernel %.4f ms (%+.2f %%)  tail kernel %.4f ms (%+.2f %%)  device time of the scan %.4f ms (%+.2f %%) | %d matches" % (
                kw, tag, f, (f / bf - 1) * 100, t - f, ((t - f) / (bt - bf) - 1) * 100 if bt > bf else 0.0, t, (t / bt - 1) * 100, counts[tag]), flush=True)


if __name__ == "__main__":
    main()
