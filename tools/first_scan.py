# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: what ONE synchronous scan of a search costs the FIRST time it is made -- a ROM hacker
scans each keyword once.  Before every timed scan one ROM byte is rewritten in place (mmh_rom_poke: every memo the library
keeps about earlier searches of this ROM is dropped), so that every scan is a first scan; for comparison the same keyword
scanned again and again.  C2's ROM and the text-like ROM of tools/candidate_density.py.  -> profiles/r05_first_scan*.log"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
os.environ.setdefault("MM_DENSITY_PIECES", "16")
from __graft_entry__ import load_package

mm = load_package()
BLOCK = 524288
PIECE = 256 << 20
NPIECES = int(os.environ["MM_DENSITY_PIECES"])


def first_and_repeat(eng, label, keyword, wildcard=0, n=16):
    plan = mm.plan_relative(1, keyword, wildcard)
    byte0 = eng.download(0, 1)
    for _ in range(6):
        eng.poke(0, byte0)
        eng.scan(plan, block_bytes=BLOCK, cap=1 << 20)
    first, filt, tot = [], [], []
    for _ in range(n):
        eng.poke(0, byte0)                                   # the ROM "changed": no memo of an earlier search survives
        t0 = time.perf_counter()
        offs = eng.scan(plan, block_bytes=BLOCK, cap=1 << 20)
        first.append((time.perf_counter() - t0) * 1e3)
        tm = eng.timings()
        filt.append(tm["filter_ms"]); tot.append(tm["total_ms"])
    ctr = eng.counters()
    again = []
    for _ in range(n):
        t0 = time.perf_counter()
        offs2 = eng.scan(plan, block_bytes=BLOCK, cap=1 << 20)
        again.append((time.perf_counter() - t0) * 1e3)
    assert np.array_equal(offs, offs2)
    nb = eng_bytes
    f, a = float(np.median(first)), float(np.median(again))
    print("%-26s '%s'  candidates %8d  matches %8d  path %d | FIRST scan %.3f ms (min %.3f) = %.3f of peak | scanned again %.3f ms (min %.3f) = %.3f of peak" % (
        label, keyword, ctr["candidates"], len(offs), ctr["path"], f, min(first), nb / f / 1e6 / 8000, a, min(again), nb / a / 1e6 / 8000), flush=True)


eng = mm.Engine(0)
eng_bytes = NPIECES * PIECE
print("# first-scan probe: %d GiB, 512 KiB blocks, engine semantics; env: %s" % (
    eng_bytes >> 30, " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("MMOORE_")) or "(defaults)"))
# C2's own ROM
spec = mm.synth.RomSpec(42, eng_bytes, "relativesrch", 1, None, False, BLOCK)
eng.alloc(eng_bytes)
spec.apply_device(eng)
first_and_repeat(eng, "C2 ROM", "relativesrch")
first_and_repeat(eng, "C2 ROM", "re*ative*ear*hxy", ord("*"))
import importlib.util
spec_cd = importlib.util.spec_from_file_location("cd", os.path.join(os.path.dirname(os.path.abspath(__file__)), "candidate_density.py"))
src = open(spec_cd.origin).read().split("eng = mm.Engine(0)")[0]      # its ROM builders only
ns = {"__file__": spec_cd.origin}
exec(compile(src, spec_cd.origin, "exec"), ns)
rng = np.random.default_rng(2026)
for per_mib in (16, 64):
    rom = ns["random_piece"](rng)
    ns["plant"](rom, rng, "relativesrch", per_mib)
    for k in range(NPIECES):
        eng.poke(k * PIECE, rom)
    first_and_repeat(eng, "random + %d plants / MiB" % per_mib, "relativesrch")
# (the very text-like ROM of tools/candidate_density.py: the generator's state after that probe's six random ROMs)
rng = np.random.default_rng(2026)
for per_mib in (1, 4, 16, 64, 256, 4096):
    ns["plant"](ns["random_piece"](rng), rng, "relativesrch", per_mib)
rom = ns["text_like_piece"](rng)
for k in range(NPIECES):
    eng.poke(k * PIECE, rom)
for kw in ("relativesrch", "water", "c*ke", "and", "th*s", "the"):
    first_and_repeat(eng, "text-like ROM", kw, ord("*") if "*" in kw else 0, n=8 if kw == "the" else 16)
