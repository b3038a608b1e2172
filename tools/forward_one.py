# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: ONE forward-engine case, 30 forced scans of 1 GiB (mmh_set_engine 2) -- device time per scan, and the case for
rocprofv3 --pmc averages of mm_forward (tools/pmc_kernels.sh "SQ_INSTS_SALU ..." tools/forward_one.py CASE).
    python tools/forward_one.py flood4096|qz|q*v|qzv|alpha16|pad|plain8|wild8|plain16|long41 [LIB]
LIB: another build of libmmoore_hip.so (tools/ab/...) -- before / after on one box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

mm = load_package()
case = sys.argv[1] if len(sys.argv) > 1 else "flood4096"
if len(sys.argv) > 2:
    mm.LIB_PATH = os.path.abspath(sys.argv[2])
eng = mm.Engine(0)
n = 1 << 30
PIECE = 256 << 20
rng = np.random.default_rng(2026)
elem, kw, wc, be = 1, "relativesrch", 0, False
if case in ("flood4096", "alpha16", "qz", "q*v", "qzv", "pad"):
    rom = rng.integers(0, 16 if case == "alpha16" else 256, PIECE, dtype=np.uint8)
    if case == "flood4096":
        k = np.frombuffer(kw.encode(), np.uint8).astype(np.int64)
        m = (rom.size >> 20) * 4096
        pos = np.sort(rng.choice((rom.size - 64) // 32, size=m, replace=False)) * 32 + rng.integers(0, 16, m)
        sh = rng.integers(-int(k.min()), 256 - int(k.max()), m)
        for j, v in enumerate(k):
            rom[pos + j] = (v + sh).astype(np.uint8)
    elif case == "pad":
        # what a flood looks like in a real ROM: runs of one byte (64 KiB - 1 MiB) over half of it, keyword of equal symbols
        kw, at = "aaaa", 0
        while at < rom.size - (2 << 20):
            run = int(rng.integers(64 << 10, 1 << 20))
            rom[at:at + run] = int(rng.integers(0, 256))
            at += run + int(rng.integers(64 << 10, 1 << 20))
    else:
        kw = {"alpha16": "abc"}.get(case, case)
        wc = ord("*") if "*" in kw else 0
    eng.alloc(n)
    for i in range(4):
        eng.poke(i * PIECE, rom)
else:
    elem, kw, wc, be = {"plain8": (1, "relativesrch", 0, False), "wild8": (1, "re*ative*ear*hxy", ord("*"), False),
                        "plain16": (2, "textsrch", 0, False), "long41": (1, "a quite long keyword of forty-one symbols", 0, False)}[case]
    eng.alloc(n)
    mm.synth.RomSpec(42, n, kw if len(kw) <= 32 else kw[:12], elem, wc or None, be).apply_device(eng)
plan = mm.plan_relative(elem, kw, wc)
eng.set_engine(2)
dev = []
for _ in range(30):
    r = eng.scan(plan, block_bytes=524288, big_endian=be, cap=1 << 23)
    dev.append(eng.timings()["total_ms"])
print("%-10s %d-bit '%s' 1 GiB forced forward engine: %d matches, checksum %x, device %.3f ms median  [%s]" % (
    case, 8 * elem, kw, len(r), int(np.bitwise_xor.reduce(r.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))) if len(r) else 0,
    float(np.median(dev)), os.path.basename(mm.LIB_PATH)))
