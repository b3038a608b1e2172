# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: what candidates cost.  4 GiB ROMs with 1 / 16 / 256 / 4096 planted matches per
MiB on random bytes, and a text-like ROM (ASCII runs from a small vocabulary, pointer tables, 0x00 / 0xFF padding
between random stretches) searched with short everyday keywords -- streaming kernel, everything behind it, whole scan
one at a time, scans in flight.  -> profiles/rNN_candidate_density.log

The ROM is built from a 256 MiB host piece (numpy) uploaded 16 times: candidates repeat every 256 MiB, which is all
the same to the kernels (every wave streams its own spans)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

mm = load_package()
PIECE = 256 << 20
NPIECES = int(os.environ.get("MM_DENSITY_PIECES", "16"))
BLOCK = 524288


def random_piece(rng):
    return rng.integers(0, 256, PIECE, dtype=np.uint8)


def plant(rom, rng, keyword, per_mib):
    """per_mib shifted copies of the keyword per MiB at random places (no byte wraps)"""
    kw = np.frombuffer(keyword.encode(), np.uint8).astype(np.int64)
    n = (rom.size >> 20) * per_mib
    pos = np.sort(rng.choice((rom.size - 64) // 32, size=n, replace=False)) * 32 + rng.integers(0, 16, n)
    shift = rng.integers(-int(kw.min()), 256 - int(kw.max()), n)
    for j, v in enumerate(kw):
        rom[pos + j] = (v + shift).astype(np.uint8)
    return n


def text_like_piece(rng):
    """what ROMs look like: code / compressed data (random), text, tables, padding"""
    rom = random_piece(rng)
    words = [w.encode() for w in ("the and of to a in is it you that he was for on are with as his they be at one have this from "
                                   "or had by hot but some what there we can out other were all your when up use word how said an "
                                   "each she which do their time if will way about many then them would write like so these her "
                                   "long make thing see him two has look more day could go come did my sound no most number who "
                                   "over know water than call first people may down side been now find").split()]
    at = 0
    while at < rom.size - (8 << 20):
        # ~3 % text (a few MiB of script in a 256 MiB piece), ~8 % pointer tables, ~16 % padding, the rest stays random
        kind = int(rng.choice([0, 4, 6, 7, 9], p=[0.10, 0.20, 0.20, 0.20, 0.30]))
        n = int(rng.integers(64 << 10, 1 << 20))
        if kind < 4:                                          # text: words + blanks, at a base other than ASCII half the time
            k = n // 5
            idx = rng.integers(0, len(words), k)
            buf = b" ".join(words[i] for i in idx)[:n]
            t = np.frombuffer(buf, np.uint8).astype(np.int64)
            base = 0 if rng.random() < 0.5 else int(rng.integers(-30, 100))
            rom[at:at + t.size] = ((t + base) & 0xFF).astype(np.uint8)
            n = t.size
        elif kind < 6:                                        # 16-bit little-endian pointer table, ascending
            k = n // 2
            ptr = (np.cumsum(rng.integers(1, 40, k)) + int(rng.integers(0, 30000))) & 0xFFFF
            rom[at:at + 2 * k] = ptr.astype("<u2").view(np.uint8)
        elif kind < 8:                                        # padding
            rom[at:at + n] = 0x00 if kind == 6 else 0xFF
        at += n + int(rng.integers(256 << 10, 4 << 20))       # random stretch in between
    return rom


def measure(eng, label, keyword, wildcard=0):
    plan = mm.plan_relative(1, keyword, wildcard)
    n = eng_bytes
    for _ in range(8):
        offs = eng.scan(plan, block_bytes=BLOCK, cap=1 << 20)
    f, t = [], []
    t0 = time.perf_counter()
    for _ in range(20):
        offs = eng.scan(plan, block_bytes=BLOCK, cap=1 << 20)
        tm = eng.timings()
        f.append(tm["filter_ms"])
        t.append(tm["total_ms"])
    sync = (time.perf_counter() - t0) / 20 * 1e3
    ctr = eng.counters()

    def in_flight(k):
        tickets, last = [], None
        t1 = time.perf_counter()
        for _ in range(k):
            tickets.append(eng.submit(plan, block_bytes=BLOCK))
            if len(tickets) == 3:
                last = eng.collect(tickets.pop(0), cap=1 << 20)
        while tickets:
            last = eng.collect(tickets.pop(0), cap=1 << 20)
        return (time.perf_counter() - t1) / k * 1e3, last
    in_flight(12)
    per20, last = in_flight(20)
    per200, last = in_flight(200)
    assert np.array_equal(last, offs), label
    fm, tt = float(np.mean(f)), float(np.mean(t))
    print("%-34s '%s'  candidates %8d (%7.1f / MiB)  matches %8d  path %d | filter %.3f ms  behind it %.3f ms  device %.3f ms | "
          "one at a time %.3f ms = %.0f GB/s = %.3f of peak | in flight %.3f ms (20 steps) %.3f ms (200 steps) = %.3f of peak" % (
              label, keyword, ctr["candidates"], ctr["candidates"] / (n / (1 << 20)), len(offs), ctr["path"], fm, tt - fm, tt, sync,
              n / sync / 1e6, n / sync / 1e6 / 8000, per20, per200, n / per200 / 1e6 / 8000), flush=True)


eng = mm.Engine(0)
eng_bytes = NPIECES * PIECE
eng.alloc(eng_bytes)
rng = np.random.default_rng(2026)
print("# candidate density probe: %d GiB ROM, 512 KiB blocks, engine semantics" % (eng_bytes >> 30))
for per_mib in (1, 4, 16, 64, 256, 4096):
    rom = random_piece(rng)
    n = plant(rom, rng, "relativesrch", per_mib)
    for k in range(NPIECES):
        eng.poke(k * PIECE, rom)
    measure(eng, "random + %d plants / MiB" % per_mib, "relativesrch")
rom = text_like_piece(rng)
for k in range(NPIECES):
    eng.poke(k * PIECE, rom)
for kw in ("relativesrch", "water", "people", "the", "c*ke", "th*s", "and", "number"):
    measure(eng, "text-like ROM", kw, ord("*") if "*" in kw else 0)
