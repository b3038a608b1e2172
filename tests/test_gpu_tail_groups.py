# SPDX-License-Identifier: GPL-3.0-or-later
"""The tail kernel's grouped candidates (mm_tail2.h, mm_resolve_sub: eight / four / two candidates per wave for keywords
of up to 4 / 13 / 16 symbols, once there are more candidates than an eighth of the grid has waves) against the oracle: ROMs of 12 MiB -- beyond the
single-launch kernel -- with tens of thousands of planted matches (every one a candidate), near-misses, a low-entropy
stretch whose candidates a short window cannot settle (those go back to the one-per-wave resolver), matches in the
first positions of blocks and at the ROM's end.  Every group width is run (the widths are chosen per process:
MMOORE_TAIL_SUB, MMOORE_TAIL_QUAD_MAXL), each in a process of its own.  Bit-exact."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import _diag

NBYTES = 12 << 20
LENGTHS = [2, 3, 4, 5, 8, 9, 13, 14, 16]
LETTERS = "etaoinshrdlucmfw"


def _case(rng, L, elem, be, wild):
    """(keyword, wildcard, ROM bytes)"""
    kw = list(rng.choice(list(LETTERS), L))
    if wild and L >= 4:
        for i in rng.choice(np.arange(1, L - 1), max(1, L // 5), replace=False):
            kw[int(i)] = "*"
    kw = "".join(kw)
    n = NBYTES // elem
    hi = 256 if elem == 1 else 65536
    d = rng.integers(0, hi, n).astype(np.int64)
    # a low-entropy stretch: dense candidates, long undecided phase sets
    a = int(rng.integers(n // 4, n // 2))
    base = int(rng.integers(0, hi - 3))
    d[a:a + 8192] = rng.integers(0, 3, 8192) + base
    vals = [None if c == "*" else ord(c) for c in kw]
    lits = [v for v in vals if v is not None]
    nplants = 30000
    # plants every ~400 bytes: some at the very start of blocks, some touching each other, the last at the ROM's end
    pos = np.sort(rng.choice(n - L, nplants, replace=False))
    pos[:64] = (np.arange(64) * (65536 // elem)) + rng.integers(0, 3, 64)
    pos[-1] = n - L
    sh = rng.integers(-min(lits), hi - max(lits), nplants)
    miss = rng.random(nplants) < 0.2                       # near-misses: the last literal off by one
    for j, v in enumerate(vals):
        if v is None:
            continue
        col = v + sh
        if j == max(i for i, w in enumerate(vals) if w is not None):
            col = np.where(miss, np.clip(col ^ 1, 0, hi - 1), col)
        d[pos + j] = col
    arr = d.astype(np.uint8 if elem == 1 else (">u2" if be else "<u2")).view(np.uint8)
    ragged = rng.integers(0, 256, int(rng.integers(0, 4))).astype(np.uint8)
    return kw, ord("*"), np.concatenate([arr, ragged])


@pytest.mark.parametrize("L", LENGTHS)
@pytest.mark.parametrize("elem,be", [(1, False), (2, False), (2, True)])
def test_grouped_candidates_against_oracle(mm, gpu_engine, oracle, L, elem, be):
    rng = np.random.default_rng(4400 + 10 * L + elem + be)
    for wild in (False, True):
        for _ in range(20):
            kw, wc, rom = _case(rng, L, elem, be, wild)
            try:
                oplan = oracle.plan(elem, kw, wc, None)      # (what the reference rejects -- "aa" -- is drawn again)
                break
            except RuntimeError:
                continue
        else:
            pytest.skip("no acceptable keyword")
        plan = mm.plan_relative(elem, kw, wc, None)
        gpu_engine.upload(rom)
        for block in (524288, 65536):
            want = oracle.engine(oplan, rom, block, be)
            got = gpu_engine.scan(plan, block_bytes=block, big_endian=be, cap=1 << 20)
            c = gpu_engine.counters()
            _diag.same(mm, gpu_engine, rom, plan, got, want, (L, elem, be, wild, kw, block, c), block_bytes=block, big_endian=be)
            if c["path"] == 0:
                assert c["candidates"] > 8192, c           # more candidates than the tail kernel has waves: grouped
            tickets = [gpu_engine.submit(plan, block_bytes=block, big_endian=be) for _ in range(2)]
            for got in [gpu_engine.collect(t, cap=1 << 20) for t in tickets]:
                _diag.same(mm, gpu_engine, rom, plan, got, want, (L, elem, be, wild, kw, block, "lanes"), block_bytes=block, big_endian=be)
        # one chain over the whole buffer
        if not be:
            whole = rom[: (rom.size // elem) * elem]
            data = whole if elem == 1 else whole.view("<u2")
            assert gpu_engine.scan(plan, cap=1 << 20).tolist() == oracle.search(oplan, data).tolist(), (L, elem, kw, "whole")


@pytest.mark.parametrize("knobs", [{"MMOORE_TAIL_SUB": "1"}, {"MMOORE_TAIL_SUB": "2"}, {"MMOORE_TAIL_SUB": "4"}, {"MMOORE_TAIL_QUAD_MAXL": "6"}],
                         ids=["one", "pairs", "quads", "quads-to-6"])
def test_every_group_width(knobs):
    """the cases above with the other widths (a process each: the knobs are read once)"""
    if os.environ.get("MM_TAIL_GROUPS_CHILD"):
        pytest.skip("this is the child")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MM_TAIL_GROUPS_CHILD="1", **knobs)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "grouped_candidates"], env=env, capture_output=True, text=True, timeout=1500, cwd=os.path.dirname(here))
    print(r.stdout[-1500:])
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
