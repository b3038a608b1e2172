#!/usr/bin/env python3
# SPDX-License-Identifier: GPL-3.0-or-later
"""Differential fuzz: oracle/liboracle.so (restatement) vs oracle/_ref/libmmref.so
(the unmodified reference).  Runs only where the reference was compiled
(this container).  Usage: python oracle/fuzz_vs_ref.py [trials] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from _oracle import Oracle, Ref  # noqa: E402


def random_keyword(rng, mode):
    L = int(rng.integers(2, 17))
    if mode == "lower":
        kw = [int(rng.integers(97, 123)) for _ in range(L)]
    elif mode == "narrow":
        base = int(rng.integers(97, 120))
        kw = [base + int(rng.integers(0, 3)) for _ in range(L)]
    elif mode == "mixed":
        kw = [int(rng.integers(97, 123)) if rng.random() < 0.7 else int(rng.integers(65, 91)) for _ in range(L)]
    else:
        kw = [int(rng.integers(32, 127)) for _ in range(L)]
    return kw


def random_data(rng, n, elem_bytes, kw_vals, style):
    hi = 256 if elem_bytes == 1 else 65536
    if style == 0:
        d = rng.integers(0, hi, n)
    elif style == 1:
        k = int(rng.integers(2, 8))
        d = rng.integers(0, k, n) + int(rng.integers(0, hi - k))
    elif style == 2:
        d = np.full(n, int(rng.integers(0, hi)))
    elif style == 3:
        d = (np.arange(n) * int(rng.integers(1, 4)) + int(rng.integers(0, hi))) % hi
    else:
        d = rng.integers(0, hi, n)
    d = d.astype(np.int64)
    # plant some matches (shifted keyword values)
    if kw_vals is not None and n > len(kw_vals) + 2:
        for _ in range(int(rng.integers(0, 6))):
            pos = int(rng.integers(0, n - len(kw_vals)))
            shift = int(rng.integers(-40, 40))
            for j, v in enumerate(kw_vals):
                if v is None:
                    continue
                d[pos + j] = (v + shift) % hi
    return d.astype(np.uint8 if elem_bytes == 1 else np.uint16)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    orc, ref = Oracle(), Ref()
    bad = 0
    stats = {"search": 0, "value": 0, "engine": 0, "nonempty": 0, "maps": 0}
    for t in range(trials):
        elem = 1 if rng.random() < 0.6 else 2
        kind = rng.random()
        n = int(rng.integers(0, 400))
        if kind < 0.15:
            L = int(rng.integers(2, 10))
            vals = [int(rng.integers(0, 200)) for _ in range(L)]
            data = random_data(rng, n, elem, vals, int(rng.integers(0, 5)))
            try:
                plan = orc.plan_values(elem, vals)
            except RuntimeError:
                continue
            a = orc.search(plan, data)
            b = ref.value_scan(elem, vals, data)
            stats["value"] += 1
        else:
            mode = ["lower", "narrow", "mixed", "any"][int(rng.integers(0, 4))]
            kw = random_keyword(rng, mode)
            wildcard = 0
            if rng.random() < 0.5:
                wildcard = ord("*")
                for i in range(len(kw)):
                    if rng.random() < 0.25:
                        kw[i] = wildcard
            seq = None
            if rng.random() < 0.2:
                seq = list(rng.permutation(np.arange(97, 123)))
                kw = [c if (c == wildcard or 97 <= c <= 122) else 97 + (c % 26) for c in kw]
            kw_vals = [None if c == wildcard else (seq.index(c) if seq else c) for c in kw]
            data = random_data(rng, n, elem, kw_vals, int(rng.integers(0, 5)))
            try:
                plan = orc.plan(elem, kw, wildcard, seq)
            except RuntimeError as e:
                if "loop forever" in str(e):
                    continue
                try:
                    ref.search(elem, kw, data, wildcard, seq)
                    print("oracle refused but ref accepted", kw, e)
                    bad += 1
                except RuntimeError:
                    pass
                continue
            if rng.random() < 0.6:
                a = orc.search(plan, data)
                b = ref.search(elem, kw, data, wildcard, seq)
                stats["search"] += 1
                for i, pos in enumerate(a[:3]):
                    if i < len(b) and ref.result_map(i) != orc.values_map(plan, data, pos):
                        print("VALUES MAP MISMATCH", kw, wildcard, seq, pos, ref.result_map(i), orc.values_map(plan, data, pos))
                        bad += 1
                    stats["maps"] += 1
            else:
                fb = data.view(np.uint8)
                if fb.size == 0:
                    continue
                block = int(rng.integers(2 if elem == 1 else 4, 208))
                be = bool(rng.integers(0, 2)) if elem == 2 else False
                a = orc.engine(plan, fb, block, be)
                b = ref.engine(elem, fb, kw, wildcard, seq, big_endian=be, threads=int(rng.integers(1, 5)), block_size=block)
                stats["engine"] += 1
        if len(a):
            stats["nonempty"] += 1
        if len(a) != len(b) or (a != b).any():
            bad += 1
            print("MISMATCH trial", t, "elem", elem, "oracle", a[:10], "ref", b[:10])
    print("trials", trials, "mismatches", bad, stats)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
