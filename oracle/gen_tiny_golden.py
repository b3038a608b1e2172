#!/usr/bin/env python3
# SPDX-License-Identifier: GPL-3.0-or-later
""">= 10 000 tiny differential vectors (inputs <= 1 KiB) from the compiled reference
(oracle/_ref, the unmodified sources of /root/reference), across modes (plain, wildcard, mixed
case, custom sequence, value scan), widths, endianness and block sizes -- weighted towards
alphabets of 2-7 symbols, which force the reference's unsafe skips and overlapping hits.

    python oracle/gen_tiny_golden.py        # -> tests/golden/diff_tiny.json.gz

Build container only (needs oracle/_ref/libmmref.so).  The file is data: raw inputs (base64 of
the little-endian element bytes / the file bytes) + the offsets the reference reported.
"""
import base64
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, HERE)
from _oracle import Ref  # noqa: E402
from gen_golden import safe_kw  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden", "diff_tiny.json.gz")
N_SEARCH, N_ENGINE = 6500, 4500


def b64(a):
    return base64.b64encode(np.ascontiguousarray(a).tobytes()).decode("ascii")


def keyword(rng, L, style):
    """Code points of a keyword of one of the reference's modes."""
    wildcard, seq = 0, None
    if style == "plain":
        kw = [int(rng.integers(97, 123)) for _ in range(L)]
    elif style == "narrow":                          # few distinct symbols: self-overlapping keywords
        pool = [int(x) for x in rng.integers(97, 123, int(rng.integers(1, 4)))]
        kw = [pool[int(rng.integers(0, len(pool)))] for _ in range(L)]
    elif style == "wild":
        wildcard = int(rng.choice([ord("*"), ord("?"), ord("$")]))
        kw = [int(rng.integers(97, 123)) for _ in range(L)]
        for i in range(L):
            if rng.random() < 0.3:
                kw[i] = wildcard
    elif style == "mixed":
        wildcard = ord("*")
        kw = [int(rng.integers(65, 91)) if rng.random() < 0.4 else int(rng.integers(97, 123)) for _ in range(L)]
    else:                                            # custom sequence
        n = int(rng.integers(max(4, L // 2), 40))
        seq = [int(x) for x in rng.choice(np.arange(0x3041, 0x3041 + 80), size=n, replace=False)]
        kw = [seq[int(rng.integers(0, n))] for _ in range(L)]
        if rng.random() < 0.3:
            wildcard = ord("*")
            kw[int(rng.integers(1, L))] = wildcard
    return kw, wildcard, seq


def data_for(rng, n, hi, base_vals):
    style = rng.choice(["alphabet", "alphabet", "alphabet", "uniform", "constant", "ramp", "period"])
    if style == "uniform":
        d = rng.integers(0, hi, n)
    elif style == "alphabet":
        k = int(rng.integers(2, 8))
        d = rng.integers(0, k, n) + int(rng.integers(0, hi - k))
    elif style == "constant":
        d = np.full(n, int(rng.choice([0, hi - 1, int(rng.integers(0, hi))])))
    elif style == "ramp":
        d = (np.arange(n) * int(rng.integers(1, 3)) + int(rng.integers(0, hi))) % hi
    else:
        per = int(rng.integers(2, 5))
        d = rng.integers(0, hi, per)[np.arange(n) % per]
    d = d.astype(np.int64)
    L = len(base_vals)
    if n > L + 1:
        for _ in range(int(rng.integers(0, 6))):
            pos = int(rng.integers(0, n - L))
            sh = int(rng.integers(-40, 60))
            for j, v in enumerate(base_vals):
                if v is not None:
                    d[pos + j] = (v + sh) % hi
    return d


def main():
    ref = Ref()
    rng = np.random.default_rng(20261004)
    search, engine = [], []
    while len(search) < N_SEARCH:
        elem = 1 if rng.random() < 0.6 else 2
        hi = 256 if elem == 1 else 65536
        L = int(rng.integers(2, 14))
        values = None
        if rng.random() < 0.1:
            values = [int(rng.integers(0, 200)) for _ in range(L)]
            kw, wildcard, seq = None, 0, None
            base = values
        else:
            kw, wildcard, seq = keyword(rng, L, rng.choice(["plain", "narrow", "wild", "mixed", "seq"]))
            if not safe_kw(kw, wildcard) or (wildcard and all(c == wildcard for c in kw)):
                continue
            idx = {c: i for i, c in enumerate(seq)} if seq else None
            base = [None if (wildcard and c == wildcard) else (idx[c] if seq else c) for c in kw]
        n = int(rng.integers(1, 1024 // elem + 1))
        d = data_for(rng, n, hi, base).astype(np.uint8 if elem == 1 else "<u2")
        try:
            res = ref.value_scan(elem, values, d) if values is not None else ref.search(elem, kw, d, wildcard, seq)
        except RuntimeError:
            continue                                  # "Skip table index out of bounds": not a vector
        search.append(dict(e=elem, k=kw, w=wildcard, s=seq, v=values, d=b64(d), x=[int(t) for t in res]))
    while len(engine) < N_ENGINE:
        elem = 1 if rng.random() < 0.5 else 2
        hi = 256 if elem == 1 else 65536
        L = int(rng.integers(3, 13))
        kw, wildcard, seq = keyword(rng, L, rng.choice(["plain", "narrow", "wild", "mixed"]))
        wildcard = wildcard or ord("*")
        if not safe_kw(kw, wildcard) or all(c == wildcard for c in kw):
            continue
        nbytes = int(rng.integers(1, 1025))
        be = bool(rng.integers(0, 2)) if elem == 2 else False
        base = [None if c == wildcard else c for c in kw]
        d = data_for(rng, nbytes // elem + 2, hi, base)
        fb = d.astype(np.uint8 if elem == 1 else (">u2" if be else "<u2")).view(np.uint8)[:nbytes].copy()
        # 16-bit: shift some files by one byte so that matches sit at odd offsets too
        if elem == 2 and rng.random() < 0.5 and nbytes > 2:
            fb = np.concatenate([np.array([int(rng.integers(0, 256))], np.uint8), fb[:-1]])
        block = int(rng.choice([int(rng.integers(2 if elem == 1 else 4, 64)), int(rng.integers(64, 300)), 2 * int(rng.integers(2, 100))]))
        try:
            res = ref.engine(elem, fb, kw, wildcard, None, big_endian=be, threads=int(rng.integers(1, 4)), block_size=block)
        except RuntimeError:
            continue
        engine.append(dict(e=elem, k=kw, w=wildcard, be=be, b=block, f=b64(fb), x=[int(t) for t in res]))
    doc = dict(note="e elem bytes, k keyword code points, w wildcard, s custom sequence, v value-scan values, "
                    "d base64 of the little-endian elements / f base64 of the file bytes, be big endian, b block size, x expected offsets",
               search=search, engine=engine)
    with gzip.GzipFile(OUT, "wb", mtime=0) as f:
        f.write(json.dumps(doc, separators=(",", ":")).encode())
    print("wrote", OUT, os.path.getsize(OUT), "bytes:", len(search), "search cases (", sum(1 for c in search if c["x"]), "non-empty ),",
          len(engine), "engine cases (", sum(1 for c in engine if c["x"]), "non-empty )")


if __name__ == "__main__":
    main()
