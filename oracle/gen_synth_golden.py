#!/usr/bin/env python3
# SPDX-License-Identifier: GPL-3.0-or-later
"""Golden vectors for the SYNTHETIC bench ROMs: what the compiled reference
(oracle/_ref, the unmodified sources of /root/reference) reports on 16 MiB instances of the
BASELINE configurations, with the SHA-256 of the regenerated buffer.

    python oracle/gen_synth_golden.py        # -> tests/golden/synth_roms.json

Inputs are regenerated from (seed, size, keyword) by monkey-moore_amd/synth.py (host side) or
mm_synth_fill + pokes (device side); only parameters and expected offsets are stored.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from _oracle import Ref  # noqa: E402
from conftest import load_package  # noqa: E402

M16, M256 = 16 << 20, 256 << 20
CASES = [
    # name, elem, keyword, wildcard, big_endian, nbytes, block sizes (G5: the library default, the GUI default, one block = the file)
    ("C2 shape: 8-bit, 12 symbols", 1, "relativesrch", None, False, M16, [524288, 8388608, M16]),
    ("C3 shape: 8-bit, 16 symbols, 3 wildcards", 1, "re*ative*ear*hxy", ord("*"), False, M16, [524288, 8388608, M16]),
    ("C4 shape: 16-bit LE, 8 symbols", 2, "textsrch", None, False, M16, [524288, 8388608, M16]),
    ("C4 shape: 16-bit BE, 8 symbols", 2, "textsrch", None, True, M16, [524288, 8388608, M16]),
    ("ragged: 8-bit, odd size, small blocks", 1, "relativesrch", None, False, (3 << 20) + 4099, [65536, 8191]),
    # G4 at 256 MiB
    ("C2 shape at 256 MiB", 1, "relativesrch", None, False, M256, [524288, 8388608, M256]),
    ("C3 shape at 256 MiB", 1, "re*ative*ear*hxy", ord("*"), False, M256, [524288, 8388608]),
    ("C4 shape at 256 MiB: 16-bit LE", 2, "textsrch", None, False, M256, [524288, 8388608]),
]


def main():
    mm = load_package()
    ref = Ref()
    out = []
    for name, elem, kw, wc, be, n, blocks in CASES:
        spec = mm.synth.RomSpec(42, n, kw, elem, wc, be, 524288)
        rom = spec.host_rom()
        entry = dict(name=name, seed=42, nbytes=n, elem_bytes=elem, keyword=kw, wildcard=wc, big_endian=be,
                     sha256=hashlib.sha256(rom.tobytes()).hexdigest(), engine={})
        for b in blocks:
            offs = ref.engine(elem, rom, kw, wc if wc is not None else ord("*"), None, big_endian=be, threads=8, block_size=b)
            entry["engine"][str(b)] = [int(x) for x in offs]
        if not be and n <= M16:
            data = rom[: (n // elem) * elem].view(np.uint8 if elem == 1 else "<u2")
            entry["whole_buffer"] = [int(x) for x in ref.search(elem, kw, data, wc or 0)]
        out.append(entry)
        print(name, {k: len(v) for k, v in entry["engine"].items()}, len(entry.get("whole_buffer", [])))
    # C1: the reference benchmark's own input (bench_search.cpp:11-22), whole-buffer chain
    for elem in (1, 2):
        data = ref.bench_data(elem, 16 << 20)
        view = data.view(np.uint8 if elem == 1 else "<u2")
        entry = dict(name="C1: mt19937(42) benchmark buffer, %d-bit" % (8 * elem), nbytes=16 << 20, elem_bytes=elem,
                     sha256=hashlib.sha256(data.tobytes()).hexdigest(), bench_data=True, search={})
        for kw in ("abcde", "monkey", "relativesrch"):
            entry["search"][kw] = [int(x) for x in ref.search(elem, kw, view)]
        out.append(entry)
        print(entry["name"], {k: len(v) for k, v in entry["search"].items()})
    path = os.path.join(ROOT, "tests", "golden", "synth_roms.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
