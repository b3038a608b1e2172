# SPDX-License-Identifier: GPL-3.0-or-later
"""Run by tests/test_gpu_multi.py::test_scan_multi_two_contexts_one_gpu under LD_PRELOAD=tests/shim/libfake_rccl.so:
ONE process, N contexts on GPU 0 as ranks 0 .. N-1 of an mmh_comm_init_all communicator (RCCL proper wants N devices),
mmh_scan_multi = one host thread per context for the scans + the grouped collective (ncclGroupStart / ncclGroupEnd) +
the delivery through rank 0 -- src/core/search_engine.cpp:66-188, :193-197 with more than one worker."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402
from _oracle import Oracle  # noqa: E402

BLOCK = 65536


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    assert ctypes.CDLL(None).fake_rccl_loaded() == 1, "the RCCL stand-in is not preloaded"
    mm = load_package()
    orc = Oracle()
    engines = [mm.Engine(0) for _ in range(n)]
    mm.comm_init_all(engines)
    assert [e.comm_info() for e in engines] == [(r, n) for r in range(n)]
    rng = np.random.default_rng(77)
    checked = 0
    for elem, kw, wc, be, nbytes in ((1, "relativesrch", 0, False, (24 << 20) + 4099), (2, "textsrch", 0, True, (12 << 20) + 2 * BLOCK + 6),
                                      (1, "re*ative*ear*hxy", ord("*"), False, (9 << 20) + 1), (1, "monkeybars", 0, False, 40 << 20),
                                      (1, "aaa", 0, False, 6 << 20), (1, "relativesrch", 0, False, BLOCK - 7)):
        spec = mm.synth.RomSpec(5 + checked, nbytes, kw, elem, wc or None, be, BLOCK, partitions=n, plants_per_mib=2)
        rom = spec.host_rom()
        if kw == "monkeybars":
            # more than a gather record holds on the last rank: the second, padded phase inside the group
            letters = np.frombuffer(kw.encode(), np.uint8).astype(np.int64) - ord("a")
            for at in range(nbytes * (n - 1) // n + 1000, nbytes - 64, 600):
                rom[at:at + len(kw)] = (letters + int(rng.integers(0, 200))).astype(np.uint8)
        if kw == "aaa":
            rom[1 << 20: 2 << 20] = 9                  # more matches than a published block holds: a host list on rank 0
        want = orc.engine(orc.plan(elem, kw, wc), rom, BLOCK, be)
        bases = []
        for r, e in enumerate(engines):
            first, nb = mm.partition_range(nbytes, BLOCK, len(kw), elem, r, n)
            bases.append(first)
            if nb:
                e.upload(rom[first:first + nb])
            else:
                e.alloc(0)
        plan = mm.plan_relative(elem, kw, wc)
        for cap in (4, 1 << 20):                        # (4: the capacity retry)
            got = mm.scan_multi(engines, plan, BLOCK, bases, big_endian=be, cap=cap)
            assert got.tolist() == want.tolist(), (kw, len(got), len(want))
            checked += 1
        assert len(want) >= 1
    for e in engines:
        assert e.health()["fallback_reason"] == 0
        e.close()
    print("scan_multi over %d contexts on one GPU ok: %d gathered lists checked against the oracle" % (n, checked), flush=True)


if __name__ == "__main__":
    main()
