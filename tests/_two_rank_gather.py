# SPDX-License-Identifier: GPL-3.0-or-later
"""One rank of tests/test_gpu_multi.py::test_native_gather_two_ranks_one_gpu.

    LD_PRELOAD=tests/shim/libfake_rccl.so python tests/_two_rank_gather.py RANK NRANKS RENDEZVOUS_DIR

NRANKS processes on GPU 0, each with the library's own communicator (csrc/mm_multi.hip) over the test-only RCCL
stand-in of tests/shim/fake_rccl.cpp (RCCL proper refuses two ranks on one device).  No torch: the id travels through
a file.  Every rank builds the same ROMs from a seed, uploads ITS mmh_partition, and holds every gathered list
against the oracle's run over the whole ROM -- the reference's dispatcher + merge, src/core/search_engine.cpp:66-188,
:193-197, on more than one rank for the first time."""
import ctypes
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402
from _oracle import Oracle  # noqa: E402

BLOCK = 524288


def main():
    rank, nranks, rdv = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    # MM_RANK_OWN_GPU=1: one rank per GPU over librccl itself (tests/test_gpu_multi.py arms that on a node with >= 2 GPUs)
    own_gpu = os.environ.get("MM_RANK_OWN_GPU") == "1"
    try:
        stand_in = ctypes.CDLL(None).fake_rccl_loaded() == 1
    except (AttributeError, OSError):
        stand_in = False
    assert stand_in != own_gpu, "the RCCL stand-in is %s" % ("preloaded, and RCCL itself was asked for" if stand_in else "not preloaded")
    mm = load_package()
    orc = Oracle()
    eng = mm.Engine(rank if own_gpu else 0)
    assert eng.comm_info() == (0, 0)

    # rendezvous: rank 0 makes the id, the others read it from the file
    id_path = os.path.join(rdv, "id")
    if rank == 0:
        with open(id_path + ".tmp", "wb") as f:
            f.write(mm.comm_unique_id())
        os.rename(id_path + ".tmp", id_path)
    t0 = time.time()
    while not os.path.exists(id_path):
        assert time.time() - t0 < 60, "no id from rank 0"
        time.sleep(0.01)
    with open(id_path, "rb") as f:
        uid = f.read()
    eng.comm_init_rank(uid, nranks, rank)
    assert eng.comm_info() == (rank, nranks)

    def load(rom, L, elem=1):
        """this rank's partition of `rom` becomes the engine's ROM; returns its base offset"""
        first, n = mm.partition_range(len(rom), BLOCK, L, elem, rank, nranks)
        if n:
            eng.upload(rom[first:first + n])
        else:
            eng.alloc(0)
        return first

    def owned(want, total_bytes, r):
        """the entries of a whole-ROM list that rank r's partition reports: matches that START in one of its blocks"""
        nblocks = -(-total_bytes // BLOCK)
        b0, b1 = nblocks * r // nranks, nblocks * (r + 1) // nranks
        blk = want // np.uint64(BLOCK)
        return want[(blk >= b0) & (blk < b1)]

    def check(got, want, what):
        assert got.dtype == np.uint64 and got.tolist() == want.tolist(), "rank %d, %s: %d offsets, expected %d" % (rank, what, len(got), len(want))

    rng = np.random.default_rng(2024)
    total = (64 << 20) + 3 * BLOCK + 777
    rom = rng.integers(0, 256, total).astype(np.uint8)
    rom[3 << 20: 4 << 20] = 0                                              # padding runs, one in either half: floods for short keywords
    rom[40 << 20: 41 << 20] = np.arange(1 << 20, dtype=np.uint64).astype(np.uint8)
    keywords = [("relativesrch", 0), ("elativesrch", 0), ("re*ativesrch", ord("*")), ("relativesrc", 0), ("lativesrch", 0), ("rel*tivesrch", ord("*"))]
    # plants of the longest keyword all over, also across every partition's edge; the shorter keywords match inside them
    kw0 = np.frombuffer(b"relativesrch", np.uint8).astype(np.int64)
    for at in list(range(5000, total - 64, 1 << 19)) + [(total // BLOCK // nranks) * BLOCK * g - 5 for g in range(1, nranks)]:
        rom[at:at + 12] = (kw0 + int(rng.integers(-90, 100))).astype(np.uint8)
    wants = [orc.engine(orc.plan(1, kw, wc), rom, BLOCK) for kw, wc in keywords]
    assert all(len(w) >= 100 for w in wants) and len({tuple(w.tolist()) for w in wants}) >= 3   # (a mixed-up copy shows)
    base = load(rom, 12)

    # 1. bench.py's order at N > 1: three tickets outstanding, collect k, start gather k, finish gather k - 1
    plans = [mm.plan_relative(1, kw, wc) for kw, wc in keywords] * 3
    tickets, gathers, delivered = [], 0, []
    for i, plan in enumerate(plans):
        tickets.append(eng.submit(plan, block_bytes=BLOCK, base_offset=base))
        if len(tickets) == 3:
            eng.collect(tickets.pop(0))
            eng.gather_start(None, want_list=True)
            gathers += 1
            if gathers == 2:
                delivered.append(eng.gather_finish())
                gathers -= 1
    while tickets:
        eng.collect(tickets.pop(0))
        eng.gather_start(None)
        gathers += 1
        if gathers == 2:
            delivered.append(eng.gather_finish())
            gathers -= 1
    delivered.append(eng.gather_finish())
    assert len(delivered) == len(plans)
    for i, got in enumerate(delivered):
        check(got, wants[i % len(wants)], "tickets, step %d" % i)

    # 2. synchronous scans, ranks that only want the count, and scans (one of them a flood that retries through
    #    other engines) BETWEEN a gather's start and its finish on one rank only
    flood = mm.plan_relative(1, "abcd")                                     # a million candidates on the ramp: bucket overflow, flood path
    for i, (kw, wc) in enumerate(keywords):
        local = eng.scan(mm.plan_relative(1, kw, wc), block_bytes=BLOCK, base_offset=base)
        check(local, owned(wants[i], total, rank), "its own partition's list, step %d" % i)
        want_list = rank == 0 or i % 2 == 0
        eng.gather_start(None, want_list=want_list)
        if rank == (i % nranks):
            eng.scan(flood, block_bytes=BLOCK, base_offset=base)            # publishes into the result copies again, and again
            eng.scan(mm.plan_relative(1, "relativesrch"), block_bytes=BLOCK, base_offset=base)
        got = eng.gather_finish(want_list=want_list)
        if want_list:
            check(got, wants[i], "synchronous, step %d" % i)
        else:
            assert got == len(wants[i]), (rank, i, got, len(wants[i]))

    # 3. a list beyond a gather record on ONE rank (the last): 16384 < n <= 262144 slots stay on the device and take the
    #    second, padded phase from the copy the gather kept; the other ranks' lists are short
    kw = "monkeybars"
    rom2 = rng.integers(0, 256, 48 << 20).astype(np.uint8)
    letters = np.frombuffer(kw.encode(), np.uint8).astype(np.int64) - ord("a")
    lo2 = (len(rom2) // BLOCK) * (nranks - 1) // nranks * BLOCK
    for j, at in enumerate(range(lo2 + 100, len(rom2) - 64, 400 if nranks <= 3 else 128)):   # (the last rank's share shrinks with the ranks)
        rom2[at:at + len(kw)] = (letters + int(rng.integers(0, 200))).astype(np.uint8)
        if j % 3 == 0:
            rom2[at] ^= 0x55                                               # a candidate that is no match: a hole in the slots
    want2 = orc.engine(orc.plan(1, kw), rom2, BLOCK)
    assert len(want2) > 16384
    base2 = load(rom2, len(kw))
    plan2 = mm.plan_relative(1, kw)
    for via_ticket in (False, True):
        if via_ticket:
            eng.collect(eng.submit(plan2, block_bytes=BLOCK, base_offset=base2), cap=1 << 17)
        else:
            eng.scan(plan2, block_bytes=BLOCK, base_offset=base2, cap=1 << 17)
        eng.gather_start(None)
        for _ in range(2):
            eng.scan(mm.plan_relative(1, "relativesrch"), block_bytes=BLOCK, base_offset=base2)   # behind it, on the same workspace
        check(eng.gather_finish(cap=1 << 17), want2, "long device list, ticket %s" % via_ticket)

    # 4. a list that only exists in host memory (more matches than the published block holds), on rank 0 this time; and
    #    a host list handed to the gather by the caller
    rom3 = rng.integers(0, 256, 8 << 20).astype(np.uint8)
    rom3[100000:100000 + (1 << 20)] = 7
    want3 = orc.engine(orc.plan(1, "aaa"), rom3, BLOCK)
    assert len(want3) > 262144
    base3 = load(rom3, 3)
    local3 = eng.scan(mm.plan_relative(1, "aaa"), block_bytes=BLOCK, base_offset=base3, cap=1 << 20)
    eng.gather_start(None)
    check(eng.gather_finish(cap=1 << 20), want3, "long host list")
    eng.gather_start(local3[: 20000 + rank])                               # caller's lists: rank r sends 20000 + r offsets
    got = eng.gather_finish(cap=1 << 20)
    check(local3, owned(want3, len(rom3), rank), "its own partition's long list")
    check(got, np.concatenate([owned(want3, len(rom3), r)[: 20000 + r] for r in range(nranks)]), "caller's host lists")

    # 5. empty lists: a keyword found in the first rank's partition only, then in nobody's; a ROM of ONE block (every
    #    rank but the last has nothing to scan at all)
    rom4 = rng.integers(0, 256, 16 << 20).astype(np.uint8)
    rom4[70000:70012] = (kw0 + 3).astype(np.uint8)
    base4 = load(rom4, 12)
    for word in ("relativesrch", "qzjxkvwpqzjx"):
        want4 = orc.engine(orc.plan(1, word), rom4, BLOCK)
        local = eng.scan(mm.plan_relative(1, word), block_bytes=BLOCK, base_offset=base4)
        check(local, owned(want4, len(rom4), rank), "its own partition's list (empty on all ranks but the first)")
        assert len(local) == (len(want4) if rank == 0 else 0)
        eng.gather_start(None)
        check(eng.gather_finish(), want4, "empty lists, " + word)
    rom5 = rom4[: BLOCK - 100].copy()
    first5, n5 = mm.partition_range(len(rom5), BLOCK, 12, 1, rank, nranks)
    assert (n5 == 0) == (rank < nranks - 1)
    base5 = load(rom5, 12)
    want5 = orc.engine(orc.plan(1, "relativesrch"), rom5, BLOCK)
    assert len(want5) == 1
    eng.scan(mm.plan_relative(1, "relativesrch"), block_bytes=BLOCK, base_offset=base5)
    eng.gather_start(None)
    check(eng.gather_finish(), want5, "one block, %d ranks" % nranks)

    t = eng.gather_timings()
    assert t["device_ms"] > 0 and t["host_ms"] > 0
    health = eng.health()
    assert health["fallback_reason"] == 0, health
    assert eng.comm_info() == (rank, nranks)
    eng.close()
    print("rank %d of %d ok: rccl_ranks %d, %d gathers checked against the oracle" % (rank, nranks, nranks, len(plans) + len(keywords) + 7), flush=True)


if __name__ == "__main__":
    main()
