# SPDX-License-Identifier: GPL-3.0-or-later
"""The N > 1 path on CPU: 2 processes, gloo.  Each rank takes its block-aligned partition
(mmh_partition's rule restated in tests/_gather_double.py, the torch.distributed double of the library's gather), produces its offsets (with the
oracle here -- there is no GPU in this suite) and the lists are gathered to rank 0, which
must hold exactly the offsets of the whole ROM, ascending."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_package
import _gather_double


def _worker(rank, world, port, total, block, kw, elem, be, q, width=None):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from _oracle import Oracle
    from conftest import load_package as lp
    import _gather_double
    mm = lp()
    if width:
        _gather_double.GATHER_WIDTH = width            # force the long-list (two collective) path
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = Oracle()
        plan = orc.plan(elem, kw)
        spec_all = mm.synth.RomSpec(42, total, kw, elem, None, be, block, partitions=world, runs=False)
        first, nbytes = _gather_double.shard_range(total, block, len(kw), elem, rank, world)
        spec = mm.synth.RomSpec(42, total, kw, elem, None, be, block, base=first, nbytes=nbytes, partitions=world, runs=False)
        rom = spec.host_rom()
        assert (rom == spec_all.host_rom()[first:first + nbytes]).all()       # shard generation == slice of the whole
        local = orc.engine(plan, rom, block, be) + np.uint64(first)
        merged = _gather_double.gather_offsets(local, rank, world, torch.device("cpu"), dist)
        if rank == 0:
            want = orc.engine(plan, spec_all.host_rom(), block, be)
            q.put((merged.tolist(), want.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("elem,kw,be,width", [(1, "relativesrch", False, None), (2, "textsrch", True, None),
                                               (1, "relativesrch", False, 4)])
def test_partition_and_gather_world2(elem, kw, be, width):
    import torch.multiprocessing as mp
    load_package()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total, block = (6 << 20) + 4099, 65536
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, block, kw, elem, be, q, width)) for r in range(2)]
    for p in procs:
        p.start()
    got, want = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == want
    assert len(got) >= 6 and got == sorted(got)


def _pipelined_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from conftest import load_package as lp
    import _gather_double
    mm = lp()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = _gather_double.OffsetGather(rank, world, torch.device("cpu"), dist, width=64)
        rng = np.random.default_rng(5)                      # same stream on both ranks: both know every list
        rounds = []
        for k in range(9):
            sizes = [int(rng.integers(0, 40)) if k % 3 else int(rng.integers(60, 200)) for _ in range(world)]
            lists = [np.sort(rng.choice(1 << 30, size=n, replace=False)).astype(np.uint64) + np.uint64(r << 32)
                     for r, n in enumerate(sizes)]
            rounds.append(lists)
        got, pending = [], None
        for lists in rounds:                                # the bench's pattern: start k, then finish k-1
            h = g.start(lists[rank])
            if pending is not None:
                got.append(g.finish(pending))
            pending = h
        got.append(g.finish(pending))
        if rank == 0:
            q.put([(a.tolist(), np.concatenate(l).tolist()) for a, l in zip(got, rounds)])
        else:
            assert all(x is None for x in got)
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_world2():
    """OffsetGather.start / finish with one gather in flight while the next is started (what
    bench.py does at N > 1), including rounds whose lists overflow the fixed-width record."""
    import torch.multiprocessing as mp
    load_package()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    pairs = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(pairs) == 9
    for got, want in pairs:
        assert got == want


def test_shard_ranges_tile_the_rom():
    mm = load_package()
    total, block, L = (64 << 20) + 12345, 524288, 12
    for world in (1, 2, 4, 8):
        covered = 0
        for r in range(world):
            first, n = _gather_double.shard_range(total, block, L, 1, r, world)
            assert first % block == 0 and first == covered
            nxt = _gather_double.shard_range(total, block, L, 1, r + 1, world)[0] if r + 1 < world else total
            assert first + n == min(nxt + L - 1, total)
            covered = nxt
        assert covered == total


def test_c_abi_partition_matches_the_double_and_tiles_the_rom():
    """mmh_partition (the product's rule, C) == partition.shard_range (the test double), and the
    partitions tile the file on block boundaries with the pattern-length overlap."""
    mm = load_package()
    for total in (0, 1, 100, 524288, 524289, (64 << 20) + 12345, 7 * 524288, (8 << 30) * 8, (1 << 44) + 3):
        for block in (4096, 524288, 8388608):
            for L, S in ((12, 1), (8, 2), (2, 1)):
                for world in (1, 2, 3, 8, 64):
                    covered = 0
                    for r in range(world):
                        first, n = mm.partition_range(total, block, L, S, r, world)
                        assert (first, n) == _gather_double.shard_range(total, block, L, S, r, world)
                        assert first % block == 0 and first == covered
                        nxt = mm.partition_range(total, block, L, S, r + 1, world)[0] if r + 1 < world else total
                        assert first + n == min(nxt + (L - 1) * S, total) or (n == 0 and nxt == first)
                        covered = nxt
                    assert covered == total or total == 0
    with pytest.raises(mm.MMError):
        mm.partition_range(100, 0, 12, 1, 0, 1)
    with pytest.raises(mm.MMError):
        mm.partition_range(100, 16, 12, 1, 2, 2)


def test_partition_rule_in_cpp():
    """tests/cpp/partition_tests.cpp: the same properties from C++ through the C ABI header, plus the
    per-device attribution of merged offsets SearchEngine<T>::run relies on (a match that starts in
    partition i lies wholly inside the bytes partition i holds)."""
    import subprocess
    mm = load_package()
    mm.build.build_all()
    build = os.path.join(ROOT, "tests", "cpp", "build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "partition_tests")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "partition_tests.cpp"), "-L" + mm.build.LIB_DIR, "-lmmoore_hip",
                           "-Wl,-rpath," + mm.build.LIB_DIR, "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:]
    assert " 0 failures" in r.stdout


def test_rccl_stand_in_builds_and_covers_what_the_library_calls():
    """tests/shim/libfake_rccl.so (the stand-in the GPU suite preloads so that several ranks can share one GPU): builds with
    plain g++, links against neither HIP nor RCCL, and defines every nccl* symbol libmmoore_hip.so imports."""
    import subprocess
    from conftest import build_fake_rccl
    mm = load_package()
    mm.build.build_all()
    shim = build_fake_rccl()
    defined = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", shim], text=True).splitlines() if l.strip()}
    wanted = {l.split()[-1].split("@")[0] for l in subprocess.check_output(["nm", "-D", "--undefined-only", mm.LIB_PATH], text=True).splitlines()
              if " nccl" in l}
    assert wanted and wanted <= defined, sorted(wanted - defined)
    needed = subprocess.check_output(["readelf", "-d", shim], text=True)
    assert "libamdhip64" not in needed and "librccl" not in needed
