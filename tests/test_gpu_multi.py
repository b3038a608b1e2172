# SPDX-License-Identifier: GPL-3.0-or-later
"""The multi-GPU half of the C ABI (csrc/mm_multi.hip) on the one GPU this suite has: a
communicator of one.  Everything a real 8-rank run does is exercised except the wire --
communicator bring-up through librccl, the all-gather sent from the device-side copy of the
scan's list, the packing kernel, the second (padded) phase for long lists, start / finish
overlap with two gathers in flight, mmh_scan_multi, and SearchEngine<T>::run's multi-device
path.  Needs a real MI355X: run with `pytest -m gpu`."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

BLOCK = 65536


@pytest.fixture(scope="module")
def comm_engine(mm):
    if mm.device_count() == 0:
        pytest.fail("no HIP device: the gpu tests need a real MI355X (there is no CPU fallback)")
    eng = mm.Engine(0)
    assert eng.comm_info() == (0, 0)                       # no communicator yet
    with pytest.raises(mm.MMError):
        eng.gather_start(None)                             # ... so no gather
    mm.comm_init_all([eng])
    assert eng.comm_info() == (0, 1)
    yield eng
    eng.close()


def _rom(mm, eng, nbytes, kw, elem=1, be=False, **spec_kw):
    spec = mm.synth.RomSpec(11, nbytes, kw, elem, None, be, BLOCK, **spec_kw)
    eng.alloc(nbytes)
    spec.apply_device(eng)
    return eng.download(0, nbytes)


def test_scan_then_gather_from_device(mm, comm_engine, oracle):
    eng = comm_engine
    rom = _rom(mm, eng, (8 << 20) + 777, "relativesrch")
    plan = mm.plan_relative(1, "relativesrch")
    want = oracle.engine(oracle.plan(1, "relativesrch"), rom, BLOCK)
    local = eng.scan(plan, block_bytes=BLOCK, base_offset=1 << 40)
    eng.gather_start(None)                                 # the list of that scan, straight from HBM
    merged = eng.gather_finish()
    assert merged.tolist() == local.tolist() == (want + np.uint64(1 << 40)).tolist()
    assert len(merged) >= 8
    t = eng.gather_timings()
    assert t["device_ms"] > 0 and t["host_ms"] > 0
    # ranks that only need the total
    eng.scan(plan, block_bytes=BLOCK)
    eng.gather_start(None, want_list=False)
    assert eng.gather_finish(want_list=False) == len(want)


@pytest.mark.parametrize("elem,kw,be", [(1, "relativesrch", False), (2, "textsrch", True), (1, "re*ative*ear*hxy", False)])
def test_scan_multi_communicator_of_one(mm, comm_engine, oracle, elem, kw, be):
    eng = comm_engine
    wc = ord("*") if "*" in kw else 0
    spec = mm.synth.RomSpec(5, (6 << 20) + 4099, kw, elem, wc or None, be, BLOCK)
    eng.alloc(spec.nbytes)
    spec.apply_device(eng)
    rom = eng.download(0, spec.nbytes)
    want = oracle.engine(oracle.plan(elem, kw, wc), rom, BLOCK, be)
    got = mm.scan_multi([eng], mm.plan_relative(elem, kw, wc), BLOCK, [0], big_endian=be, cap=4)   # cap 4: the capacity retry
    assert got.tolist() == want.tolist() and len(got) >= 6


def test_gather_behind_a_split_scan(mm, comm_engine, oracle):
    """mmh_scan on a ROM of >= 1 GiB runs as a pipeline of parts (the parts' lists live in three lanes' blocks): the gather of
    "the last scan's list" takes the concatenated list from the host -- short (a narrow record), long (second phase), and
    behind MMH_ROUTE_NO_SPLIT from the device again."""
    from _oracle import oracle_engine_parallel
    eng = comm_engine
    n = (1 << 30) + 3 * 524288 + 5
    spec = mm.synth.RomSpec(23, n, "relativesrch", 1, None, False, 524288, plants_per_mib=24)
    eng.alloc(n)
    spec.apply_device(eng)
    rom = eng.download(0, n)
    for kw in ("relativesrch", "elativesrch"):
        want = oracle_engine_parallel(oracle, oracle.plan(1, kw), rom, 524288)
        assert len(want) > 20000
        plan = mm.plan_relative(1, kw)
        for route in (0, mm.ROUTE_NO_SPLIT, 0):
            eng.set_route(route)
            local = eng.scan(plan, block_bytes=524288, base_offset=1 << 36, cap=1 << 16)
            assert (eng.timings()["parts"] > 0) == (route == 0)
            eng.gather_start(None)
            eng.scan(mm.plan_relative(1, "zzzzqqqq"), block_bytes=524288)          # (another scan behind the gather's start)
            merged = eng.gather_finish(cap=1 << 16)
            assert merged.tolist() == local.tolist() == (want + np.uint64(1 << 36)).tolist(), (kw, route)
    eng.set_route(0)
    eng.alloc(1 << 20)


def test_gather_of_host_lists_short_long_and_empty(mm, comm_engine):
    eng = comm_engine
    rng = np.random.default_rng(3)
    for n in (0, 1, 4223, 16384, 16385, 50000):
        offs = np.sort(rng.choice(1 << 40, size=n, replace=False)).astype(np.uint64)
        eng.gather_start(offs)
        got = eng.gather_finish(cap=16)
        assert got.dtype == np.uint64 and got.tolist() == offs.tolist(), n


def test_long_scan_lists_take_the_second_phase(mm, comm_engine, oracle):
    # a short keyword on constant data: far more matches than a gather record holds, and the
    # list only exists in host memory after the scan (radix sort / forward engine)
    eng = comm_engine
    n = 1 << 20
    eng.alloc(n)
    eng.fill(0, n, 7)
    rom = eng.download(0, n)
    want = oracle.engine(oracle.plan(1, "aaa"), rom, BLOCK)
    local = eng.scan(mm.plan_relative(1, "aaa"), block_bytes=BLOCK, cap=1 << 20)
    assert len(want) > 100000 and local.tolist() == want.tolist()
    eng.gather_start(None)
    assert eng.gather_finish(cap=1 << 20).tolist() == want.tolist()


def test_long_device_list_survives_the_scans_behind_its_gather(mm, comm_engine, oracle):
    """a list beyond a gather record (16384 slots) travels in the second phase, which mmh_gather_finish
    enqueues -- by then later scans have published into the result copy the first phase sent from (they only wait for the
    first phase).  The gather keeps its own copy of the block: what comes out is the list of the scan it was started
    for, also with holes in the slots, also when the list itself is short but its slots are many, also from a ticket."""
    eng = comm_engine
    rng = np.random.default_rng(12)
    nbytes = 24 << 20
    kw = "monkeybars"
    plan = mm.plan_relative(1, kw)
    oplan = oracle.plan(1, kw)
    other = mm.plan_relative(1, "relativesrch")

    def rom_with(every, breaks=0):
        rom = rng.integers(0, 256, nbytes).astype(np.uint8)
        for i, at in enumerate(range(100, nbytes - 64, every)):
            base = int(rng.integers(0, 200))
            rom[at:at + len(kw)] = [base + ord(c) - ord("a") for c in kw]
            if breaks and i % breaks == 0:
                rom[at] ^= 0x55                            # a candidate of the streaming filter (its key is the keyword's tail) that is no match: a hole
        return rom

    for every, breaks, via_ticket in ((1000, 0, False), (1000, 3, False), (600, 2, True), (1400, 2, False)):
        rom = rom_with(every, breaks)
        want = oracle.engine(oplan, rom, BLOCK).tolist()
        eng.upload(rom)
        if via_ticket:
            local = eng.collect(eng.submit(plan, block_bytes=BLOCK), cap=1 << 16)
        else:
            local = eng.scan(plan, block_bytes=BLOCK, cap=1 << 16)
        assert local.tolist() == want and eng.counters()["candidates"] > 16384 and eng.counters()["path"] == 0, eng.counters()
        eng.gather_start(None)
        # scans behind it on the same workspace: the second one publishes into the copy the gather was sent from
        for _ in range(2):
            assert len(eng.scan(other, block_bytes=BLOCK)) < len(want)
        assert eng.gather_finish(cap=1 << 16).tolist() == want, (every, breaks, via_ticket)


def test_two_gathers_in_flight_overlap_the_next_scan(mm, comm_engine, oracle):
    # bench.py's pattern at N > 1: scan k, start gather k, finish gather k-1
    eng = comm_engine
    rom = _rom(mm, eng, 4 << 20, "relativesrch")
    plans = [("relativesrch", 0), ("elativesrch", 0), ("re*ativesrch", ord("*")), ("relativesrc", 0), ("srch", 0)]
    wants = [oracle.engine(oracle.plan(1, kw, wc), rom, BLOCK) for kw, wc in plans]
    done, pending = [], 0
    for kw, wc in plans:
        eng.scan(mm.plan_relative(1, kw, wc), block_bytes=BLOCK)
        eng.gather_start(None)
        pending += 1
        if pending == 2:
            done.append(eng.gather_finish())
            pending -= 1
    done.append(eng.gather_finish())
    assert [d.tolist() for d in done] == [w.tolist() for w in wants]
    with pytest.raises(mm.MMError):
        eng.gather_finish()                                # nothing outstanding
    # a third start without a finish is refused, and leaves the two outstanding ones intact
    eng.scan(mm.plan_relative(1, "relativesrch"), block_bytes=BLOCK)
    eng.gather_start(None)
    eng.gather_start(np.arange(5, dtype=np.uint64))
    with pytest.raises(mm.MMError):
        eng.gather_start(None)
    assert eng.gather_finish().tolist() == wants[0].tolist()
    assert eng.gather_finish().tolist() == [0, 1, 2, 3, 4]


def test_gather_of_collected_tickets_two_scans_and_two_gathers_in_flight(mm, comm_engine, oracle):
    """bench.py's default pattern at N > 1: submit scan k, collect scan k-1, start its gather (sent from the lane's
    own device-side copy), finish gather k-2 -- with keywords whose lists differ, so that a mixed-up copy shows."""
    eng = comm_engine
    rom = _rom(mm, eng, (6 << 20) + 12345, "relativesrch")
    plans = [("relativesrch", 0), ("elativesrch", 0), ("re*ativesrch", ord("*")), ("relativesrc", 0), ("lativesrch", 0),
             ("srch", 0),                       # candidate flood: collect rescans synchronously
             ("relativesrch", 0), ("ativesrch", 0), ("rel*tivesrch", ord("*"))] * 2
    wants = [oracle.engine(oracle.plan(1, kw, wc), rom, BLOCK) for kw, wc in plans[:9]] * 2
    base = 1 << 33
    delivered, tickets, gathers = [], [], 0
    for kw, wc in plans:
        tickets.append(eng.submit(mm.plan_relative(1, kw, wc), block_bytes=BLOCK, base_offset=base))
        if len(tickets) == 2:
            local = eng.collect(tickets.pop(0))
            eng.gather_start(None)
            gathers += 1
            if gathers == 2:
                delivered.append(eng.gather_finish())
                gathers -= 1
            assert local.tolist() == (wants[len(delivered) + gathers - 1] + np.uint64(base)).tolist()
    eng.collect(tickets.pop(0))
    eng.gather_start(None)
    delivered.append(eng.gather_finish())
    delivered.append(eng.gather_finish())
    assert len(delivered) == len(plans)
    for got, want in zip(delivered, wants):
        assert got.tolist() == (want + np.uint64(base)).tolist()
    assert len({tuple(w.tolist()) for w in wants}) > 3


def test_packing_of_many_ranks_records(mm, comm_engine):
    """mm_gather_pack on the table an 8- / 3- / 64-rank all-gather would have left (the box has one GPU, so
    the wire never delivers more than one record here): dense records (the rank kernels'), sparse ones with
    holes (the tail kernel's), empty ranks, a full record, and the long-list flag."""
    eng = comm_engine
    rng = np.random.default_rng(8)
    W = mm.MMH_GATHER_RECORD_WORDS
    for nranks in (1, 3, 8, 64):
        for trial in range(4):
            rec = rng.integers(0, 1 << 62, (nranks, W), dtype=np.uint64)        # garbage wherever nothing is defined
            want, extents = [], []
            for r in range(nranks):
                kind = int(rng.integers(0, 5)) if trial else 4
                n = [0, 1, int(rng.integers(2, 5000)), 16384, int(rng.integers(2, 9000))][kind]
                base = np.uint64(r) << np.uint64(40)
                offs = np.sort(rng.choice(1 << 30, size=n, replace=False)).astype(np.uint64) + base
                rec[r, :8] = 0
                if kind == 4 and n > 1:
                    # sparse: more slots than matches, ~0 in the holes
                    slots = n + int(rng.integers(1, 200))
                    slots = min(slots, 16384)
                    hole = np.ones(slots, bool)
                    hole[np.sort(rng.choice(slots, size=n, replace=False))] = False
                    body = np.full(slots, np.uint64(0xFFFFFFFFFFFFFFFF))
                    body[~hole] = offs
                    rec[r, 8:8 + slots] = body
                    rec[r, 0], rec[r, 4], rec[r, 6] = slots, 1, n + 1
                    extents.append(slots)
                else:
                    rec[r, 8:8 + n] = offs
                    rec[r, 0], rec[r, 4], rec[r, 6] = n + int(rng.integers(0, 3)), 0, n + 1   # (dense: word 0 may count dropped slots too)
                    extents.append(n)
                want.append(offs)
            got, longest = eng.selftest_gather_pack(rec)
            flat = np.concatenate(want)
            assert longest == max(extents)           # what a record has to hold: the list, or one slot per candidate when it has holes
            assert got.tolist() == flat.tolist(), (nranks, trial)
            assert (np.diff(got.astype(np.int64)) > 0).all() or got.size < 2
    # a rank with a list longer than a record: nothing is delivered, the second phase would follow
    rec = np.zeros((2, W), np.uint64)
    rec[0, 6], rec[1, 6] = 5 + 1, 20000 + 1
    got, longest = eng.selftest_gather_pack(rec)
    assert got is None and longest == 20000


def test_native_gather_next_to_torch_nccl():
    # the way bench.py runs at N > 1: torch.distributed ("nccl") owns the rendezvous and the
    # barriers, the library brings up its OWN RCCL communicator from an id that travels through
    # torch.distributed, both in one process
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_native_gather_check.py")], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "native gather ok" in r.stdout


def _run_ranks(script, nranks, extra_env=None, timeout=420, stand_in=True):
    """nranks processes of a tests/_*.py script on GPU 0 under the RCCL stand-in (stand_in=False: no preload -- librccl itself,
    the script puts every rank on a GPU of its own); every one must exit 0.  Returns their output."""
    import glob
    import tempfile
    from conftest import build_fake_rccl
    rdv = tempfile.mkdtemp(prefix="mm_rdv_")
    env = dict(os.environ, MMOORE_GATHER_TIMEOUT_S="60", **(extra_env or {}))
    if stand_in:
        env["LD_PRELOAD"] = build_fake_rccl()
    else:
        env.pop("LD_PRELOAD", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", script), str(r), str(nranks), rdv], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=env, cwd=ROOT) for r in range(nranks)]
    outs, failed = [], False
    import time
    deadline = time.time() + timeout
    try:
        for p in procs:
            try:
                outs.append(p.communicate(timeout=max(1.0, deadline - time.time()))[0])
            except subprocess.TimeoutExpired:
                failed = True
                break
            failed = failed or p.returncode != 0
            if failed:
                break                                  # (its peers wait for it in a collective: end them)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                               # exactly the processes started here
                outs.append(p.communicate()[0])
        for f in glob.glob(os.path.join(rdv, "*")):
            os.unlink(f)
        os.rmdir(rdv)
        for f in glob.glob("/dev/shm/fake-rccl-*"):
            try:
                os.unlink(f)
            except OSError:
                pass
    text = "\n".join("---- rank %d (rc %s)\n%s" % (i, procs[i].returncode, o[-3000:]) for i, o in enumerate(outs))
    assert not failed and all(p.returncode == 0 for p in procs), text
    return outs


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_native_gather_two_ranks_one_gpu(mm, nranks):
    """The library's own communicator and gather (csrc/mm_multi.hip) with MORE THAN ONE RANK: nranks processes on the one
    GPU, RCCL's nine entry points served by tests/shim/fake_rccl.cpp (stream-ordered, asynchronous, shared memory; RCCL
    itself refuses ranks that share a device).  Every rank scans its mmh_partition; gathers in bench.py's order (three
    tickets outstanding, two gathers in flight), scans between a gather's start and finish, lists beyond a record on one
    rank (device-side copy kept; host list), callers' host lists, ranks with empty lists and with nothing to scan --
    every merged list against the oracle on the WHOLE ROM, on every rank."""
    outs = _run_ranks("_two_rank_gather.py", nranks)
    for r, o in enumerate(outs):
        assert "rank %d of %d ok: rccl_ranks %d" % (r, nranks, nranks) in o, o[-2000:]


def _visible_gpus(mm):
    return mm.device_count()


@pytest.mark.parametrize("want", [2, 8])
def test_native_gather_over_rccl_one_rank_per_gpu(mm, want):
    """ARMS ITSELF on a node with >= 2 GPUs (skipped on the one-GPU pool): the same script, one rank per GPU, NO stand-in --
    the library's communicator and gathers over librccl and xGMI.  The contract is the reference dispatcher's and merge's
    (src/core/search_engine.cpp:66-188, :193-197): every block searched once, the merged list ascending -- every gathered
    list against the oracle over the WHOLE ROM, on every rank.  want = 2, and as many ranks as there are GPUs (at most 8)."""
    have = _visible_gpus(mm)
    if have < 2:
        pytest.skip("%d GPU visible: RCCL refuses ranks that share a device (the stand-in tests above cover N > 1 here)" % have)
    nranks = 2 if want == 2 else min(8, have)
    if want == 8 and nranks == 2:
        pytest.skip("2 GPUs: the two-rank run above is all there is")
    outs = _run_ranks("_two_rank_gather.py", nranks, extra_env={"MM_RANK_OWN_GPU": "1"}, timeout=900, stand_in=False)
    for r, o in enumerate(outs):
        assert "rank %d of %d ok: rccl_ranks %d" % (r, nranks, nranks) in o, o[-2000:]


@pytest.mark.parametrize("want", [2, 8])
def test_bench_over_rccl_one_rank_per_gpu(mm, want):
    """ARMS ITSELF on a node with >= 2 GPUs: `bench.py --gpus N` the way the driver launches it (one rank per GPU, librccl,
    no stand-in) -- rccl_ranks == N on every rank, the native gather identical to the torch.distributed double's, weak and
    strong legs, rank 0's merged list holding every partition's plants."""
    import json
    have = _visible_gpus(mm)
    if have < 2:
        pytest.skip("%d GPU visible" % have)
    nranks = 2 if want == 2 else min(8, have)
    if want == 8 and nranks == 2:
        pytest.skip("2 GPUs: the two-rank run above is all there is")
    env = dict(os.environ, MASTER_PORT=str(29300 + os.getpid() % 300))
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--gib-per-gpu", "0.25", "--steps", "6",
                        "--warmup", "2", "--prewarm-s", "0.02", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    res = json.loads(line[0])
    assert res["n_gpus"] == nranks and res["value"] > 0 and "shared_device" not in res
    assert res["rccl_ranks"] == {"min": nranks, "max": nranks, "expected": nranks}
    assert "librccl" in res["gather_backend"]
    assert "identical" in res["gather_check"] and "%d ranks" % nranks in res["gather_check"]
    assert res["config"]["matches"] > 250 * nranks
    assert res["strong"]["n_gpus"] == nranks and res["strong"]["matches"] > 250


@pytest.mark.parametrize("n", [2, 4, 8])
def test_scan_multi_two_contexts_one_gpu(mm, n):
    """mmh_comm_init_all + mmh_scan_multi with n > 1 contexts (one process, one host thread per context, the collective
    inside ncclGroupStart / ncclGroupEnd, the padded second phase inside the group) -- on one GPU through the stand-in."""
    from conftest import build_fake_rccl
    env = dict(os.environ, LD_PRELOAD=build_fake_rccl())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_two_contexts_one_gpu.py"), str(n)], capture_output=True, text=True,
                       timeout=420, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "scan_multi over %d contexts on one GPU ok" % n in r.stdout


def test_search_engine_run_multi_device_path(mm):
    # SearchEngine<T>::run through mmh_comm_init_all + mmh_scan_multi (MMOORE_HIP_MULTI=1 takes that
    # path with the one device there is): the reference's own engine vectors, previews, progress, abort
    from test_facade import _build_tests
    exe = _build_tests(mm)
    env = dict(os.environ, MMOORE_HIP_MULTI="1")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 failures" in r.stdout


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_search_engine_run_over_several_contexts_one_gpu(mm, devices):
    """SearchEngine<T>::run's multi-device path with MORE THAN ONE device context: MMOORE_HIP_DEVICE_LIST names GPU 0 two / three
    times (RCCL would refuse that: the stand-in of tests/shim serves the communicator), MMOORE_HIP_MULTI=1 takes the path
    whatever the file's size -- partition rounds over the contexts, one ingest thread per context, mmh_scan_multi, the
    merged list, previews, progress and abort: the facade's own test program (the reference's vectors among them)."""
    from conftest import build_fake_rccl
    from test_facade import _build_tests
    exe = _build_tests(mm)
    env = dict(os.environ, MMOORE_HIP_MULTI="1", MMOORE_HIP_DEVICE_LIST=devices, LD_PRELOAD=build_fake_rccl(), FAKE_RCCL_TRACE="1")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 failures" in r.stdout
    n = devices.count(",") + 1
    assert "fake_rccl: rank %d / %d up" % (n - 1, n) in r.stdout and "process-local all-gather of %d ranks" % n in r.stdout


@pytest.mark.parametrize("extra", [[], ["--depth", "1"], ["--depth", "2"], ["--sync-gather"], ["--torch-gather"]])
def test_bench_multi_rank_path_with_one_rank(extra):
    """bench.py's N > 1 code path (process group, unique id, communicator, overlapped gather of collected
    tickets, the native-vs-torch gather check) with the one rank the box has."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-gather", "--gib-per-gpu", "0.25", "--steps", "6", "--warmup", "2",
           "--prewarm-s", "0.02", "--no-cpu-baseline"] + extra
    env = dict(os.environ, MASTER_PORT=str(29600 + os.getpid() % 300))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    res = json.loads(line[0])
    assert res["n_gpus"] == 1 and res["steps"] == 6 and res["value"] > 0
    assert res["config"]["matches"] == 256 + 3 + 3 or res["config"]["matches"] > 200       # 1 plant / MiB + straddlers
    depth = int(extra[1]) if "--depth" in extra else 3
    assert res["config"]["scans_in_flight"] == depth
    assert res["overlap"] == ("--sync-gather" not in extra)
    assert "gather_note" not in res
    if "--torch-gather" in extra:
        assert "test double" in res["gather_backend"] and "gather_check" not in res
    else:
        assert "librccl" in res["gather_backend"]
        assert "identical" in res["gather_check"]
        assert res["gather_ms"]["device_collective_and_pack"] > 0
    other = res["in_flight" if depth == 1 else "synchronous"]
    assert other["same_offsets"] is True


@pytest.mark.parametrize("nranks", [2, 8])
def test_bench_two_ranks_one_gpu_through_the_stand_in(nranks):
    """`bench.py --gpus N` end to end with N REAL ranks (torch.distributed.run, gloo for the rendezvous, the library's own
    communicator + gathers over the RCCL stand-in, all on GPU 0): weak and strong legs, gather_check against the
    torch.distributed double, rccl_ranks == N on every rank -- N = 2, and N = 8 (the driver's largest run: a communicator of
    eight, 32 MiB partitions in the strong leg).  And: the stand-in is refused where it was not asked for."""
    import json
    from conftest import build_fake_rccl
    env = dict(os.environ, LD_PRELOAD=build_fake_rccl())
    common = ["--gib-per-gpu", "0.25", "--steps", "6", "--warmup", "2", "--prewarm-s", "0.02", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--allow-shared-device"] + common,
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    res = json.loads(line[0])
    assert res["n_gpus"] == nranks and res["value"] > 0 and "shared_device" in res
    assert res["rccl_ranks"] == {"min": nranks, "max": nranks, "expected": nranks}
    assert "identical" in res["gather_check"] and "%d ranks" % nranks in res["gather_check"]
    assert res["config"]["matches"] > 250 * nranks                          # every partition's plants in rank 0's merged list
    assert res["gather_ms"]["device_collective_and_pack"] > 0
    assert res["strong"]["n_gpus"] == nranks and res["strong"]["matches"] > 250 and res["strong"]["gather_ms"]["device_collective_and_pack"] > 0
    if nranks != 2:
        return
    # one rank, stand-in loaded, not asked for: refused before anything is timed
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-pmc"] + common[:2] + common[6:],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "stand-in" in (r.stderr + r.stdout) and not [l for l in r.stdout.splitlines() if l.startswith("{")]
