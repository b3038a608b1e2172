#!/usr/bin/env python3
# SPDX-License-Identifier: GPL-3.0-or-later
"""bench.py -- GB/s scanned by the MI355X relative-search engine.

  python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5]
  N > 1 without RANK / WORLD_SIZE in the environment: bench.py launches its N ranks itself
  (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...) before anything
  touches the GPU, passes rank 0's JSON line through and exits with the children's code.  Under a
  launcher WORLD_SIZE must equal --gpus, else the run is refused.

Workload (BASELINE.json configs[1], "C2"): 8-bit relative search, 12-character
keyword, engine semantics with the reference's default 512 KiB blocks, on a
synthetic ROM already resident in HBM.  One step = one full scan of the ROM:
filter kernel + resolver + ordering + D2H of the offsets, plus -- at N > 1 --
the RCCL gather of the per-GPU offset lists, issued by the library itself from
device memory (include/mmoore_hip.h: mmh_gather_start / mmh_gather_finish).
Weak scaling (`value`): every GPU holds its own partition (block-aligned, pattern-length
overlap) of an N x 4 GiB ROM (--config C5: N x 8 GiB, BASELINE.json configs[4]).
Strong scaling (`strong`, in the same line; `--scaling strong` makes it the value): ONE
4 GiB ROM dealt over the N GPUs with mmh_partition, the way the reference's dispatcher
deals one file over its workers (src/core/search_engine.cpp:66-188, :218-253).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel
(mm_filter_u8) with HIP events recorded on the scan's own stream;
`cpu_baseline` times the reference's multi-threaded SearchEngine<uint8_t>::run
(oracle/_ref, built from the unmodified reference sources) on the same 4 GiB ROM
in a tmpfs file on the host cores, as BASELINE.md section 3 prescribes -- a
reported baseline, not the target.
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLOCK = 524288
SEED = 42
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak 8 TB/s (measured achievable ~6.3)
# BASELINE.json configs[1..4]: GiB per GPU, element bytes, keyword, wildcard, byte order
CONFIGS = {
    "C2": dict(gib=4.0, elem=1, keyword="relativesrch", wildcard=None, be=False,
               what="8-bit relative search, keyword 'relativesrch' (L=12)"),
    "C3": dict(gib=4.0, elem=1, keyword="re*ative*ear*hxy", wildcard=ord("*"), be=False,
               what="8-bit relative search, keyword 're*ative*ear*hxy' (L=16, 3 wildcards)"),
    "C4": dict(gib=8.0, elem=2, keyword="textsrch", wildcard=None, be=False,
               what="16-bit little-endian relative search, keyword 'textsrch' (L=8), both byte alignments"),
    "C4BE": dict(gib=8.0, elem=2, keyword="textsrch", wildcard=None, be=True,
                 what="16-bit big-endian relative search, keyword 'textsrch' (L=8), both byte alignments"),
    "C5": dict(gib=8.0, elem=1, keyword="relativesrch", wildcard=None, be=False,
               what="8-bit relative search, keyword 'relativesrch' (L=12), one GPU's shard of the 64 GiB / 8 GPU configuration"),
}
OTHER_CONFIGS = ("C3", "C4", "C4BE")   # measured behind the timed region of the default (C2, N = 1) run


def rccl_stand_in_loaded():
    """True when tests/shim/libfake_rccl.so (the TEST-ONLY stand-in for RCCL, LD_PRELOADed) serves this process."""
    import ctypes
    try:
        return ctypes.CDLL(None).fake_rccl_loaded() == 1
    except (AttributeError, OSError):
        return False


def launch_ranks(n, argv, dry, shared_device=False):
    """--gpus N > 1 outside a launcher: start the N ranks as children of THIS process -- which has neither
    imported torch nor touched the GPU (a process that has must not start another program in its place) --
    and hand their exit code back.  Rank 0's JSON line reaches stdout through the inherited descriptor."""
    import socket
    import subprocess
    if not dry:
        # (torch.cuda.device_count() does not initialise the GPU on this stack; it still runs in a child)
        probe = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        have = int(probe.stdout.strip() or 0) if probe.returncode == 0 else 0
        if have < (1 if shared_device else n):
            sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible on this node -- refusing to run (one rank per GPU, "
                             "no oversubscription, no CPU fallback)\n" % (n, have))
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stderr.write("bench.py: launching %d ranks: %s\n" % (n, " ".join(cmd)))
    return subprocess.run(cmd, env=env).returncode


class c_stdout_to_stderr:
    """What C libraries printf to stdout inside the block goes to stderr instead: RCCL prints a version banner from its
    communicator bring-up, and rank 0's stdout is ONE JSON line.  (Python's own stdout buffer is flushed first; the C
    stdio buffers are flushed while file descriptor 1 points at stderr.)"""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self.libc = ctypes.CDLL(None)
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *a):
        self.libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)


def _timed(fn, warmups, runs):
    for _ in range(warmups):
        fn()
    times, last = [], None
    for _ in range(runs):
        t0 = time.perf_counter()
        last = fn()
        times.append(time.perf_counter() - t0)
    return times, last


class Facade:
    """libmonkey-core.so through its C bindings (host/c_bindings.cpp): SearchEngine<T>::run of THIS repository."""

    def __init__(self, mm):
        import ctypes as C
        self.C = C
        self.lib = C.CDLL(mm.build.CORE_SO)
        u32p, u64p, i16p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int16)
        self.lib.mmoore_c_last_error.restype = C.c_char_p
        self.lib.mmoore_c_engine_run.restype = C.c_int64
        self.lib.mmoore_c_engine_run.argtypes = [C.c_int, C.c_char_p, u32p, C.c_int, C.c_uint32, u32p, C.c_int, i16p, C.c_int, C.c_int,
                                                 C.c_int, C.c_int, C.c_int, u64p, C.c_uint64]
        self.lib.mmoore_c_result_maps.restype = C.c_int64
        self.lib.mmoore_c_result_maps.argtypes = [u32p, u32p, C.c_uint64, C.c_int]

    def engine(self, elem, path, keyword, wildcard, big_endian, block, cap):
        C = self.C
        kw = np.array([ord(c) for c in keyword], np.uint32)
        out = np.zeros(cap, np.uint64)
        n = self.lib.mmoore_c_engine_run(elem, path.encode(), kw.ctypes.data_as(C.POINTER(C.c_uint32)), len(kw), wildcard, None, 0, None, 0,
                                         int(big_endian), block, 50, 0, out.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
        if n < 0:
            raise RuntimeError("SearchEngine<T>::run on the GPU facade failed: " + self.lib.mmoore_c_last_error().decode())
        return out[:min(n, cap)].copy()

    def maps(self, n, pairs=2):
        C = self.C
        sym, val = np.zeros(n * pairs, np.uint32), np.zeros(n * pairs, np.uint32)
        got = self.lib.mmoore_c_result_maps(sym.ctypes.data_as(C.POINTER(C.c_uint32)), val.ctypes.data_as(C.POINTER(C.c_uint32)), n, pairs)
        return got, sym, val


def end_to_end(mm, ref, path, nbytes, cfg, ref_offs, warmups, runs):
    """The apples-to-apples figure ("also report H2D-inclusive end-to-end separately"): the facade's
    SearchEngine<T>::run on the very tmpfs file the CPU baseline was timed on -- file -> parallel readers -> pinned
    staging -> PCIe -> HBM -> scan -> offsets + equivalency maps -- against the reference's run of the same call."""
    # one GPU, the rank's own: `--gpus N` is what the line is about (left to itself the facade deals a file of 2 GiB and more
    # over all VISIBLE GPUs -- on a multi-GPU node that would be another, faster experiment than "N = 1")
    os.environ["MMOORE_HIP_DEVICES"] = "1"
    fac = Facade(mm)
    elem, kw, wc, be = cfg["elem"], cfg["keyword"], cfg["wildcard"] or ord("*"), cfg["be"]
    cap = len(ref_offs) + 64
    times, offs = _timed(lambda: fac.engine(elem, path, kw, wc, be, BLOCK, cap), warmups, runs)
    same_offsets = bool(np.array_equal(offs, np.asarray(ref_offs, dtype=np.uint64)))
    # equivalency maps of every match, both sides (ASCII keyword: the 'A' and 'a' bases)
    got, sym, val = fac.maps(len(offs))
    ref.engine(elem, None, kw, wc, None, big_endian=be, threads=os.cpu_count() or 1, block_size=BLOCK, path=path)
    same_maps = got == len(offs)
    for i in range(len(offs)) if same_maps else ():
        if sorted(ref.result_map(i).items()) != sorted(zip(sym[2 * i: 2 * i + 2].tolist(), val[2 * i: 2 * i + 2].tolist())):
            same_maps = False
            break
    med, best = float(np.median(times)), min(times)
    assert same_offsets and same_maps, "the facade's file search differs from the reference's (offsets %s, maps %s)" % (same_offsets, same_maps)
    return dict(value=nbytes / med / 1e9, unit="GB/s", ms_per_run=med * 1e3, best_ms=best * 1e3, best_GBps=nbytes / best / 1e9,
                runs=runs, warmups=warmups, bytes=nbytes, matches=int(len(offs)),
                same_offsets_as_reference=same_offsets, same_values_maps_as_reference=same_maps,
                what="SearchEngine<uint%d_t>::run of libmonkey-core.so (the include/mmoore facade over the C ABI) on the tmpfs file of "
                     "cpu_baseline, ONE GPU (MMOORE_HIP_DEVICES=1): ingest over PCIe + scan + equivalency maps, no previews; median of the timed runs; NOT `value` "
                     "(which scans a ROM resident in HBM)" % (8 * elem))


def cpu_baseline(mm, eng, shard_bytes, cfg, want_bytes, warmups, runs, with_end_to_end=True):
    """the reference CPU engine on the bench ROM written to tmpfs,
    hardware_concurrency threads, 512 KiB blocks, >= 3 warm-ups, >= 10 timed runs, median + min.
    Returns (json object, offsets, bytes covered, end_to_end object or None)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import Oracle, Ref
    cores = os.cpu_count() or 1
    plan_kw, elem, wc, be = cfg["keyword"], cfg["elem"], cfg["wildcard"] or 0, cfg["be"]
    if not Ref.available():
        # no compiled reference on this box: the scalar C restatement on a bounded sample
        n = min(shard_bytes, 256 << 20)
        rom = eng.download(0, n)
        orc = Oracle()
        t0 = time.perf_counter()
        offs = orc.engine(orc.plan(elem, plan_kw, wc), rom, BLOCK, be)
        dt = time.perf_counter() - t0
        return dict(value=n / dt / 1e9, unit="GB/s", cores=1, kind="port",
                    sample="first %d MiB of the bench ROM, scalar C restatement, 1 run" % (n >> 20)), offs, n, None
    ref = Ref()
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    # the whole 4 GiB (the largest file the shipped engine handles, search_engine.cpp:241-242) when
    # it fits next to everything else in tmpfs, else the largest power-of-two fraction that does
    n = min(want_bytes, shard_bytes)
    free = shutil.disk_usage(tmpdir).free
    shrunk = False
    while n > (64 << 20) and n + (1 << 30) > free:
        n //= 2
        shrunk = True
    n = n // BLOCK * BLOCK
    path = os.path.join(tmpdir, "mm_cpu_baseline_%d.bin" % os.getpid())
    try:
        with open(path, "wb") as f:
            piece = 512 << 20
            for at in range(0, n, piece):                     # 512 MiB at a time: no second copy of the ROM in RAM
                f.write(memoryview(eng.download(at, min(piece, n - at))))
        times, offs = _timed(lambda: ref.engine(elem, None, plan_kw, wc or ord("*"), None, big_endian=be, threads=cores,
                                                block_size=BLOCK, path=path), warmups, runs)
        e2e = None
        if with_end_to_end:
            try:
                e2e = end_to_end(mm, ref, path, n, cfg, offs, warmups, runs)
            except AssertionError:
                raise
            except Exception as e:                            # noqa: BLE001 -- reported, never hidden
                e2e = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
    finally:
        if os.path.exists(path):
            os.unlink(path)
    # per-core figures in the shape of the reference's own benchmark:
    # MonkeyMoore<uint8_t>::search, one thread, 16 MiB of its mt19937(42) buffer
    c1 = ref.bench_data(1, 16 << 20)
    single = {}
    for kw in ("abcde", "monkey"):
        t1, _ = _timed(lambda: ref.search(1, kw, c1), 3, 10)
        single[kw] = dict(median_GBps=(16 << 20) / float(np.median(t1)) / 1e9, best_GBps=(16 << 20) / min(t1) / 1e9)
    med, best = float(np.median(times)), min(times)
    return dict(value=n / med / 1e9, unit="GB/s", cores=cores, kind="reference",
                sample="%s%d MiB of the bench ROM in a tmpfs file, SearchEngine<T>::run built from the reference sources, "
                       "%d threads, 512 KiB blocks, no previews; %d warm-ups, %d timed runs, value = median"
                       % ("(tmpfs too small for 4 GiB) " if shrunk else "", n >> 20, cores, warmups, runs),
                median_GBps=n / med / 1e9, best_GBps=n / best / 1e9, runs=runs, warmups=warmups, bytes=n,
                single_thread=single,
                single_thread_sample="MonkeyMoore<uint8_t>::search, 1 thread, 16 MiB mt19937(42) buffer "
                                     "(benchmarks/bench_search.cpp shape), 3 warm-ups, 10 runs"), offs, n, e2e


def measure_pmc_traffic(kernel, config="C2"):
    """HBM bytes per launch of the dominant kernel and its instruction mix, COUNTED in this run: rocprofv3 --pmc FETCH_SIZE,
    --pmc WRITE_SIZE and --pmc SQ_INSTS_VALU SQ_INSTS_LDS (three passes, nothing else traced) around short child runs of this
    script on the same configuration -- started before this process has touched the GPU, the program itself directly behind
    `--`.  FETCH_SIZE is doubled per the gfx950 rule of MI355X_MICROARCH.md (it reports half of a wide coalesced streaming
    read), both byte counters are KiB; the SQ counters count wave instructions (x 64 lanes).
    Returns (bytes per launch or None, how it was obtained / why not, {counter: mean per launch})."""
    import csv
    import glob
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found: not measured in this run", {}
    child = [sys.executable, os.path.abspath(__file__), "--config", config, "--steps", "3", "--warmup", "1", "--depth", "1", "--no-cpu-baseline",
             "--no-other-depth", "--no-other-configs", "--no-strong", "--no-read-probe", "--no-pmc", "--no-split", "--prewarm-s", "0.05"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["TMPDIR"] = "/tmp"
    got = {}
    for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_INSTS_LDS")):
        d = tempfile.mkdtemp(prefix="mm_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=120)
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    rows += [row for row in csv.DictReader(fh) if row["Kernel_Name"].startswith("void " + kernel + "(")]
            for counter in counters:
                vals = [float(row["Counter_Value"]) for row in rows if row.get("Counter_Name") == counter]
                # (the library's device warm-up at mmh_create launches the same kernel on an 8 MiB ROM: only the launches
                # over the bench ROM count -- everything within half of the largest)
                vals = [v for v in vals if v >= 0.5 * max(vals)] if vals else vals
                if r.returncode != 0 or not vals:
                    if counter.startswith("SQ_"):
                        continue                              # (the instruction mix is an extra: the traffic figure stands without it)
                    return None, "rocprofv3 --pmc %s: rc %d, %d launches of %s counted (%s): not measured in this run" % (
                        counter, r.returncode, len(vals), kernel, (r.stderr or "").strip().splitlines()[-1][:120] if r.stderr else ""), {}
                got[counter] = (sum(vals) / len(vals), len(vals))
        except Exception as e:                                # noqa: BLE001 -- a box without counters must not cost the line
            if counters[0].startswith("SQ_"):
                continue
            return None, "rocprofv3 --pmc %s failed (%s: %s): not measured in this run" % (counters[0], type(e).__name__, e), {}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = (got["FETCH_SIZE"][0] * 1024.0, got["FETCH_SIZE"][1]), (got["WRITE_SIZE"][0] * 1024.0, got["WRITE_SIZE"][1])
    return 2 * fetch[0] + write[0], (
        "counted in this run: rocprofv3 --pmc FETCH_SIZE (%d launches of %s, %.0f bytes each, doubled per the gfx950 rule) and "
        "--pmc WRITE_SIZE (%d launches, %.0f bytes each) around child runs of this script (--config %s --steps 3 --depth 1, the same "
        "ROM and keyword, --no-split: every launch covers the whole ROM), started before this process touched the GPU" % (
            fetch[1], kernel, fetch[0], write[1], write[0], config)), {k: v[0] for k, v in got.items()}


def pmc_traffic(mm, shard):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes -- only when they were taken
    with THIS device code (the summary carries the hash of the library's sources), else null."""
    for name in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            pmc = json.load(f)
        src = mm.build.device_source_sha16()
        with open(mm.LIB_PATH, "rb") as f:
            lib = hashlib.sha256(f.read()).hexdigest()[:16]
        # the same device code: the same sources (a rebuild elsewhere gives another binary), or the very binary
        # (a comment edit gives other sources)
        if pmc.get("device_source_sha16") != src and pmc.get("library_sha16") != lib:
            return None, "profiles/%s was taken with other device code (sources %s / library %s, here %s / %s): not reported" % (
                name, pmc.get("device_source_sha16"), pmc.get("library_sha16"), src, lib)
        return (pmc["hbm_traffic_bytes_per_launch"] * shard / pmc["algorithmic_bytes_per_launch"],
                "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH doubled per the gfx950 rule; "
                "same device code: sources sha256 %s, library %s)" % (name, src, lib))
    return None, None


def condition_device(eng, scan, settle=0.005, most_s=1.5, least_s=0.1):
    """Scans until the streaming kernel's duration (HIP events) has settled: three consecutive scans within `settle` of
    each other, at least least_s and at most most_s of load.  Returns what it took."""
    t0 = time.perf_counter()
    last, scans = [], 0
    while True:
        scan()
        scans += 1
        last = (last + [eng.timings()["filter_ms"]])[-3:]
        dt = time.perf_counter() - t0
        settled = len(last) == 3 and min(last) > 0 and (max(last) - min(last)) / min(last) <= settle
        if (settled and dt >= least_s) or dt >= most_s:
            return {"scans": scans, "seconds": round(dt, 3), "settled": bool(settled), "last_kernel_ms": [round(x, 4) for x in last]}


def other_config(mm, torch, dev, name, scans, steps):
    """One of BASELINE.json's other single-GPU configurations behind the timed region: `scans` synchronous scans
    (mmh_scan) for the kernel's own duration and the caller's latency, `steps` steps with three scans in flight
    for the throughput, and parity of the offsets with the oracle on the blocks inside a 256 MiB prefix."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import Oracle, oracle_engine_parallel
    cfg = CONFIGS[name]
    kw, elem, wc, be = cfg["keyword"], cfg["elem"], cfg["wildcard"], cfg["be"]
    nbytes = int(cfg["gib"] * (1 << 30))
    buf = torch.empty(nbytes + 32, dtype=torch.uint8, device=dev)
    eng = mm.Engine(dev.index or 0)
    try:
        eng.attach(buf.data_ptr(), nbytes)
        mm.synth.RomSpec(SEED, nbytes, kw, elem, wc, be, BLOCK).apply_device(eng)
        plan = mm.plan_relative(elem, kw, wc or 0)
        # untimed: clocks back up after the ROM's set-up (an 8 GiB allocation + fill leaves the device idle-clocked for
        # longer than a fixed 0.2 s covers on some boxes: round 3's driver line had C3's kernel 5 % above its profile),
        # the lanes' streams and workspaces created.  Condition until the streaming kernel's own duration has settled:
        # three consecutive scans within 0.5 % of each other (at most 1.5 s).
        # (the lanes first: creating their streams and workspaces -- allocations, pinned memory, memsets -- leaves the device
        # idle for milliseconds, and a conditioning in FRONT of that had the synchronous scans behind it start on dropped
        # clocks again: round 4's lines had C3's kernel at 0.710 ms in the conditioning and 0.742 in the scans that counted)
        for t in [eng.submit(plan, block_bytes=BLOCK, big_endian=be) for _ in range(3)]:
            eng.collect(t)
        # the kernel with the device to itself: ONE launch over the whole ROM per scan (MMH_ROUTE_NO_SPLIT)
        eng.set_route(mm.ROUTE_NO_SPLIT)
        conditioning = condition_device(eng, lambda: eng.scan(plan, block_bytes=BLOCK, big_endian=be))
        for _ in range(scans):
            offs = eng.scan(plan, block_bytes=BLOCK, big_endian=be)
        filt, tot = eng.timing_history(min(scans, 64))
        ctr = eng.counters()
        eng.set_route(0)
        # what a caller of the synchronous API gets (a pipeline of parts on ROMs of this size)
        for _ in range(4):
            offs = eng.scan(plan, block_bytes=BLOCK, big_endian=be)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(scans):
            offs = eng.scan(plan, block_bytes=BLOCK, big_endian=be)
        sync_s = time.perf_counter() - t0
        sync_parts = eng.timings().get("parts", 0)
        tickets, last = [], None
        t0 = time.perf_counter()
        for _ in range(steps):
            tickets.append(eng.submit(plan, block_bytes=BLOCK, big_endian=be))
            if len(tickets) == 3:
                last = eng.collect(tickets.pop(0))
        while tickets:
            last = eng.collect(tickets.pop(0))
        flight_s = time.perf_counter() - t0
        assert np.array_equal(last, offs), name + ": scans in flight delivered another list"
        n = 256 << 20
        rom = eng.download(0, n)
        orc = Oracle()
        want = oracle_engine_parallel(orc, orc.plan(elem, kw, wc or 0), rom, BLOCK, be)
        lim = (n // BLOCK - 1) * BLOCK
        g, c = offs[offs < lim].tolist(), [int(x) for x in want if x < lim]
        assert g == c, name + ": GPU offsets differ from the oracle on the first %d MiB" % (lim >> 20)
        k = float(np.median(filt))
        return {
            "workload": "%s: %s, engine semantics, 512 KiB blocks, %.1f GiB splitmix64 ROM resident in HBM" % (name, cfg["what"], cfg["gib"]),
            "kernel": "mm_filter_u%d<%d>" % (8 * elem, mm.filter_shape(plan)["shape"]),
            "kernel_ms": k, "kernel_ms_mean": float(np.mean(filt)), "kernel_ms_min": float(np.min(filt)),
            "kernel_ms_is": "median over %d synchronous scans of ONE launch each (MMH_ROUTE_NO_SPLIT)" % scans, "conditioning": conditioning,
            "scan_device_ms": float(np.median(tot)),
            "achieved_GBps": nbytes / (k * 1e-3) / 1e9, "frac": nbytes / (k * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "synchronous": {"scans": scans, "ms_per_scan": sync_s / scans * 1e3, "GBps": nbytes * scans / sync_s / 1e9, "parts": sync_parts},
            "in_flight": {"steps": steps, "ms_per_step": flight_s / steps * 1e3, "GBps": nbytes * steps / flight_s / 1e9},
            "matches": int(len(offs)), "candidates": ctr["candidates"], "path": ctr["path"],
            "parity": "first %d MiB: %d offsets identical to the oracle (C restatement, all host cores)" % (lim >> 20, len(g)),
        }
    finally:
        eng.close()
        del buf


def c1_config(mm, scans):
    """BASELINE.json configs[0] (C1) on the GPU: the reference benchmark's own input -- bench_search.cpp:11-22, 16 MiB of
    mt19937(42) bytes -- and its keyword `abcde` plus BASELINE's 6-character `monkey`, searched the way the benchmark does:
    ONE chain over the whole buffer (MonkeyMoore<uint8_t>::search, `block_bytes` 0).  Two ways in: the C ABI with the buffer
    resident in HBM (what `value` measures for C2), and the include/mmoore facade's MonkeyMoore<uint8_t>::search on the
    host buffer (the call the reference's benchmark times: the 16 MiB cross PCIe inside it).  Offsets against the compiled
    reference where oracle/_ref travelled, else against the oracle's C restatement."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import Oracle, Ref
    n = 16 << 20
    data = mm.synth.bench_search_buffer(1, n)
    ref = Ref() if Ref.available() else None
    if ref is not None:
        assert np.array_equal(ref.bench_data(1, n), data), "C1: the mt19937(42) buffer differs from the reference build's"
    orc = Oracle()
    fac = Facade(mm)
    u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    fac.lib.mmoore_c_search.restype = C.c_int64
    fac.lib.mmoore_c_search.argtypes = [C.c_int, u32p, C.c_int, C.c_uint32, u32p, C.c_int, C.c_void_p, C.c_uint64, u64p, C.c_uint64]
    eng = mm.Engine(0)
    out = {"workload": "C1: 8-bit relative search, whole-buffer chain (MonkeyMoore<uint8_t>::search), 16 MiB mt19937(42) buffer of "
                       "benchmarks/bench_search.cpp", "bytes": n, "keywords": {}}
    try:
        eng.upload(data)
        for kw in ("abcde", "monkey"):
            plan = mm.plan_relative(1, kw, 0)
            want = (ref.search(1, kw, data, 0) if ref is not None else orc.search(orc.plan(1, kw, 0), data)).tolist()
            for _ in range(3):
                got = eng.scan(plan)
            assert got.tolist() == want, "C1 %s: the GPU's offsets differ from the reference's" % kw
            t0 = time.perf_counter()
            for _ in range(scans):
                got = eng.scan(plan)
            wall = (time.perf_counter() - t0) / scans
            filt, tot = eng.timing_history(min(scans, 64))
            ctr = eng.counters()
            # the facade: MonkeyMoore<uint8_t>(keyword).search(host pointer, elements)
            kwa = np.array([ord(c) for c in kw], np.uint32)
            res = np.zeros(max(64, len(want) + 8), np.uint64)
            call = lambda: fac.lib.mmoore_c_search(1, kwa.ctypes.data_as(u32p), len(kwa), 0, None, 0, data.ctypes.data, n,       # noqa: E731
                                                   res.ctypes.data_as(u64p), res.size)
            ft, found = _timed(call, 3, scans)
            assert found == len(want) and res[:found].tolist() == want, "C1 %s: the facade's offsets differ from the reference's" % kw
            k = float(np.median(filt))
            out["keywords"][kw] = {
                "kernel": "mm_filter_u8<%d>" % mm.filter_shape(plan)["shape"], "kernel_ms": k, "scan_device_ms": float(np.median(tot)),
                "achieved_GBps": n / (k * 1e-3) / 1e9, "frac": n / (k * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "synchronous": {"scans": scans, "ms_per_scan": wall * 1e3, "GBps": n / wall / 1e9},
                "facade_search": {"ms_per_call": float(np.median(ft)) * 1e3, "GBps": n / float(np.median(ft)) / 1e9,
                                  "what": "MonkeyMoore<uint8_t>::search of libmonkey-core.so on the HOST buffer: 16 MiB over PCIe + scan, per call"},
                "matches": len(want), "candidates": ctr["candidates"], "path": ctr["path"],
                "parity": "%d offsets identical to %s" % (len(want), "the compiled reference's MonkeyMoore<uint8_t>::search (oracle/_ref)"
                                                          if ref is not None else "the oracle's (C restatement)"),
            }
        out["frac_is"] = ("16 MiB streams in ~7 us at the 4 GiB rate: a scan of this size is launch- and latency-bound (kernel_ms is the "
                          "streaming kernel alone, scan_device_ms adds the tail kernel, synchronous the caller's wait), not a bandwidth figure")
        return out
    finally:
        eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2",
                    help="C2: 4 GiB per GPU, 12-char 8-bit keyword (the headline metric); C3: 16 chars with 3 wildcards; C4 / C4BE: 8 GiB "
                         "16-bit LE / BE; C5: 8 GiB per GPU (64 GiB over 8 GPUs)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the C3 / C4 / C4BE measurements behind the timed region of the default run (`other_configs`)")
    ap.add_argument("--other-scans", type=int, default=10, help="synchronous scans per configuration in `other_configs`")
    ap.add_argument("--dry-launch", action="store_true",
                    help="tests: every rank prints its RANK / WORLD_SIZE as one JSON line and exits before anything touches a GPU")
    ap.add_argument("--gib-per-gpu", type=float, default=None, help="overrides the config's size")
    ap.add_argument("--cpu-sample-mib", type=int, default=4096)
    ap.add_argument("--cpu-runs", type=int, default=10)
    ap.add_argument("--cpu-warmups", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm-s", type=float, default=0.3, help="device clock conditioning before the warm-up steps")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N > 1: finish every step's offset gather before the next scan starts (no overlap)")
    ap.add_argument("--torch-gather", action="store_true",
                    help="N > 1: gather through torch.distributed (the test double) instead of the library's own RCCL calls")
    ap.add_argument("--depth", type=int, choices=(1, 2, 3), default=3,
                    help="tickets outstanding in the timed region: 2 or 3 = mmh_scan_submit / mmh_scan_collect, the next scan's "
                         "streaming kernel runs while the previous scan's tail kernel, result hand-over and gather finish (two "
                         "scans are at work on the device either way; with 3 the host has the next one enqueued ahead and may be "
                         "late); 1 = mmh_scan, every step waits for its own result")
    ap.add_argument("--host-delay-us", type=float, default=0.0,
                    help="development probe: the host idles this long after every step's result (a host that is late); "
                         "never used for a reported figure")
    ap.add_argument("--force-gather", action="store_true",
                    help="N = 1 only (tests): take the N > 1 path anyway -- process group, communicator and gather of ONE rank")
    ap.add_argument("--allow-shared-device", action="store_true",
                    help="TESTS ONLY: the N ranks share GPU 0 through the RCCL stand-in of tests/shim (LD_PRELOAD; RCCL itself refuses "
                         "ranks that share a device).  The line says so (`shared_device`) and its value is NOT a multi-GPU figure; "
                         "without this flag a process the stand-in is loaded into refuses to run")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): N x the config's ROM, one partition per GPU -- `value`; the line also carries `strong`: "
                         "ONE ROM of the config's size dealt over the N GPUs.  strong: that figure becomes `value`")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling leg behind the timed region")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the facade's SearchEngine<T>::run on the CPU baseline's tmpfs file (`end_to_end`)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not count the dominant kernel's HBM traffic in this run (two rocprofv3 --pmc child runs of this script "
                         "before the GPU is touched: ~12 s); `roofline.traffic` then comes from profiles/ when the device code matches")
    ap.add_argument("--no-read-probe", action="store_true", help="skip the pure-read probe (`roofline.measured_read_ceiling_GBps`)")
    ap.add_argument("--no-split", action="store_true",
                    help="every synchronous scan is ONE streaming launch over the whole ROM (MMH_ROUTE_NO_SPLIT) instead of the pipeline of "
                         "parts mmh_scan runs on ROMs of >= 1 GiB: for rocprofv3 kernel statistics of the whole-ROM launch")
    ap.add_argument("--no-other-depth", action="store_true",
                    help="skip the extra K steps at the other depth after the timed region (the 'synchronous' / 'in_flight' object)")
    args = ap.parse_args()

    in_launcher = "RANK" in os.environ or "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_launcher:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], args.dry_launch, args.allow_shared_device))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to print a line whose n_gpus is not what was asked for\n"
                         % (args.gpus, world))
        raise SystemExit(2)
    if args.dry_launch:
        # (host only: the partition rule of mmh_partition for both scaling modes, no GPU touched)
        from __graft_entry__ import load_package
        mm = load_package()
        cfg = CONFIGS[args.config]
        gib = args.gib_per_gpu if args.gib_per_gpu is not None else cfg["gib"]
        per_gpu = int(gib * (1 << 30)) // BLOCK * BLOCK
        L, ELEM = len(cfg["keyword"]), cfg["elem"]
        wb, ws = mm.partition_range(per_gpu * world, BLOCK, L, ELEM, rank, world)
        sb, ss = mm.partition_range(per_gpu, BLOCK, L, ELEM, rank, world)
        print(json.dumps({"dry_launch": True, "rank": rank, "local_rank": local_rank, "world": world, "gpus": args.gpus,
                          "scaling": args.scaling, "weak": {"total": per_gpu * world, "base": wb, "bytes": ws},
                          "strong": {"total": per_gpu, "base": sb, "bytes": ss, "overlap": (L - 1) * ELEM}}), flush=True)
        return

    # `roofline.traffic` counted in the run itself: the driver's own command line only (N = 1, C2 at full size, nothing
    # switched off), and before anything here touches the GPU
    live_traffic = (None, None, {})
    if (world == 1 and args.gib_per_gpu is None and not args.force_gather and not args.no_pmc
            and (args.config != "C2" or (not args.no_cpu_baseline and not args.no_other_configs))):
        # the kernel the timed configuration's keyword selects (host only -- and asked of a child: this process loads
        # nothing of HIP before the counting children have come and gone)
        import subprocess
        _cfg = CONFIGS[args.config]
        _probe = subprocess.run([sys.executable, "-c",
                                 "import sys; sys.path.insert(0, %r)\nfrom __graft_entry__ import load_package\nmm = load_package()\n"
                                 "print(mm.filter_shape(mm.plan_relative(%d, %r, %d))['shape'])" % (ROOT, _cfg["elem"], _cfg["keyword"], _cfg["wildcard"] or 0)],
                                capture_output=True, text=True, timeout=300)
        if _probe.returncode == 0 and _probe.stdout.strip().isdigit():
            live_traffic = measure_pmc_traffic("mm_filter_u%d<%s>" % (8 * _cfg["elem"], _probe.stdout.strip()), args.config)
        else:
            live_traffic = (None, "the filter shape of %s could not be asked of the library: %s" % (args.config, (_probe.stderr or "").strip()[-200:]), {})

    # (the pool's host driver only supports dmabuf IPC: RCCL across processes needs this; exported on the boxes already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # HIP streams beyond the 4th share hardware queues by default and then serialize: at N > 1 the process has
    # torch's stream(s), torch's NCCL stream, the library's two lane streams and its gather stream
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    if world > 1 or args.force_gather or args.torch_gather:
        # the gather's CHECKER (torch.distributed's all_gather of the same lists) lives with the tests: found now, before
        # anything is timed, or the run stops here with a plain message -- not with an ImportError behind the timed region
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        try:
            import _gather_double                                # noqa: F401
        except ImportError:
            raise SystemExit("bench.py: tests/_gather_double.py (the checker of the multi-rank offset gather) is not next to bench.py: "
                             "run it from a checkout of the repository")

    # The RCCL stand-in of tests/shim only ever serves a run that asks for it: a figure measured through it is not RCCL's.
    shared = bool(args.allow_shared_device)
    if rccl_stand_in_loaded() != shared:
        raise SystemExit("bench.py: %s" % ("the test-only RCCL stand-in (tests/shim/libfake_rccl.so) is loaded into this process "
                                           "without --allow-shared-device: refusing to time it as RCCL" if not shared else
                                           "--allow-shared-device needs LD_PRELOAD=tests/shim/libfake_rccl.so (RCCL refuses ranks that share a device)"))
    if shared:
        local_rank = 0                                       # every rank on GPU 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d needs GPU %d, %d visible: the engine has no CPU fallback" % (rank, local_rank,
                                                                                                         torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if shared else dev            # where the rendezvous' own tensors live (gloo when ranks share a device)
    multi = world > 1 or args.force_gather
    banner = c_stdout_to_stderr() if multi else None     # (RCCL's version banner: until both communicators are up)
    if multi:
        banner.__enter__()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from __graft_entry__ import load_package
    mm = load_package()
    if not os.path.exists(mm.LIB_PATH):
        if local_rank == 0:                                  # one builder per node, the others wait
            mm.build.build_all()
        if multi:
            dist.barrier()

    cfg = CONFIGS[args.config]
    KEYWORD, ELEM, WC, BE = cfg["keyword"], cfg["elem"], cfg["wildcard"], cfg["be"]
    gib = args.gib_per_gpu if args.gib_per_gpu is not None else cfg["gib"]
    per_gpu = int(gib * (1 << 30)) // BLOCK * BLOCK
    total = per_gpu * world
    L = len(KEYWORD)
    # block-aligned partition + pattern-length overlap into the next one
    base, shard = mm.partition_range(total, BLOCK, L, ELEM, rank, world)

    # HBM-resident shard owned by torch; the engine borrows the pointer and runs on torch's stream
    buf = torch.empty(shard + 32, dtype=torch.uint8, device=dev)
    eng = mm.Engine(local_rank)
    eng.attach(buf.data_ptr(), shard)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.no_split:
        eng.set_route(mm.ROUTE_NO_SPLIT)
    spec = mm.synth.RomSpec(SEED, total, KEYWORD, ELEM, WC, BE, BLOCK, base=base, nbytes=shard, partitions=8)
    spec.apply_device(eng)
    torch.cuda.synchronize()
    plan = mm.plan_relative(ELEM, KEYWORD, WC or 0)
    kernel_name = "mm_filter_u%d<%d>" % (8 * ELEM, mm.filter_shape(plan)["shape"])

    # N > 1: the gather of the per-GPU offset lists (already ascending, partitions in rank order).
    # Product path: the library's own communicator -- rank 0 makes the RCCL id, torch.distributed
    # (the launcher's rendezvous) only distributes its 128 bytes; the collective is issued by
    # libmmoore_hip.so from the device-side copy of the scan's list.  It is started right after a
    # scan and finished after the NEXT scan: the collective overlaps that scan (one gather in
    # flight; --sync-gather turns the overlap off).  Every step still delivers one merged list.
    gather_backend, gather_note = None, None
    gatherer = None
    if multi:
        if not args.torch_gather:
            try:
                box = [mm.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                eng.comm_init_rank(box[0], world, rank)
                gather_backend = "librccl via the C ABI (mmh_gather_start / mmh_gather_finish), lists sent from HBM"
            except Exception as e:                            # noqa: BLE001 -- a box the builder could not test on
                gather_note = "NATIVE RCCL COMMUNICATOR FAILED on rank %d (%s: %s)" % (rank, type(e).__name__, e)
                sys.stderr.write("rank %d: %s\n" % (rank, gather_note))
        # all ranks must take the same path
        ok = torch.tensor([1 if (gather_backend or args.torch_gather) else 0], device=cdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        all_up = int(ok.item())
        banner.__exit__()                                     # both communicators have made their first calls
        if all_up == 0:
            # The product path is the library's own collective.  Timing the torch.distributed double in its place would
            # print a throughput the product did not deliver: say so in a line without a value and fail the run.
            if rank == 0:
                print(json.dumps({"metric": "GB/s scanned (%g GiB synthetic ROM per GPU)" % gib, "value": None, "unit": "GB/s",
                                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                                  "scaling": "weak", "config": {"workload": cfg["what"], "name": args.config},
                                  "error": "the library's RCCL communicator did not come up on every rank; nothing was timed "
                                           "(--torch-gather runs the test double instead, for diagnosis only)",
                                  "gather_note": gather_note}), flush=True)
            dist.barrier()
            dist.destroy_process_group()
            raise SystemExit(3)
        if args.torch_gather:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import _gather_double                             # (the torch.distributed double: diagnosis only, never a reported figure)
            gatherer = _gather_double.OffsetGather(rank, world, cdev, dist)
    native = multi and gather_backend is not None
    if multi and not native:
        gather_backend = "torch.distributed all_gather (test double, --torch-gather: NOT the product path)"
    in_flight = []
    gather_dev_ms, gather_host_ms = [], []

    def gather_start(offs):
        if native:
            eng.gather_start(None, want_list=(rank == 0))
            return True
        return gatherer.start(offs, async_op=not args.sync_gather)

    def gather_finish(h):
        if native:
            merged = eng.gather_finish(want_list=(rank == 0))
            t = eng.gather_timings()
            gather_dev_ms.append(t["device_ms"])
            gather_host_ms.append(t["host_ms"])
            return merged if rank == 0 else None
        t0 = time.perf_counter()
        merged = gatherer.finish(h)
        gather_host_ms.append((time.perf_counter() - t0) * 1e3)
        return merged

    def deliver(offs):
        """what a step hands over: the scan's list at N = 1, else the gathered one (of this or the previous step)"""
        if not multi:
            return offs
        if args.sync_gather:
            merged = gather_finish(gather_start(offs))
            return merged if rank == 0 else offs
        in_flight.append(gather_start(offs))
        merged = gather_finish(in_flight.pop(0)) if len(in_flight) == 2 else None
        return merged if (rank == 0 and merged is not None) else offs

    def drain(last):
        merged = None
        while in_flight:
            merged = gather_finish(in_flight.pop(0))
        return merged if (rank == 0 and merged is not None) else last

    def late_host():
        if args.host_delay_us > 0:
            t_end = time.perf_counter() + args.host_delay_us * 1e-6
            while time.perf_counter() < t_end:
                pass

    def run_steps(k, depth, at=None, gather=True, per_step=None):
        """k steps = k scans of the shard + k deliveries, nothing left in flight at the end.  at: the shard's base offset
        (default: this rank's weak-scaling partition); gather=False: no collective (a rank measuring on its own);
        per_step: a list that receives every step's own wall time in ms (depth 1 only: a step is one call there)"""
        at = base if at is None else at
        hand = deliver if gather else (lambda offs: offs)
        last, tickets = None, []
        for _ in range(k):
            late_host()
            if depth == 1:
                t_step = time.perf_counter()
                last = hand(eng.scan(plan, block_bytes=BLOCK, big_endian=BE, base_offset=at))
                if per_step is not None:
                    per_step.append((time.perf_counter() - t_step) * 1e3)
            else:
                tickets.append(eng.submit(plan, block_bytes=BLOCK, big_endian=BE, base_offset=at))
                if len(tickets) == depth:
                    last = hand(eng.collect(tickets.pop(0)))
        while tickets:
            last = hand(eng.collect(tickets.pop(0)))
        return drain(last) if gather else last

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # Device conditioning (not a step of the workload): after the idle setup phase the MI355X
    # needs a few tens of ms of sustained load before its memory/fabric clocks are back up --
    # the first ~20 scans run ~12 % slower than steady state.  Scan until PREWARM_S have
    # passed, then do the W warm-up steps and the K timed steps of the contract.
    launches_by_phase = []                                  # (phase, launches of the streaming kernel): tools/summarize_profiles.py cuts the kernel trace by it

    def phase(name, _seen=[0]):
        v = eng.health()["validated"]                        # one validated block per polled launch pair (streaming + tail kernel)
        launches_by_phase.append((name, v - _seen[0]))
        _seen[0] = v
    phase("set-up")
    t_pre = time.perf_counter()
    prewarm_scans = 0
    while time.perf_counter() - t_pre < args.prewarm_s:
        eng.scan(plan, block_bytes=BLOCK, big_endian=BE, base_offset=base)
        prewarm_scans += 1
    if multi:
        # ... and the collective's own first calls (RCCL sets up its channels and buffers on the first all-gathers:
        # measured with one rank, the first ~10 gathered steps cost 0.1 ms more each than the steady state)
        run_steps(max(12, args.warmup), args.depth)
        prewarm_scans += max(12, args.warmup)
    phase("pre-warm")
    run_steps(args.warmup, args.depth)
    fence()
    phase("warm-up, %d ticket(s) outstanding" % args.depth)
    del gather_dev_ms[:], gather_host_ms[:]
    t0 = time.perf_counter()
    offs = run_steps(args.steps, args.depth)             # the last collect / gather belongs to the timed region
    fence()
    elapsed = time.perf_counter() - t0
    phase("TIMED steps, %d ticket(s) outstanding" % args.depth)
    # HIP-event timings of the timed steps: recorded on the scan's stream during the steps,
    # read back afterwards (the library keeps the timings of the last 64 scans)
    filt_ms, tot_ms = eng.timing_history(min(args.steps, 64))
    post_ms = tot_ms - filt_ms
    gather_dev, gather_host = list(gather_dev_ms), list(gather_host_ms)

    # Extra, NOT part of `value`: the same K steps at the other depth.
    other_depth = 1 if args.depth > 1 else 3
    offs_other, elapsed_other = None, 0.0
    if not args.no_other_depth:
        run_steps(max(args.warmup, 4), other_depth)
        fence()
        t1 = time.perf_counter()
        phase("warm-up of the other leg")
        other_per_step = []
        offs_other = run_steps(args.steps, other_depth, per_step=other_per_step)
        fence()
        elapsed_other = time.perf_counter() - t1
        filt_other, tot_other = eng.timing_history(min(args.steps, 64))
        other_parts = eng.timings().get("parts", 0)
        phase("other leg, %d ticket(s) outstanding" % other_depth)
        # ... and, one scan at a time, once more WITHOUT the event at every scan's start (mmh_set_timing(0): what the include/mmoore
        # facade runs with -- the reference API has no timings; the event costs a scan's first dispatch ~4.5 us)
        elapsed_untimed, untimed_per_step = 0.0, []
        if other_depth == 1:
            eng.set_timing(False)
            run_steps(max(args.warmup, 4), 1)
            fence()
            t1 = time.perf_counter()
            offs_untimed = run_steps(args.steps, 1, per_step=untimed_per_step)
            fence()
            elapsed_untimed = time.perf_counter() - t1
            eng.set_timing(True)
            assert np.array_equal(offs_untimed, offs_other), "scans without timing events delivered another list"
            phase("other leg again, no timing events")
    # The dominant kernel with the device to itself, ONE launch over the whole shard (MMH_ROUTE_NO_SPLIT: a synchronous scan
    # of a ROM of >= 1 GiB otherwise runs as a pipeline of parts whose kernels overlap): what `roofline` prices.
    def alone_scans(k, at=None):
        eng.set_route(mm.ROUTE_NO_SPLIT)
        for _ in range(4):
            eng.scan(plan, block_bytes=BLOCK, big_endian=BE, base_offset=base if at is None else at)
        for _ in range(k):
            eng.scan(plan, block_bytes=BLOCK, big_endian=BE, base_offset=base if at is None else at)
        f, t = eng.timing_history(min(k, 64))
        eng.set_route(mm.ROUTE_NO_SPLIT if args.no_split else 0)
        return f, t
    fence()
    kernel_alone_filt, kernel_alone_tot = alone_scans(min(max(args.steps, 16), 64))
    fence()
    phase("the kernel alone: one launch over the whole shard per scan")
    # N > 1, after the timed region: the library's gather against the torch.distributed double on one more
    # scan of every rank -- the first place the native collective meets a real second rank
    gather_check = None
    if native:
        local = eng.scan(plan, block_bytes=BLOCK, big_endian=BE, base_offset=base)
        eng.gather_start(None, want_list=(rank == 0))
        mine = eng.gather_finish(want_list=(rank == 0))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _gather_double                                 # the checker: torch.distributed's gather of the same lists
        theirs = _gather_double.gather_offsets(local, rank, world, cdev, dist)
        if rank == 0:
            same = np.array_equal(np.asarray(mine, dtype=np.uint64), np.asarray(theirs, dtype=np.uint64))
            gather_check = ("%d offsets of %d ranks identical to the torch.distributed gather" % (len(mine), world) if same
                            else "MISMATCH: library %d offsets, torch.distributed gather %d" % (len(mine), len(theirs)))
            if not same:
                sys.stderr.write("bench.py: NATIVE GATHER " + gather_check + "\n")
    if multi:
        tmax = torch.tensor([elapsed, elapsed_other, elapsed_untimed if not args.no_other_depth else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, elapsed_other, elapsed_untimed = float(tmax[0].item()), float(tmax[1].item()), float(tmax[2].item())

    def over_ranks(x):
        """every rank's x, as a list on every rank"""
        if not multi:
            return [float(x)]
        t = torch.zeros(world, dtype=torch.float64, device=cdev)
        t[rank] = float(x)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    # what RCCL saw: the size of the library's own communicator on every rank (0: none -- N = 1 without --force-gather)
    rccl_ranks = over_ranks(eng.comm_info()[1] if native else 0)
    # the streaming kernel's own duration on every rank (launches that overlap nothing where the run has them)
    kernel_ms_ranks = over_ranks(float(np.mean(kernel_alone_filt)))

    # The box's measured read ceiling beside the data-sheet peak: a pure-read kernel over this rank's ROM, <= 50 ms
    read_probe = None
    if not args.no_read_probe:
        try:
            read_probe = eng.read_probe(20)
        except Exception as e:                                # noqa: BLE001
            read_probe = {"error": "%s: %s" % (type(e).__name__, e)}
    probe_ranks = over_ranks(read_probe.get("mean_GBps", 0.0) if read_probe else 0.0)

    # ---- strong scaling: ONE ROM of the config's size dealt over the N GPUs -------------------------------------------
    # (the reference's dispatcher deals ONE file over its workers, search_engine.cpp:66-188 / :218-253; at N = 8 a 4 GiB
    # ROM is 512 MiB per GPU = ~90 us of streaming + tail + gather: the one configuration in which the RCCL gather's
    # latency shows at all)
    strong = None
    if not args.no_strong:
        stotal = per_gpu
        sbase, sshard = mm.partition_range(stotal, BLOCK, L, ELEM, rank, world)
        # N = 1 in the same run: rank 0 scans the whole ROM of that size alone (its weak-scaling shard IS one), the others wait
        n1_ms = 0.0
        fence()
        if rank == 0:
            run_steps(max(args.warmup, 4), args.depth, gather=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(args.steps, args.depth, gather=False)
            torch.cuda.synchronize()
            n1_ms = (time.perf_counter() - t1) / args.steps * 1e3
        fence()
        if world > 1:
            # this rank's partition of the ONE ROM: regenerated in place (the generator is position-determined)
            eng.attach(buf.data_ptr(), sshard)
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
            if sshard:
                mm.synth.RomSpec(SEED, stotal, KEYWORD, ELEM, WC, BE, BLOCK, base=sbase, nbytes=sshard, partitions=8).apply_device(eng)
            torch.cuda.synchronize()
        for _ in range(8):
            eng.scan(plan, block_bytes=BLOCK, big_endian=BE, base_offset=sbase)
        run_steps(max(args.warmup, 4), args.depth, at=sbase)
        fence()
        del gather_dev_ms[:], gather_host_ms[:]
        t1 = time.perf_counter()
        soffs = run_steps(args.steps, args.depth, at=sbase)
        fence()
        selapsed = time.perf_counter() - t1
        sfilt, stot = eng.timing_history(min(args.steps, 64))
        sg_dev, sg_host = list(gather_dev_ms), list(gather_host_ms)
        # one scan at a time as well: the latency a caller of the reference API sees (scan + gather, nothing overlapped)
        fence()
        t1 = time.perf_counter()
        soffs1 = run_steps(args.steps, 1, at=sbase)
        fence()
        selapsed1 = time.perf_counter() - t1
        sfilt1, _ = alone_scans(8, at=sbase) if sshard else (np.zeros(1), None)   # (one launch over this rank's partition, no gather)
        fence()
        if multi:
            tm = torch.tensor([selapsed, selapsed1], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            selapsed, selapsed1 = float(tm[0].item()), float(tm[1].item())
        s_kernel_ranks = over_ranks(float(np.mean(sfilt1)))
        n1_ms = max(over_ranks(n1_ms))
        if rank == 0:
            assert (np.diff(soffs.astype(np.int64)) > 0).all() and np.array_equal(soffs, soffs1), "strong-scaling lists differ"
            sms = selapsed / args.steps * 1e3
            strong = {
                "rom_bytes_total": stotal, "n_gpus": world, "bytes_per_gpu": [int(v) for v in over_ranks(sshard)] if False else None,
                "ms_per_step": sms, "GBps": stotal / (selapsed / args.steps) / 1e9,
                "one_at_a_time": {"ms_per_step": selapsed1 / args.steps * 1e3, "GBps": stotal / (selapsed1 / args.steps) / 1e9},
                "n1_ms_per_step_same_run": n1_ms,
                "efficiency_vs_n1": n1_ms / (world * sms) if sms > 0 else None,
                "kernel_ms_per_rank": {"min": min(s_kernel_ranks), "max": max(s_kernel_ranks)},
                "gather_ms": ({"device_collective_and_pack": float(np.mean(sg_dev)) if sg_dev else None,
                               "host_start_plus_finish": float(np.mean(sg_host)) if sg_host else None} if multi else None),
                "matches": int(len(soffs)), "scans_in_flight": args.depth,
                "what": "ONE %.1f GiB ROM dealt over %d GPU(s) with mmh_partition (whole 512 KiB blocks per rank, %d bytes of "
                        "pattern-length overlap), every step = every rank scans its partition + the offset lists are gathered; "
                        "efficiency = this run's N = 1 time per step / (N x this time)" % (stotal / (1 << 30), world, (L - 1) * ELEM),
            }
        if world > 1:
            # back to the weak-scaling shard (cpu_baseline reads the ROM of the timed configuration)
            eng.attach(buf.data_ptr(), shard)
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
            spec.apply_device(eng)
            torch.cuda.synchronize()

    if rank == 0:
        ctr = eng.counters()
        assert (np.diff(offs.astype(np.int64)) > 0).all(), "gathered offsets are not ascending"
        # The roofline figure is the dominant kernel's duration when it has the device to itself: the steps at
        # depth 1.  With two scans in flight the streaming kernels of consecutive scans overlap (the next one
        # starts while the last waves of this one drain), so a launch there lasts longer than its share of the
        # device -- those durations are reported beside it, not used.
        alone_filt, alone_tot, alone_src = kernel_alone_filt, kernel_alone_tot, (
            "%d synchronous scans behind the timed region, ONE launch over the whole shard each (MMH_ROUTE_NO_SPLIT): launches that "
            "overlap nothing" % len(kernel_alone_filt))
        filt = float(np.mean(alone_filt))
        assert filt > 0, "the library reported no streaming-phase timing"
        assert float(np.mean(filt_ms)) > 0
        achieved = shard / (filt * 1e-3) / 1e9
        # (C3 runs the same kernel instantiation with other constants: the same traffic per byte in all likelihood, but not what was counted)
        traffic, traffic_src = live_traffic[:2] if live_traffic[0] is not None else pmc_traffic(mm, shard) if kernel_name == "mm_filter_u8<4>" and args.config in ("C2", "C5") else (
            None, "the committed PMC passes were taken on mm_filter_u8<4> under C2's workload (C5: the same scan on a bigger shard), not on %s under %s" % (
                kernel_name, args.config))
        res = {
            "metric": "GB/s scanned (%g GiB synthetic ROM per GPU, %d-char %d-bit relative pattern%s)" % (
                gib, L, 8 * ELEM, "" if not WC else ", %d wildcards" % KEYWORD.count(chr(WC))),
            "value": total * args.steps / elapsed / 1e9,
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u%d" % (8 * ELEM),
            "data": "synthetic",
            "config": {
                "workload": "%s: %s, engine semantics, 512 KiB blocks, "
                            "%.1f GiB splitmix64 ROM per GPU resident in HBM, 1 planted match/MiB + boundary straddlers "
                            "+ 0x00/0xFF/ramp runs" % (args.config if args.gib_per_gpu is None else "custom", cfg["what"],
                                                       per_gpu / (1 << 30)),
                "name": args.config if args.gib_per_gpu is None else "custom",
                "rom_bytes_total": total,
                "matches": int(len(offs)),
                "candidates_rank0": ctr["candidates"],
                "parallelism": "%d partition(s) on block boundaries%s" % (
                    world, "" if not multi else ", offset gather: " + gather_backend + (" [RCCL STAND-IN, ranks share GPU 0]" if shared else "") +
                    (" (synchronous)" if args.sync_gather else ", overlapped with the next scan")),
                "scans_in_flight": args.depth,
                "step": ("mmh_scan_submit + mmh_scan_collect of an earlier ticket: K scans submitted and K results delivered "
                         "inside the timed region, %d tickets outstanding (two scans at work on the device) -- `value` is the PERIOD of "
                         "scans in flight; what ONE scan takes a caller of the reference's synchronous API is "
                         "`synchronous.without_timing_events` (mmh_scan, one at a time)" % args.depth
                         if args.depth > 1 else "mmh_scan: every step waits for its own result"),
                "prewarm_scans": prewarm_scans,
                "launches_by_phase": launches_by_phase,
                **({"host_delay_us": args.host_delay_us} if args.host_delay_us > 0 else {}),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": achieved,
                "peak": PEAK_HBM_GBS,
                "unit": "GB/s",
                "frac": achieved / PEAK_HBM_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_over_algorithmic": (traffic / shard) if traffic else None,
                # SURVEY 8(d)'s secondary ceilings, counted like the traffic (wave instructions x 64 lanes / ROM bytes of a launch)
                "valu_lane_ops_per_byte": (live_traffic[2]["SQ_INSTS_VALU"] * 64.0 / shard) if "SQ_INSTS_VALU" in live_traffic[2] else None,
                "lds_lane_reads_per_byte": (live_traffic[2]["SQ_INSTS_LDS"] * 64.0 / shard) if "SQ_INSTS_LDS" in live_traffic[2] else None,
                **({"traffic_not_counted_in_this_run": live_traffic[1]} if live_traffic[0] is None and live_traffic[1] else {}),
                "measured_read_ceiling_GBps": (read_probe or {}).get("mean_GBps"),
                "frac_of_measured": (achieved / read_probe["mean_GBps"]) if read_probe and read_probe.get("mean_GBps") else None,
                "measured_read_ceiling": read_probe,
                "algorithmic_bytes": shard,
                "kernel_ms": filt,
                "kernel_ms_median": float(np.median(alone_filt)),
                "kernel_ms_min": float(np.min(alone_filt)),
                "kernel_ms_source": "HIP events riding on the kernel's dispatch, mean over the last %d of %s" % (len(alone_filt), alone_src),
                "scan_device_ms": float(np.mean(alone_tot)),
                "scan_device_ms_median": float(np.median(alone_tot)),
                "scan_device_ms_min": float(np.min(alone_tot)),
                "timed_region": {
                    "scans_in_flight": args.depth,
                    "kernel_ms": float(np.mean(filt_ms)),
                    "kernel_period_ms": elapsed / args.steps * 1e3,
                    "bytes_per_period_GBps": shard / (elapsed / args.steps) / 1e9,
                    "note": ("launch durations in the timed region; with scans in flight consecutive streaming kernels overlap, "
                             "so kernel_ms here exceeds the period at which launches complete" if args.depth > 1 else
                             "one scan at a time: the same launches as kernel_ms above"),
                },
            },
            "stages_ms": {"filter": filt, "resolve_order_publish": float(np.mean(alone_tot)) - filt,
                          "device_total": float(np.mean(alone_tot)), "host_wall_per_step": elapsed / args.steps * 1e3},
            "counters_rank0": ctr,
            # what RCCL saw (the library's communicator on every rank; [0]: N = 1 without a communicator) and the streaming
            # kernel's own duration on every rank
            "rccl_ranks": {"min": int(min(rccl_ranks)), "max": int(max(rccl_ranks)), "expected": world if native else 0},
            "kernel_ms_per_rank": {"min": min(kernel_ms_ranks), "max": max(kernel_ms_ranks),
                                   **({"all": kernel_ms_ranks} if world > 1 else {})},
            "health_rank0": eng.health(),
        }
        if shared:
            res["shared_device"] = ("TEST RUN: the %d ranks share GPU 0 and the library's RCCL calls are served by tests/shim/libfake_rccl.so "
                                    "(shared memory): the gather path is exercised, the value is NOT a multi-GPU figure" % world)
        if world > 1:
            res["roofline"]["measured_read_ceiling_per_rank_GBps"] = {"min": min(probe_ranks), "max": max(probe_ranks)}
        if strong is not None:
            res["strong"] = strong
            if args.scaling == "strong":
                # the strong-scaling figure as the line's value: ONE ROM over N GPUs
                res["weak"] = {"value": res["value"], "ms_per_step": res["ms_per_step"], "rom_bytes_total": total}
                res["value"], res["ms_per_step"], res["scaling"] = strong["GBps"], strong["ms_per_step"], "strong"
                res["metric"] = "GB/s scanned (ONE %g GiB synthetic ROM over %d GPU(s), %d-char %d-bit relative pattern)" % (
                    gib, world, L, 8 * ELEM)
                res["config"]["rom_bytes_total"] = per_gpu
        if multi:
            # where an N > 1 step's time goes besides the scan: the collective + packing on the device
            # (HIP events on the communication stream) and the host's share of start + finish
            res["gather_ms"] = {
                "device_collective_and_pack": float(np.mean(gather_dev)) if gather_dev else None,
                "host_start_plus_finish": float(np.mean(gather_host)) if gather_host else None,
                # what a step costs beyond the streaming kernel run alone: an overlapped gather should leave it near 0
                "per_step_beyond_streaming_kernel": elapsed / args.steps * 1e3 - filt,
            }
            res["overlap"] = not args.sync_gather
            res["gather_backend"] = gather_backend
            if gather_check:
                res["gather_check"] = gather_check
            if gather_note:
                res["gather_note"] = gather_note
        if not args.no_other_depth:
            same = bool(np.array_equal(offs_other, offs))
            assert same, "the two depths delivered different lists"
            res["in_flight" if other_depth > 1 else "synchronous"] = {
                "value": total * args.steps / elapsed_other / 1e9, "unit": "GB/s", "ms_per_step": elapsed_other / args.steps * 1e3,
                "kernel_ms": float(np.mean(filt_other)), "scan_device_ms": float(np.mean(tot_other)),
                "parts": other_parts if other_depth == 1 else 0,
                **({"ms_per_step_median": float(np.median(other_per_step)), "ms_per_step_min": float(np.min(other_per_step)),
                    "ms_per_step_max": float(np.max(other_per_step))} if other_per_step else {}),
                "kernel_ms_is": ("the streaming kernels of the scan's parts SUMMED (they overlap: mmh_scan runs a ROM of >= 1 GiB as a pipeline "
                                 "of parts), scan_device_ms = the pipeline's wall time on the host" if other_depth == 1 and not args.no_split and shard >= (1 << 30)
                                 else "HIP events on the scan's own launches"),
                "same_offsets": same,
                **({"without_timing_events": {
                    "ms_per_step": elapsed_untimed / args.steps * 1e3, "value": total * args.steps / elapsed_untimed / 1e9,
                    "ms_per_step_median": float(np.median(untimed_per_step)) if untimed_per_step else None,
                    "what": "the same K scans with mmh_set_timing(0): no HIP event at a scan's start (it costs the first dispatch ~4.5 us) -- "
                            "how the include/mmoore facade runs; every other figure of this line is taken with the events on"}}
                   if other_depth == 1 and elapsed_untimed > 0 else {}),
                "note": "not the headline value: the same K steps " + (
                    "through mmh_scan_submit / mmh_scan_collect, %d tickets outstanding" % other_depth if other_depth > 1 else
                    "through mmh_scan, one scan at a time: the latency of a single 4 GiB scan as a caller sees it"),
            }
        if not multi and args.config == "C2" and args.gib_per_gpu is None and not args.no_other_configs:
            # BASELINE.json's other single-GPU configurations, NOT part of `value`
            res["other_configs"] = {"C1": c1_config(mm, 2 * args.other_scans)}
            res["other_configs"].update({name: other_config(mm, torch, dev, name, args.other_scans, 2 * args.other_scans)
                                         for name in OTHER_CONFIGS})
        if not args.no_cpu_baseline:
            # (N > 1: rank 0 alone, behind every timed region; the other ranks wait at the barrier below)
            cb, cpu_offs, ncov, e2e = cpu_baseline(mm, eng, shard, cfg, args.cpu_sample_mib << 20, args.cpu_warmups, args.cpu_runs,
                                                   with_end_to_end=not args.no_end_to_end)
            res["cpu_baseline"] = cb
            if e2e is not None:
                res["end_to_end"] = e2e
            # parity of the timed configuration: the whole ROM when the CPU run covered it, else
            # the blocks fully inside the covered prefix
            mine = offs[offs < np.uint64(base + shard - (L - 1) * ELEM)] if multi else offs      # (N > 1: rank 0's partition of the gathered list)
            if ncov >= shard:
                g, c = mine.tolist(), [int(x) for x in cpu_offs]
                res["config"]["parity_vs_cpu"] = "whole ROM: %d offsets identical" % len(g) if g == c else "MISMATCH"
            else:
                lim = (ncov // BLOCK - 1) * BLOCK
                g = mine[mine < lim].tolist()
                c = [int(x) for x in cpu_offs if x < lim]
                res["config"]["parity_vs_cpu"] = "first %d MiB: %d offsets identical" % (lim >> 20, len(g)) if g == c else "MISMATCH"
            assert g == c, "GPU offsets differ from the reference CPU engine"
        print(json.dumps(res), flush=True)            # (flushed here: a redirected stdout is block-buffered, and what runs at exit must not cost the line)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
