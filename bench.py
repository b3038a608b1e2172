#!/usr/bin/env python3
"""bench.py -- GB/s scanned by the MI355X relative-search engine.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], "C2"): 8-bit relative search, 12-character
keyword, engine semantics with the reference's default 512 KiB blocks, on a
synthetic ROM already resident in HBM.  One step = one full scan of the ROM:
filter kernel + resolver + ordering + D2H of the offsets (+ the RCCL gather of
the per-GPU offset lists at N > 1).  Weak scaling: every GPU holds its own
4 GiB partition (block-aligned, pattern-length overlap) of an N x 4 GiB ROM.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel
(mm_filter_u8) with HIP events recorded on the scan's own stream;
`cpu_baseline` times the reference's multi-threaded SearchEngine<uint8_t>::run
(oracle/_ref, built from the unmodified reference sources) on a bounded sample
of the same ROM on the host cores -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KEYWORD = "relativesrch"
BLOCK = 524288
SEED = 42
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak 8 TB/s (measured achievable ~6.3)


def cpu_baseline(eng, spec_bytes, plan_kw, sample_bytes):
    """Reference CPU engine on the first sample_bytes of the same ROM (rank 0, N = 1)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import Oracle, Ref
    sample_bytes = min(sample_bytes, spec_bytes)
    rom = eng.download(0, sample_bytes)
    cores = os.cpu_count() or 1
    if Ref.available():
        ref = Ref()
        tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
        path = os.path.join(tmpdir, "mm_cpu_baseline_%d.bin" % os.getpid())
        try:
            rom.tofile(path)
            best, offs = None, None
            for _ in range(2):
                t0 = time.perf_counter()
                offs = ref.engine(1, None, plan_kw, ord("*"), None, threads=cores, block_size=BLOCK, path=path)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
        finally:
            if os.path.exists(path):
                os.unlink(path)
        # per-core figure in the shape of the reference's own benchmark (SURVEY 8d ii):
        # MonkeyMoore<uint8_t>::search, one thread, 16 MiB of its mt19937(42) buffer, keyword "abcde"
        c1 = ref.bench_data(1, 16 << 20)
        t0 = time.perf_counter()
        ref.search(1, "abcde", c1)
        single = (16 << 20) / (time.perf_counter() - t0) / 1e9
        return dict(value=sample_bytes / best / 1e9, unit="GB/s", cores=cores, kind="reference",
                    sample="first %d MiB of the bench ROM in a tmpfs file, SearchEngine<uint8_t>::run, %d threads, "
                           "512 KiB blocks, best of 2" % (sample_bytes >> 20, cores),
                    single_thread_GBps=single,
                    single_thread_sample="MonkeyMoore<uint8_t>::search, 1 thread, 16 MiB mt19937(42) buffer, keyword 'abcde' "
                                         "(benchmarks/bench_search.cpp shape)"), offs, sample_bytes
    orc = Oracle()
    t0 = time.perf_counter()
    offs = orc.engine(orc.plan(1, plan_kw), rom, BLOCK)
    dt = time.perf_counter() - t0
    return dict(value=sample_bytes / dt / 1e9, unit="GB/s", cores=1, kind="port",
                sample="first %d MiB of the bench ROM, scalar C restatement" % (sample_bytes >> 20)), offs, sample_bytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gib-per-gpu", type=float, default=4.0)
    ap.add_argument("--cpu-sample-mib", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm-s", type=float, default=0.3, help="device clock conditioning before the warm-up steps")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N > 1: finish every step's offset gather before the next scan starts (no overlap)")
    ap.add_argument("--two-in-flight", action="store_true",
                    help="after the timed region, repeat the K steps through mmh_scan_submit / mmh_scan_collect and "
                         "report that as the extra 'two_in_flight' object (never the headline value)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from __graft_entry__ import load_package
    mm = load_package()
    if not os.path.exists(mm.LIB_PATH):
        mm.build.build_all()

    per_gpu = int(args.gib_per_gpu * (1 << 30)) // BLOCK * BLOCK
    total = per_gpu * world
    L = len(KEYWORD)
    # block-aligned partition + pattern-length overlap into the next one
    base, shard = mm.partition.shard_range(total, BLOCK, L, 1, rank, world)

    # HBM-resident shard owned by torch; the engine borrows the pointer and runs on torch's stream
    buf = torch.empty(shard + 32, dtype=torch.uint8, device=dev)
    eng = mm.Engine(local_rank)
    eng.attach(buf.data_ptr(), shard)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    spec = mm.synth.RomSpec(SEED, total, KEYWORD, 1, None, False, BLOCK, base=base, nbytes=shard, partitions=8)
    spec.apply_device(eng)
    torch.cuda.synchronize()
    plan = mm.plan_relative(1, KEYWORD)

    # N > 1: the RCCL gather of the per-GPU offset lists (already ascending, partitions in rank
    # order) is started right after a scan and finished after the NEXT scan has been run: the
    # collective and its copies overlap that scan (one gather in flight; --sync-gather turns the
    # overlap off).  Every step still delivers one merged list; drain() delivers the last one.
    gatherer = mm.partition.OffsetGather(rank, world, dev, dist) if world > 1 else None
    in_flight = []

    def step():
        offs = eng.scan(plan, block_bytes=BLOCK, base_offset=base)
        if world == 1:
            return offs
        if args.sync_gather:
            merged = gatherer.finish(gatherer.start(offs, async_op=False))
            return merged if rank == 0 else offs
        in_flight.append(gatherer.start(offs))
        merged = gatherer.finish(in_flight.pop(0)) if len(in_flight) == 2 else None
        return merged if (rank == 0 and merged is not None) else offs

    def drain(last):
        merged = None
        while in_flight:
            merged = gatherer.finish(in_flight.pop(0))
        return merged if (rank == 0 and merged is not None) else last

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Device conditioning (not a step of the workload): after the idle setup phase the MI355X
    # needs a few tens of ms of sustained load before its memory/fabric clocks are back up --
    # the first ~20 scans run ~12 % slower than steady state.  Scan until PREWARM_S have
    # passed, then do the W warm-up steps and the K timed steps of the contract.
    t_pre = time.perf_counter()
    prewarm_scans = 0
    while time.perf_counter() - t_pre < args.prewarm_s:
        eng.scan(plan, block_bytes=BLOCK, base_offset=base)
        prewarm_scans += 1
    offs = None
    for _ in range(args.warmup):
        offs = step()
    drain(offs)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        offs = step()
    offs = drain(offs)                                   # the last gather belongs to the timed region
    fence()
    elapsed = time.perf_counter() - t0
    # HIP-event timings of the timed steps: recorded on the scan's stream during the steps,
    # read back afterwards (the library keeps the event triples of the last 64 scans)
    filt_ms, tot_ms = eng.timing_history(min(args.steps, 64))
    post_ms = tot_ms - filt_ms

    # Extra, NOT part of `value`: the same K steps with two scans in flight (mmh_scan_submit /
    # mmh_scan_collect): the host's share of a scan and the kernels behind the streaming filter
    # overlap the next scan's filter.  Every step still delivers its own (gathered) result.
    def pipelined(k):
        prev, last = None, None
        for _ in range(k):
            t = eng.submit(plan, block_bytes=BLOCK, base_offset=base)
            if prev is not None:
                last = eng.collect(prev)
                if world > 1:
                    last = gatherer.finish(gatherer.start(last, async_op=False))
            prev = t
        last = eng.collect(prev)
        if world > 1:
            last = gatherer.finish(gatherer.start(last, async_op=False))
        return last
    offs_pipe, elapsed_pipe = None, 0.0
    if args.two_in_flight:
        pipelined(max(args.warmup, 4))
        fence()
        t1 = time.perf_counter()
        offs_pipe = pipelined(args.steps)
        fence()
        elapsed_pipe = time.perf_counter() - t1
    if world > 1:
        tmax = torch.tensor([elapsed, elapsed_pipe], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, elapsed_pipe = float(tmax[0].item()), float(tmax[1].item())

    if rank == 0:
        ctr = eng.counters()
        assert (np.diff(offs.astype(np.int64)) > 0).all(), "gathered offsets are not ascending"
        filt = float(np.mean(filt_ms))
        achieved = shard / (filt * 1e-3) / 1e9
        # HBM traffic per launch comes from PMC counters, which need their own rocprofv3 passes
        # (profiles/README.md); scale the committed measurement to this run's shard size
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)
            traffic = pmc["hbm_traffic_bytes_per_launch"] * shard / pmc["algorithmic_bytes_per_launch"]
            traffic_src = "profiles/r01_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH doubled per the gfx950 rule)"
        res = {
            "metric": "GB/s scanned (4 GiB synthetic ROM per GPU, 12-char 8-bit relative pattern)",
            "value": total * args.steps / elapsed / 1e9,
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "C2: 8-bit relative search, keyword '%s' (L=12), engine semantics, 512 KiB blocks, "
                            "%.1f GiB splitmix64 ROM per GPU resident in HBM, 1 planted match/MiB + boundary straddlers "
                            "+ 0x00/0xFF/ramp runs" % (KEYWORD, per_gpu / (1 << 30)),
                "rom_bytes_total": total,
                "matches": int(len(offs)),
                "candidates_rank0": ctr["candidates"],
                "parallelism": "%d partition(s) on block boundaries, RCCL offset gather%s" % (
                    world, "" if world == 1 else (" (synchronous)" if args.sync_gather else " overlapped with the next scan")),
                "prewarm_scans": prewarm_scans,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "mm_filter_u8<4>",
                "achieved": achieved,
                "peak": PEAK_HBM_GBS,
                "unit": "GB/s",
                "frac": achieved / PEAK_HBM_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes": shard,
                "kernel_ms": filt,
                "scan_device_ms": float(np.mean(tot_ms)),
                "scan_device_ms_median": float(np.median(tot_ms)),
                "scan_device_ms_min": float(np.min(tot_ms)),
                "kernel_ms_median": float(np.median(filt_ms)),
                "kernel_ms_min": float(np.min(filt_ms)),
            },
            "stages_ms": {"filter": filt, "resolve_order_publish": float(np.mean(post_ms)),
                          "device_total": float(np.mean(tot_ms)), "host_wall_per_step": elapsed / args.steps * 1e3},
            "counters_rank0": ctr,
        }
        if args.two_in_flight:
            res["two_in_flight"] = {
                "value": total * args.steps / elapsed_pipe / 1e9, "unit": "GB/s", "ms_per_step": elapsed_pipe / args.steps * 1e3,
                "same_offsets": bool(np.array_equal(offs_pipe, offs)),
                "note": "not the headline value: the same K steps through mmh_scan_submit / mmh_scan_collect, two scans in flight",
            }
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_offs, nsample = cpu_baseline(eng, shard, KEYWORD, args.cpu_sample_mib << 20)
            res["cpu_baseline"] = cb
            # parity of the timed configuration on the sample (blocks fully inside it)
            lim = (nsample // BLOCK - 1) * BLOCK
            g = offs[offs < lim].tolist()
            c = [int(x) for x in cpu_offs if x < lim]
            res["config"]["parity_vs_cpu_sample"] = bool(g == c)
            assert g == c, "GPU offsets differ from the reference CPU engine on the sample"
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
