# SPDX-License-Identifier: GPL-3.0-or-later
"""What to keep when a GPU scan disagrees with the oracle: the same ROM and plan scanned again, in the same process,
through every route the library has (mmh_set_route / mmh_set_engine), the published header words, the library's
health counters and the ROM itself -- written under gpurun_out/ (merged back from the GPU box) and quoted in the
assertion.  Round 3 lost a wholesale failure (3299 of 3602 fuzz tests in one process) because only pytest's summary
line had been kept; tools/soak_fuzz.sh runs the fuzz with this switched on."""
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("MM_DIAG_DIR", os.path.join(ROOT, "gpurun_out", "fuzz_failures"))

ROUTES = {"all routes on": 0, "no single-launch": 1, "no zero-copy": 2, "no single-launch, no zero-copy": 3,
          "no buckets": 4, "plain kernels (nothing polled, no zero-copy)": 15}


def differential(mm, eng, rom, plan, want, block_bytes=0, big_endian=False, base_offset=0, cap=1 << 16):
    """Scans rom every way; returns {route name: {'equal': bool, 'n': len, 'first_difference': ...}} and restores the routes."""
    report = {}
    want = list(want)

    def note(name, scan):
        try:
            got = scan().tolist()
            first = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
            report[name] = {"equal": got == want, "n": len(got), "first_difference": None if got == want else first,
                            "got_there": got[first:first + 4], "want_there": want[first:first + 4], "counters": eng.counters(),
                            "header": eng.health()["header"]}
        except Exception as e:                                    # an error code is an answer too
            report[name] = {"equal": False, "error": repr(e)}

    try:
        for name, mask in ROUTES.items():
            eng.set_route(mask)
            eng.upload(rom)
            note(name, lambda: eng.scan(plan, block_bytes=block_bytes, big_endian=big_endian, base_offset=base_offset, cap=cap))
        eng.set_route(0)
        eng.upload(rom)
        note("submit lanes", lambda: eng.collect(eng.submit(plan, block_bytes=block_bytes, big_endian=big_endian, base_offset=base_offset), cap=cap))
        for engine, name in ((1, "sequential chain kernel"), (2, "forward engine")):
            eng.set_engine(engine)
            note(name, lambda: eng.scan(plan, block_bytes=block_bytes, big_endian=big_endian, base_offset=base_offset, cap=cap))
    finally:
        eng.set_engine(0)
        eng.set_route(0)
    report["health"] = eng.health()
    return report


def explain(mm, eng, rom, plan, got, want, label, **scan_args):
    """Called on a mismatch: differential + dump; returns the text for the assertion."""
    rom = np.ascontiguousarray(rom)
    first_health = eng.health()
    rep = differential(mm, eng, rom, plan, want, **scan_args)
    rep["health_at_failure"] = first_health
    rep["label"] = repr(label)
    rep["first_answer"] = {"n": len(got), "head": [int(v) for v in list(got)[:16]]}
    rep["want"] = {"n": len(want), "head": [int(v) for v in list(want)[:16]]}
    try:
        os.makedirs(OUT, exist_ok=True)
        stem = os.path.join(OUT, "fail_%d_%d" % (os.getpid(), int(time.time() * 1000) % 10 ** 9))
        if len(os.listdir(OUT)) < 40:                             # (a wholesale failure must not fill the 64 MiB that travel back)
            with open(stem + ".json", "w") as f:
                json.dump(rep, f, indent=1, default=str)
            if rom.nbytes <= (2 << 20):
                rom.tofile(stem + ".rom")
            plan_bytes = bytes(plan)
            with open(stem + ".plan", "wb") as f:
                f.write(plan_bytes)
    except OSError as e:
        rep["dump_error"] = repr(e)
    agree = [k for k, v in rep.items() if isinstance(v, dict) and v.get("equal") is True]
    differ = [k for k, v in rep.items() if isinstance(v, dict) and v.get("equal") is False]
    return "%r: GPU %d offsets, oracle %d; routes that agree with the oracle: %s; routes that differ: %s; health %s" % (
        label, len(got), len(want), agree, differ, first_health)


def same(mm, eng, rom, plan, got, want, label, **scan_args):
    """assert got == want, with the differential on failure."""
    got_l, want_l = got.tolist(), want.tolist()
    if got_l != want_l:
        raise AssertionError(explain(mm, eng, rom, plan, got_l, want_l, label, **scan_args))
