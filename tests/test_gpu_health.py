# SPDX-License-Identifier: GPL-3.0-or-later
"""Route health (include/mmoore_hip.h): the first-use self-test, the validation of every block a polled scan publishes,
the fallback to the plain kernels when a block fails it, and the run-time route switches -- the fence around the fast
routes MonkeyMoore<Ty>::search takes at the reference benchmark's sizes (benchmarks/bench_search.cpp:67-105)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_selftest_passes_with_every_route_on(mm, gpu_engine):
    assert mm.selftest_run(0) == 0
    h = gpu_engine.health()
    assert h["selftest"] == 1 and h["process_routes_off"] == 0 and h["routes_off"] == 0, h


def test_known_answer_through_every_route(mm, oracle):
    rom, kw, block, expected = mm.selftest_kat()
    assert oracle.engine(oracle.plan(1, kw), rom, block).tolist() == expected.tolist()
    plan = mm.plan_relative(1, kw)
    with mm.Engine(0) as eng:
        for mask in range(16):
            eng.set_route(mask)
            eng.upload(rom)
            assert eng.scan(plan, block_bytes=block).tolist() == expected.tolist(), mask
            assert eng.counters()["path"] == 0, eng.counters()         # (the self-test's scan ends in the slot validation)
            assert eng.collect(eng.submit(plan, block_bytes=block)).tolist() == expected.tolist(), (mask, "lanes")
            for engine in (1, 2):
                eng.set_engine(engine)
                assert eng.scan(plan, block_bytes=block).tolist() == expected.tolist(), (mask, engine)
            eng.set_engine(0)
            assert eng.health()["routes_off"] == mask
        h = eng.health()
        assert h["fallbacks"] == 0 and h["validated"] > 0, h


KW = "monkeybars"      # (ten distinct deltas: chains merge within a window, candidates settle without the second phase)


def _rom_with_matches(rng, nbytes, kw=KW, every=4096):
    rom = rng.integers(0, 256, nbytes).astype(np.uint8)
    for at in range(64, nbytes - 16, every):
        base = int(rng.integers(0, 200))
        rom[at:at + len(kw)] = [base + ord(c) - ord("a") for c in kw]
    return rom


@pytest.mark.parametrize("shape", ["single-launch", "streaming + tail", "lanes"])
@pytest.mark.parametrize("kind,reason", [(1, 1), (5, 2), (2, 3), (3, 4), (4, 5)])
def test_a_damaged_block_is_caught_and_the_scan_rerun(mm, oracle, shape, kind, reason):
    """mmh_debug_inject damages what the next polled scan published the way a lost or reordered PCIe write would:
    the scan must still return the reference's list (rerun through the plain kernels) and say so in mmh_health."""
    rng = np.random.default_rng(kind * 10 + len(shape))
    nbytes = 300_000 if shape == "single-launch" else (9 << 20) + 77
    rom = _rom_with_matches(rng, nbytes, every=4096 if nbytes < (1 << 20) else 65536)
    plan = mm.plan_relative(1, KW)
    want = oracle.engine(oracle.plan(1, KW), rom, 65536)
    assert len(want) >= 8
    with mm.Engine(0) as eng:
        eng.upload(rom)
        scan = (lambda: eng.collect(eng.submit(plan, block_bytes=65536))) if shape == "lanes" else (lambda: eng.scan(plan, block_bytes=65536))
        assert scan().tolist() == want.tolist()
        assert eng.health()["fallbacks"] == 0 and eng.counters()["path"] == 0, eng.counters()
        eng.inject(kind)
        assert scan().tolist() == want.tolist(), (shape, kind)
        h = eng.health()
        assert h["fallbacks"] == 1 and h["fallback_reason"] == reason and h["last_reason"] == reason, h
        # ... and the next scan is trusted again, the first violation stays on record
        assert scan().tolist() == want.tolist()
        h = eng.health()
        assert h["fallbacks"] == 1 and h["fallback_reason"] == reason, h


def test_stale_slots_cannot_pass_for_offsets(mm, oracle):
    """Slots of earlier scans are poisoned before the next launch: a long list followed by a short one, followed by a
    different one of the first's length -- every list must be its own (the validation would flag a stale slot)."""
    rng = np.random.default_rng(5)
    plan = mm.plan_relative(1, KW)
    oplan = oracle.plan(1, KW)
    with mm.Engine(0) as eng:
        for nbytes, every in ((400_000, 512), (400_000, 65536), (400_001, 640), (6 << 20, 2048), (6 << 20, 1 << 20), (6 << 20, 4096)):
            rom = _rom_with_matches(rng, nbytes, every=every)
            eng.upload(rom)
            want = oracle.engine(oplan, rom, 524288).tolist()
            assert eng.scan(plan, block_bytes=524288).tolist() == want, (nbytes, every)
            assert eng.collect(eng.submit(plan, block_bytes=524288)).tolist() == want, (nbytes, every, "lanes")
        h = eng.health()
        assert h["fallbacks"] == 0 and h["late_slots"] == 0, h


def test_bucket_store_grows_with_the_rom(mm, oracle):
    """The bucketed candidate store is sized from the ROM (round 4): small ROM first, then larger ones, then small again."""
    rng = np.random.default_rng(9)
    plan = mm.plan_relative(1, KW)
    oplan = oracle.plan(1, KW)
    with mm.Engine(0) as eng:
        eng.set_route(mm.ROUTE_NO_SINGLE_LAUNCH)              # small ROMs through the bucketed store as well
        for nbytes in (70_000, 5 << 20, 40 << 20, 90_000, 17 << 20):
            rom = _rom_with_matches(rng, nbytes, every=3000)
            eng.upload(rom)
            want = oracle.engine(oplan, rom, 524288).tolist()
            assert eng.scan(plan, block_bytes=524288).tolist() == want, nbytes
            assert eng.collect(eng.submit(plan, block_bytes=524288)).tolist() == want, (nbytes, "lanes")
        assert eng.health()["fallbacks"] == 0
