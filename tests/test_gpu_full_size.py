# SPDX-License-Identifier: GPL-3.0-or-later
"""BASELINE.json's GPU configurations at their full sizes, bit-exact against the oracle
(which is run over all host cores on block-aligned slices), plus size-independent
properties.  Needs a real MI355X and ~10 GiB of host memory."""
import os

import numpy as np
import pytest

from _oracle import oracle_engine_parallel

# wall-clock expectations are measurements (printed); only a soak run that sets MMOORE_TEST_TIMING=1 asserts them
TIMING_GATES = os.environ.get("MMOORE_TEST_TIMING", "0") not in ("", "0")

pytestmark = pytest.mark.gpu

BLOCK = 524288


def _download(eng, n, piece=1 << 30):
    rom = np.empty(n, np.uint8)
    for first in range(0, n, piece):
        k = min(piece, n - first)
        rom[first:first + k] = eng.download(first, k)
    return rom


def _full_config(mm, oracle, eng, nbytes, keyword, elem, wildcard=None, be=False):
    spec = mm.synth.RomSpec(42, nbytes, keyword, elem, wildcard, be, BLOCK)
    eng.alloc(nbytes)
    spec.apply_device(eng)
    wc = wildcard or 0
    plan = mm.plan_relative(elem, keyword, wc)
    got = eng.scan(plan, block_bytes=BLOCK, big_endian=be)
    ctr = eng.counters()
    rom = _download(eng, nbytes)
    want = oracle_engine_parallel(oracle, oracle.plan(elem, keyword, wc), rom, BLOCK, be)
    assert got.tolist() == want.tolist()
    assert len(got) >= nbytes >> 20                      # at least the one plant per MiB
    assert ctr["path"] != 1                              # the filter + resolver path, not the fallback
    return got, rom, plan


def test_c2_4gib_12char(mm, gpu_engine, oracle):
    got, rom, plan = _full_config(mm, oracle, gpu_engine, 4 << 30, "relativesrch", 1)
    # idempotence + base offset linearity: the same ROM scanned again with a base reports base + offsets
    again = gpu_engine.scan(plan, block_bytes=BLOCK, base_offset=1 << 36)
    assert (again - np.uint64(1 << 36)).tolist() == got.tolist()
    # partition property (what the multi-GPU path relies on): scanning the two halves as
    # block-aligned shards with overlap reproduces the whole
    half = (4 << 30) // 2
    parts = []
    for first, n in ((0, half + 11), (half, half)):
        gpu_engine.upload(rom[first:first + n])
        parts.append(gpu_engine.scan(plan, block_bytes=BLOCK, base_offset=first))
    assert np.concatenate(parts).tolist() == got.tolist()


def test_c3_4gib_16char_3_wildcards(mm, gpu_engine, oracle):
    _full_config(mm, oracle, gpu_engine, 4 << 30, "re*ative*ear*hxy", 1, wildcard=ord("*"))


def test_c4_8gib_16bit_le(mm, gpu_engine, oracle):
    got, rom, plan = _full_config(mm, oracle, gpu_engine, 8 << 30, "textsrch", 2)
    # the 16-bit odd-boundary rule: no reported offset is k*B - 1
    assert not np.any((got % np.uint64(BLOCK)) == np.uint64(BLOCK - 1))
    assert np.any(got % np.uint64(2) == 1)               # odd alignments are found


def test_c5_shard_8gib_8bit_ragged(mm, gpu_engine, oracle):
    # one GPU's share of BASELINE C5 (64 GiB over 8 GPUs = 8 GiB per GPU, 8-bit, 12 symbols), with
    # a ragged tail and a non-zero partition base: offsets beyond 2^32, the edge kernel behind the
    # last whole 4 KiB group, a last block shorter than the others
    nbytes = (8 << 30) + 12345
    base = 3 * (8 << 30)
    spec = mm.synth.RomSpec(42, base + nbytes, "relativesrch", 1, None, False, BLOCK, base=base, nbytes=nbytes)
    gpu_engine.alloc(nbytes)
    spec.apply_device(gpu_engine)
    tail = np.frombuffer(b"relativesrch", np.uint8) - 30
    gpu_engine.poke(nbytes - 12, tail)                    # a match ending on the ROM's last byte
    gpu_engine.poke(nbytes - 5000, tail)
    plan = mm.plan_relative(1, "relativesrch")
    got = gpu_engine.scan(plan, block_bytes=BLOCK, base_offset=base)
    rom = _download(gpu_engine, nbytes)
    want = oracle_engine_parallel(oracle, oracle.plan(1, "relativesrch"), rom, BLOCK) + np.uint64(base)
    assert got.tolist() == want.tolist()
    assert len(got) >= 8192 and int(got[-1]) == base + nbytes - 12
    # two scans in flight deliver the same
    t1 = gpu_engine.submit(plan, block_bytes=BLOCK, base_offset=base)
    t2 = gpu_engine.submit(plan, block_bytes=BLOCK, base_offset=0)
    assert gpu_engine.collect(t1).tolist() == want.tolist()
    assert (gpu_engine.collect(t2) + np.uint64(base)).tolist() == want.tolist()


def test_c5_64gib_on_one_gpu(mm, gpu_engine, oracle):
    # BASELINE C5's whole 64 GiB ROM resident in ONE MI355X's 288 GB of HBM: 131072 blocks, offsets
    # up to 2^36, one scan, compared in full with the oracle run over the host cores.  (The bench
    # shards C5 over 8 GPUs; the engine itself does not need to.)
    import psutil
    nbytes = 64 << 30
    if psutil.virtual_memory().available < (96 << 30):
        pytest.skip("needs ~70 GiB of host memory for the oracle's copy of the ROM")
    try:
        gpu_engine.alloc(nbytes)
    except mm.MMError:
        pytest.skip("not enough free HBM for a 64 GiB ROM")
    # straddlers at every 64th block boundary, partition-boundary plants and the three runs (one
    # plant per MiB would be 65536 host-side pokes; the 4-8 GiB configurations above have those)
    spec = mm.synth.RomSpec(42, nbytes, "relativesrch", 1, None, False, BLOCK, plants_per_mib=0)
    spec.apply_device(gpu_engine)
    plan = mm.plan_relative(1, "relativesrch")
    for _ in range(3):
        got = gpu_engine.scan(plan, block_bytes=BLOCK)
    t = gpu_engine.timings()
    rom = _download(gpu_engine, nbytes)
    want = oracle_engine_parallel(oracle, oracle.plan(1, "relativesrch"), rom, BLOCK)
    assert got.tolist() == want.tolist()
    assert len(got) >= 2000 and int(got[-1]) > (63 << 30)
    print("64 GiB on one GPU: %s" % (t,))                 # 11 ms at the 4 GiB rate; a measurement, gated only on request
    if TIMING_GATES:
        assert t["total_ms"] < 20.0, t
    gpu_engine.alloc(1 << 20)                             # give the HBM back to the other tests


@pytest.mark.parametrize("kw,wc", [("relativesrch", 0), ("re*ative*ear*hxy", ord("*")), ("a" * 40 + "bcdefghij" + "k" * 15, 0)])
def test_forward_engine_4gib(mm, gpu_engine, oracle, kw, wc):
    """The forward engine (mm_forward: single pass, decoupled look-back over 128 Ki batches of this ROM)
    at C2 / C3 size, and with a 64-symbol keyword (the wide phase maps), against the oracle and
    against the filter + resolver path."""
    n = 4 << 30
    spec = mm.synth.RomSpec(42, n, kw, 1, wc or None, False, BLOCK)
    gpu_engine.alloc(n)
    spec.apply_device(gpu_engine)
    plan = mm.plan_relative(1, kw, wc)
    gpu_engine.set_engine(2)
    try:
        got = gpu_engine.scan(plan, block_bytes=BLOCK)
        assert gpu_engine.counters()["path"] == 3
        whole = gpu_engine.scan(plan)                    # one chain over the whole 4 GiB: look-back across 131072 batches
    finally:
        gpu_engine.set_engine(0)
    rom = _download(gpu_engine, n)
    oplan = oracle.plan(1, kw, wc)
    want = oracle_engine_parallel(oracle, oplan, rom, BLOCK)
    assert got.tolist() == want.tolist()
    assert len(got) >= 4096 or len(kw) > 32
    if len(kw) <= 32:
        assert gpu_engine.scan(plan, block_bytes=BLOCK).tolist() == want.tolist()      # the candidate path agrees
    assert whole.tolist() == oracle.search(oplan, rom).tolist()


def test_big_rom_split_pipeline(mm, oracle):
    """mmh_scan on a ROM of a GiB or more in HBM (engine semantics) runs as a pipeline of block-aligned parts through the
    submit lanes from the FIRST scan on (csrc/mm_capi.hip: scan_split) -- an eighth of the ROM (at least 256 MiB) and three
    eighths beside it, then, by the first part's candidate count, the rest in one part, in two, or in eighths: the same list as one
    scan of the whole ROM -- plants at every edge a part can have, 16-bit with both alignments, a buffer that is too small,
    a base offset, the caller's own tickets in between, sparse and dense keywords, and MMH_ROUTE_NO_SPLIT."""
    rng = np.random.default_rng(77)
    for elem, kw, nbytes, be, nplants in ((1, "monkeybars", (2 << 30) + 524288 * 3 + 77, False, 60000), (2, "texts", (1 << 30) + 4098, True, 60000),
                                          (1, "monkeybars", (3 << 30) + 99, False, 900000)):
        rom = rng.integers(0, 256, nbytes, dtype=np.uint8)
        pos = np.sort(rng.choice((nbytes - 64) // 32, size=nplants, replace=False)) * 32 + rng.integers(0, 16, nplants)
        nblocks = -(-nbytes // BLOCK)
        unit = max(-(-nblocks // 8), (256 << 20) // BLOCK)
        left = max(nblocks - 4 * unit, 0)
        cuts = {unit * i for i in range(1, 9)} | {4 * unit + -(-left // 2)}
        edges = [b * BLOCK for b in sorted(cuts) if 0 < b < nblocks]
        pos = np.concatenate([pos, [e - 4 * elem for e in edges], [e - 1 - elem for e in edges], [e + elem for e in edges]]).astype(np.int64)
        vals = np.array([ord(c) for c in kw], np.int64)
        shift = rng.integers(0, 100, len(pos))
        for j, v in enumerate(vals):
            x = (int(v) + shift).astype(np.int64)
            if elem == 1:
                rom[pos + j] = x.astype(np.uint8)
            else:
                hi, lo = (x >> 8).astype(np.uint8), (x & 0xFF).astype(np.uint8)
                rom[pos + 2 * j], rom[pos + 2 * j + 1] = (hi, lo) if be else (lo, hi)
        want = oracle_engine_parallel(oracle, oracle.plan(elem, kw), rom, BLOCK, be)
        assert len(want) > nplants // 2
        plan = mm.plan_relative(elem, kw)
        with mm.Engine(0) as eng:
            eng.upload(rom)
            v0 = eng.health()["validated"]
            first = eng.scan(plan, block_bytes=BLOCK, big_endian=be, cap=1 << 20)             # the pipeline, from the first scan on
            parts = eng.timings()["parts"]
            assert first.tolist() == want.tolist() and eng.counters()["path"] == 0, eng.counters()
            assert eng.counters()["matches"] == len(want) and eng.counters()["candidates"] >= len(want)
            expect = {(2 << 30) + 524288 * 3 + 77: 4, (1 << 30) + 4098: 3, (3 << 30) + 99: 6}[nbytes]   # 1 + 3 + two halves / 1 + 3 of 4 units + the rest / 1 + 3 + four eighths
            assert parts == expect, eng.timings()
            assert eng.health()["validated"] - v0 == parts, (eng.health(), parts)             # one validated block per part
            eng.set_route(mm.ROUTE_NO_SPLIT)
            assert eng.scan(plan, block_bytes=BLOCK, big_endian=be, cap=1 << 20).tolist() == want.tolist() and eng.timings()["parts"] == 0
            eng.set_route(0)
            based = eng.scan(plan, block_bytes=BLOCK, big_endian=be, base_offset=1 << 40, cap=16)   # too small a buffer: twice through it
            assert (based - np.uint64(1 << 40)).tolist() == want.tolist()
            # tickets of the caller's own in between: the pipeline steps aside
            t = eng.submit(plan, block_bytes=BLOCK, big_endian=be)
            assert eng.scan(plan, block_bytes=BLOCK, big_endian=be, cap=1 << 20).tolist() == want.tolist() and eng.timings()["parts"] == 0
            assert eng.collect(t, cap=1 << 20).tolist() == want.tolist()
            # another keyword on the same ROM: sparse -- two eighths, then the rest in one part
            other = mm.plan_relative(elem, "zqxjkvbwpy"[: len(kw)])
            sparse = eng.scan(other, block_bytes=BLOCK, big_endian=be)
            assert sparse.tolist() == oracle_engine_parallel(oracle, oracle.plan(elem, "zqxjkvbwpy"[: len(kw)]), rom, BLOCK, be).tolist()
            assert eng.timings()["parts"] == 3, eng.timings()
            # ... and scanned again: the same route -- a scan leaves nothing behind that the next one's route depends on (round 6)
            assert eng.scan(other, block_bytes=BLOCK, big_endian=be).tolist() == sparse.tolist() and eng.timings()["parts"] == 3, eng.timings()
            eng.poke(5, rom[5:6])
            assert eng.scan(other, block_bytes=BLOCK, big_endian=be).tolist() == sparse.tolist() and eng.timings()["parts"] == 3, eng.timings()
            assert eng.health()["fallbacks"] == 0


def test_flood_in_a_part_takes_finer_parts_then_the_flood_paths(mm, oracle):
    """A part of the pipeline whose bucketed store overflows ends the stage THERE: what the parts in front of it delivered
    stays, the ROM from that part on goes in parts a sixteenth as wide (narrower buckets: the candidate path still does it),
    and where those flood as well the rest of the ROM is one synchronous scan on the forward engine.  Nothing is remembered:
    the second scan of a search takes the route of the first."""
    rng = np.random.default_rng(5)
    nbytes = (1 << 30) + 524288 * 5
    rom = rng.integers(0, 256, nbytes, dtype=np.uint8)
    # script-like stretches: 'the' every 24 bytes in 40 stretches of 256 KiB (5461 candidates per 128 KiB, the bucket width of
    # the first attempt's 256 MiB parts: too many; 2731 per 64 KiB: fine), and 2 MiB of constant padding ('aaa' floods anything)
    text = np.frombuffer((b"the cat sat on a red mat" * (262144 // 24 + 1))[:262144], np.uint8)
    for at in rng.choice((nbytes - (1 << 20)) // 262144, size=40, replace=False):
        rom[at * 262144: at * 262144 + 262144] = text
    rom[700 << 20: 702 << 20] = 0x41
    with mm.Engine(0) as eng:
        eng.upload(rom)
        for kw, path_set in (("the", (0, 2)), ("aaa", (0, 2, 3, 4, 5))):
            want = oracle_engine_parallel(oracle, oracle.plan(1, kw), rom, BLOCK)
            got = eng.scan(mm.plan_relative(1, kw), block_bytes=BLOCK, cap=1 << 22)
            assert got.tolist() == want.tolist() and len(want) > 100000
            assert eng.counters()["path"] in path_set, (kw, eng.counters(), eng.timings())
            parts, path = eng.timings()["parts"], eng.counters()["path"]
            if kw == "the":
                assert parts > 8, eng.timings()                                              # the finer parts settled it
            again = eng.scan(mm.plan_relative(1, kw), block_bytes=BLOCK, cap=1 << 22)
            assert again.tolist() == want.tolist()
            assert (eng.timings()["parts"], eng.counters()["path"]) == (parts, path)         # the same route again
        eng.poke(0, rom[:1])                                                                 # the ROM "changed": nothing is remembered
        assert eng.scan(mm.plan_relative(1, "aaa"), block_bytes=BLOCK, cap=1 << 22).tolist() == want.tolist()
        assert eng.health()["fallbacks"] == 0


@pytest.mark.parametrize("seed", range(int(os.environ.get("MM_FUZZ_SPLIT_FIRST", "0")),
                                        int(os.environ.get("MM_FUZZ_SPLIT_FIRST", "0")) + int(os.environ.get("MM_FUZZ_SPLIT", "8"))))
def test_fuzz_split_pipeline(mm, oracle, seed):
    """Random big ROMs through mmh_scan's pipeline of parts (ROMs of >= 1 GiB): random size, element width, byte order, block
    size, keyword (plain / wildcard / mixed case / custom character sequence / value scan, 3 .. 40 symbols), plant density from
    sparse to a flood stretch -- the default scan, the
    one-launch scan (MMH_ROUTE_NO_SPLIT) and the oracle on the whole ROM must agree; so must a scan after the ROM changed."""
    rng = np.random.default_rng(9100 + seed)
    elem = int(rng.choice([1, 1, 2]))
    be = bool(elem == 2 and rng.random() < 0.5)
    nbytes = int(rng.integers(1 << 30, (5 << 30) // 2)) // elem * elem + int(rng.integers(0, 3)) * (elem == 1)
    block = int(rng.choice([524288, 65536, 1 << 20, 524288 + 16, 4 << 20]))
    L = int(rng.integers(3, 41))
    letters = rng.integers(97, 123, L)
    wc, seq, values = 0, None, None
    kw = [int(c) for c in letters]
    mode = str(rng.choice(["plain", "plain", "wild", "case", "seq", "values"]))
    if mode == "wild" and L >= 6:
        wc = ord("*")
        for i in rng.choice(np.arange(1, L - 1), size=max(1, L // 8), replace=False):
            kw[int(i)] = wc
    elif mode == "case" and L >= 6:
        wc = ord("*")                                        # (the engine's default wildcard; the minority case becomes wildcards)
        for i in rng.choice(np.arange(0, L), size=max(1, L // 6), replace=False):
            kw[int(i)] -= 32
    elif mode == "seq":
        seq = [int(c) for c in rng.permutation(np.arange(48, 48 + 40))]      # a custom character sequence of 40 symbols
        kw = [int(c) for c in rng.choice(seq, L)]
    elif mode == "values":
        values = [int(v) for v in rng.integers(0, 120, L)]                   # value scan: the keyword IS the values
    # the element values a plant carries (None: a wildcard slot keeps the ROM's bytes)
    if values is not None:
        plant_vals = list(values)
    elif seq is not None:
        plant_vals = [seq.index(c) for c in kw]
    elif mode == "case" and wc:
        upper = sum(1 for c in kw if c < 97)
        minority_upper = upper <= L - upper                  # (ties: the upper-case letters go, monkey_moore.cpp:163-180)
        plant_vals = [None if ((c < 97) == minority_upper) else c for c in kw]
    else:
        plant_vals = [None if (wc and c == wc) else c for c in kw]
    vals = np.array([0 if v is None else v for v in plant_vals], np.int64)
    lit = np.array([v is not None for v in plant_vals])
    rom = rng.integers(0, 256, nbytes, dtype=np.uint8)
    density = str(rng.choice(["sparse", "medium", "dense", "flood"]))
    nplants = {"sparse": 500, "medium": 40000, "dense": 400000, "flood": 20000}[density]
    span = L * elem
    pos = np.sort(rng.choice((nbytes - 4 * span - 64) // 64, size=nplants, replace=False)).astype(np.int64) * 64 + rng.integers(0, 32, nplants) * elem
    if density == "flood":
        # ... most of them crowded into 8 MiB: more than a bucket of the usual parts holds
        lo = int(rng.integers(0, nbytes - (16 << 20)))
        pos[: nplants * 3 // 4] = lo + np.sort(rng.choice((8 << 20) // (2 * span), size=nplants * 3 // 4, replace=False)).astype(np.int64) * 2 * span
        pos = np.unique(pos)
    shift = rng.integers(0, 120, len(pos))
    for j in range(L):
        if not lit[j]:
            continue
        x = (int(vals[j]) + shift).astype(np.int64)
        if elem == 1:
            rom[pos + j] = x.astype(np.uint8)
        else:
            hi, lo8 = (x >> 8).astype(np.uint8), (x & 0xFF).astype(np.uint8)
            rom[pos + 2 * j], rom[pos + 2 * j + 1] = (hi, lo8) if be else (lo8, hi)
    if values is not None:
        oplan, plan = oracle.plan_values(elem, values), mm.plan_value_scan(elem, values)
    else:
        oplan, plan = oracle.plan(elem, kw, wc, seq), mm.plan_relative(elem, kw, wc, seq)
    want = oracle_engine_parallel(oracle, oplan, rom, block, be)
    with mm.Engine(0) as eng:
        eng.upload(rom)
        got = eng.scan(plan, block_bytes=block, big_endian=be, cap=1 << 20)
        t = eng.timings()
        info = (seed, mode, elem, be, nbytes, block, L, wc, density, len(want), eng.counters(), t)
        assert got.tolist() == want.tolist(), info
        if block % 16 == 0 and L <= 64:
            assert t["parts"] >= 2 or eng.counters()["path"] >= 3, info          # the pipeline (or the flood paths behind it)
        eng.set_route(mm.ROUTE_NO_SPLIT)
        assert eng.scan(plan, block_bytes=block, big_endian=be, cap=1 << 20).tolist() == want.tolist(), info
        assert eng.timings()["parts"] == 0
        eng.set_route(0)
        assert eng.scan(plan, block_bytes=block, big_endian=be, cap=1 << 20).tolist() == want.tolist(), info    # (with whatever the first scan remembered)
        # the ROM changes under the memos: a plant wiped out in the middle
        victim = int(pos[len(pos) // 2])
        rom[victim: victim + span] = 0
        eng.poke(victim, rom[victim: victim + span])
        want2 = oracle_engine_parallel(oracle, oplan, rom, block, be)
        assert eng.scan(plan, block_bytes=block, big_endian=be, cap=1 << 20).tolist() == want2.tolist(), info
        assert eng.health()["fallbacks"] == 0
