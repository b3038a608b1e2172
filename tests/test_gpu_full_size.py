# SPDX-License-Identifier: GPL-3.0-or-later
"""BASELINE.json's GPU configurations at their full sizes, bit-exact against the oracle
(which is run over all host cores on block-aligned slices), plus size-independent
properties.  Needs a real MI355X and ~10 GiB of host memory."""
import numpy as np
import pytest

from _oracle import oracle_engine_parallel

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
    assert t["total_ms"] < 20.0, t                        # 64 GiB in 11 ms at the 4 GiB rate
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


def test_dense_search_split_pipeline(mm, oracle):
    """mmh_scan on a search that the previous scan found dense (tens of thousands of candidates, ROM of a GiB or more in
    HBM, engine semantics) runs as a pipeline of block-aligned parts through the submit lanes (csrc/mm_capi.hip:
    scan_split): the same list as one scan of the whole ROM -- plants at the parts' edges, 16-bit with both alignments,
    a buffer that is too small, a base offset, then a sparse search that must leave the pipeline again."""
    rng = np.random.default_rng(77)
    for elem, kw, nbytes, be in ((1, "monkeybars", (2 << 30) + 524288 * 3 + 77, False), (2, "texts", (1 << 30) + 4098, True)):
        n_el = nbytes // elem
        rom = rng.integers(0, 256, nbytes, dtype=np.uint8)
        # ~60 K plants, among them some straddling the boundaries the pipeline cuts at (multiples of nblocks / parts blocks)
        pos = np.sort(rng.choice((nbytes - 64) // 32, size=60000, replace=False)) * 32 + rng.integers(0, 16, 60000)
        nblocks = -(-nbytes // BLOCK)
        parts = max(2, nbytes >> 30)
        edges = [nblocks * i // parts * BLOCK for i in range(1, parts)]
        pos = np.concatenate([pos, [e - 4 * elem for e in edges], [e - 1 - elem for e in edges], [e + elem for e in edges]]).astype(np.int64)
        vals = np.array([ord(c) for c in kw], np.int64)
        for p in pos:
            shift = int(rng.integers(0, 100))
            for j, v in enumerate(vals):
                x = int(v) + shift
                if elem == 1:
                    rom[p + j] = x
                else:
                    b = x.to_bytes(2, "big" if be else "little")
                    rom[p + 2 * j], rom[p + 2 * j + 1] = b[0], b[1]
        want = oracle_engine_parallel(oracle, oracle.plan(elem, kw), rom, BLOCK, be)
        plan = mm.plan_relative(elem, kw)
        with mm.Engine(0) as eng:
            eng.upload(rom)
            first = eng.scan(plan, block_bytes=BLOCK, big_endian=be, cap=1 << 18)
            assert first.tolist() == want.tolist() and eng.counters()["candidates"] >= 32768 and eng.counters()["path"] == 0, eng.counters()
            v0 = eng.health()["validated"]
            again = eng.scan(plan, block_bytes=BLOCK, big_endian=be, cap=1 << 18)           # the split pipeline
            assert again.tolist() == want.tolist()
            assert eng.health()["validated"] - v0 == parts, (eng.health(), parts)          # one validated block per part
            assert eng.counters()["matches"] == len(want) and eng.counters()["candidates"] >= len(want)
            based = eng.scan(plan, block_bytes=BLOCK, big_endian=be, base_offset=1 << 40, cap=16)   # too small a buffer: twice through it
            assert (based - np.uint64(1 << 40)).tolist() == want.tolist()
            # tickets of the caller's own in between: the pipeline steps aside
            t = eng.submit(plan, block_bytes=BLOCK, big_endian=be)
            assert eng.scan(plan, block_bytes=BLOCK, big_endian=be, cap=1 << 18).tolist() == want.tolist()
            assert eng.collect(t, cap=1 << 18).tolist() == want.tolist()
            # another keyword on the same ROM: sparse, scanned the usual way
            other = mm.plan_relative(elem, "zqxjkvbwpy"[: len(kw)])
            sparse = eng.scan(other, block_bytes=BLOCK, big_endian=be)
            assert sparse.tolist() == oracle_engine_parallel(oracle, oracle.plan(elem, "zqxjkvbwpy"[: len(kw)]), rom, BLOCK, be).tolist()
            assert eng.health()["fallbacks"] == 0
