"""Parity of the HIP path (through the C ABI) with the oracle and the golden
vectors.  Needs a real MI355X: run with `pytest -m gpu`.  Offsets are compared
bit-exactly (integer work: no tolerance)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _mm_plan(mm, c):
    if c.get("values") is not None:
        return mm.plan_value_scan(c["elem_bytes"], c["values"])
    return mm.plan_relative(c["elem_bytes"], c["keyword"], c["wildcard"], c.get("char_seq"))


def _scan_both(eng, plan, **kw):
    """auto engine (filter + resolvers), the sequential engine and the dense engine must agree."""
    eng.set_engine(0)
    fast = eng.scan(plan, **kw)
    eng.set_engine(1)
    seq = eng.scan(plan, **kw)
    eng.set_engine(2)
    dense = eng.scan(plan, **kw)
    eng.set_engine(0)
    assert fast.tolist() == seq.tolist()
    assert dense.tolist() == seq.tolist()
    return fast


@pytest.mark.parametrize("case", load_golden("kat_matcher.json"), ids=lambda c: c["name"])
def test_matcher_known_answers(mm, gpu_engine, case):
    dt = np.uint8 if case["elem_bytes"] == 1 else "<u2"
    gpu_engine.upload(np.array(case["data"], dtype=dt))
    got = _scan_both(gpu_engine, _mm_plan(mm, case))
    assert got.tolist() == case["expect"]


@pytest.mark.parametrize("case", load_golden("kat_engine.json"), ids=lambda c: c["name"])
def test_engine_known_answers(mm, gpu_engine, case):
    gpu_engine.upload(np.array(case["file"], dtype=np.uint8))
    plan = _mm_plan(mm, case)
    for bs in case["block_sizes"]:
        got = _scan_both(gpu_engine, plan, block_bytes=bs, big_endian=case["big_endian"])
        assert got.tolist() == case["expect"], bs


def test_reference_vectors_matcher(mm, gpu_engine):
    cases = load_golden("diff_search.json")
    for i, c in enumerate(cases):
        dt = np.uint8 if c["elem_bytes"] == 1 else "<u2"
        gpu_engine.upload(np.array(c["data"], dtype=dt))
        plan = _mm_plan(mm, c)
        got = _scan_both(gpu_engine, plan) if i % 8 == 0 else gpu_engine.scan(plan)
        assert got.tolist() == c["expect"], c


def test_reference_vectors_engine(mm, gpu_engine):
    cases = load_golden("diff_engine.json")
    for i, c in enumerate(cases):
        gpu_engine.upload(np.array(c["file"], dtype=np.uint8))
        plan = _mm_plan(mm, c)
        kw = dict(block_bytes=c["block_size"], big_endian=c["big_endian"])
        got = _scan_both(gpu_engine, plan, **kw) if i % 8 == 0 else gpu_engine.scan(plan, **kw)
        assert got.tolist() == c["expect"], c


def test_device_generator_matches_oracle(mm, gpu_engine, oracle):
    n = (1 << 20) + 13
    gpu_engine.alloc(n)
    gpu_engine.synth(42, 4096)
    got = gpu_engine.download(0, n)
    assert (got == oracle.synth(4096, n, 42)).all()
    assert (got == mm.synth.splitmix_bytes(42, 4096, n)).all()


def _spec_case(mm, oracle, eng, nbytes, keyword, elem, wildcard=None, be=False, block=524288, whole=False, **kw):
    spec = mm.synth.RomSpec(42, nbytes, keyword, elem, wildcard, be, block if not whole else 524288, **kw)
    eng.alloc(nbytes)
    spec.apply_device(eng)
    rom = eng.download(0, nbytes)
    assert (rom == spec.host_rom()).all()                    # device edits == host edits
    wc = wildcard if wildcard is not None else 0
    oplan = oracle.plan(elem, keyword, wc)
    plan = mm.plan_relative(elem, keyword, wc)
    if whole:
        data = rom[: (nbytes // elem) * elem].view(np.uint8 if elem == 1 else "<u2")
        want = oracle.search(oplan, data)
        got = eng.scan(plan)
    else:
        want = oracle.engine(oplan, rom, block, be)
        got = eng.scan(plan, block_bytes=block, big_endian=be)
    assert got.tolist() == want.tolist()
    ctr = eng.counters()
    eng.set_engine(2)                                         # the dense engine on the same ROM
    if whole:
        dense = eng.scan(plan)
    else:
        dense = eng.scan(plan, block_bytes=block, big_endian=be)
    eng.set_engine(0)
    assert dense.tolist() == want.tolist()
    return got, ctr


def test_c2_shape_16mib(mm, gpu_engine, oracle):
    got, ctr = _spec_case(mm, oracle, gpu_engine, 16 << 20, "relativesrch", 1)
    assert len(got) >= 8 and ctr["path"] != 1


def test_c3_shape_wildcards_16mib(mm, gpu_engine, oracle):
    got, ctr = _spec_case(mm, oracle, gpu_engine, 16 << 20, "re*ative*ear*hxy", 1, wildcard=ord("*"))
    assert len(got) >= 8 and ctr["path"] != 1


@pytest.mark.parametrize("be", [False, True])
def test_c4_shape_16bit_32mib(mm, gpu_engine, oracle, be):
    got, ctr = _spec_case(mm, oracle, gpu_engine, 32 << 20, "textsrch", 2, be=be)
    assert len(got) >= 8 and ctr["path"] != 1


@pytest.mark.parametrize("kw", ["monkey", "abcde", "relativesrch"])
def test_c1_shape_whole_buffer_16mib(mm, gpu_engine, oracle, kw):
    _spec_case(mm, oracle, gpu_engine, 16 << 20, kw, 1, whole=True)


def test_whole_buffer_16bit(mm, gpu_engine, oracle):
    _spec_case(mm, oracle, gpu_engine, 8 << 20, "textsrch", 2, whole=True)


@pytest.mark.parametrize("block", [4096, 8191, 65536, 8388608])
def test_block_sizes(mm, gpu_engine, oracle, block):
    _spec_case(mm, oracle, gpu_engine, (4 << 20) + 77, "relativesrch", 1, block=block)
    _spec_case(mm, oracle, gpu_engine, (4 << 20) + 77, "textsrch", 2, block=block)


def test_ragged_and_tiny_inputs(mm, gpu_engine, oracle):
    plan = mm.plan_relative(1, "text")
    oplan = oracle.plan(1, "text")
    rng = np.random.default_rng(3)
    for n in [0, 1, 3, 4, 5, 15, 16, 17, 31, 33, 255, 4097]:
        data = rng.integers(0, 4, n).astype(np.uint8) + 100
        gpu_engine.upload(data)
        assert _scan_both(gpu_engine, plan).tolist() == oracle.search(oplan, data).tolist(), n
        for bs in (1, 2, 7, 16, 1000):
            want = oracle.engine(oplan, data, bs).tolist() if n else []
            assert _scan_both(gpu_engine, plan, block_bytes=bs).tolist() == want, (n, bs)


def test_dense_matches_constant_data(mm, gpu_engine, oracle):
    # 'aaa' on constant bytes matches every L-1 positions: result volume and
    # the dense fallback
    n = 1 << 20
    data = np.full(n, 7, np.uint8)
    gpu_engine.upload(data)
    plan, oplan = mm.plan_relative(1, "aaa"), oracle.plan(1, "aaa")
    got = gpu_engine.scan(plan, block_bytes=4096)
    assert got.tolist() == oracle.engine(oplan, data, 4096).tolist()
    assert len(got) > n // 3
    assert gpu_engine.counters()["path"] == 3              # too dense for per-candidate work: dense engine
    small = data[:20000]
    gpu_engine.upload(small)
    assert gpu_engine.scan(plan).tolist() == oracle.search(oplan, small).tolist()


def test_low_entropy_alphabets(mm, gpu_engine, oracle):
    # small alphabets force unsafe skips, overlaps and long non-coalescing chains
    rng = np.random.default_rng(11)
    for k, kw in ((2, "abab"), (3, "abcab"), (4, "monkey"), (3, "a*b*a")):
        data = (rng.integers(0, k, 1 << 18) + 97).astype(np.uint8)
        gpu_engine.upload(data)
        wc = ord("*") if "*" in kw else 0
        plan, oplan = mm.plan_relative(1, kw, wc), oracle.plan(1, kw, wc)
        assert gpu_engine.scan(plan, block_bytes=8192).tolist() == oracle.engine(oplan, data, 8192).tolist()
        assert gpu_engine.scan(plan).tolist() == oracle.search(oplan, data).tolist()


def test_base_offset_and_capacity(mm, gpu_engine, oracle):
    spec = mm.synth.RomSpec(7, 2 << 20, "relativesrch", 1)
    rom = spec.host_rom()
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(1, "relativesrch"), oracle.plan(1, "relativesrch")
    want = oracle.engine(oplan, rom, 524288)
    got = gpu_engine.scan(plan, block_bytes=524288, base_offset=1 << 40, cap=1)   # forces the CAPACITY retry
    assert (got - np.uint64(1 << 40)).tolist() == want.tolist()
