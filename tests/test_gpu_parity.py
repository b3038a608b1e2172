# SPDX-License-Identifier: GPL-3.0-or-later
"""Parity of the HIP path (through the C ABI) with the oracle and the golden
vectors.  Needs a real MI355X: run with `pytest -m gpu`.  Offsets are compared
bit-exactly (integer work: no tolerance)."""
import os

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


def test_tiny_reference_vectors(mm, gpu_engine):
    """all 11 000 tiny vectors of the compiled reference through the C ABI (every 16th
    also through the sequential and the forward engine)."""
    from conftest import load_tiny
    search, engine = load_tiny()
    assert len(search) + len(engine) >= 10000
    for i, c in enumerate(search):
        gpu_engine.upload(c["data"])
        plan = _mm_plan(mm, c)
        got = _scan_both(gpu_engine, plan) if i % 16 == 0 else gpu_engine.scan(plan)
        assert got.tolist() == c["expect"], {k: v for k, v in c.items() if k != "data"}
    for i, c in enumerate(engine):
        gpu_engine.upload(c["file"])
        plan = _mm_plan(mm, c)
        kw = dict(block_bytes=c["block_size"], big_endian=c["big_endian"])
        got = _scan_both(gpu_engine, plan, **kw) if i % 16 == 0 else gpu_engine.scan(plan, **kw)
        assert got.tolist() == c["expect"], {k: v for k, v in c.items() if k != "file"}


def test_zero_copy_upload_boundary(mm, gpu_engine, oracle):
    """Uploads of up to 512 KiB are scanned in place from pinned host memory (mmh_rom_upload), larger ones
    from HBM: the same results on both sides of the limit, and the ROM reads back."""
    rng = np.random.default_rng(12)
    kw = "monkey"
    for n in ((512 << 10) - 1, 512 << 10, (512 << 10) + 1, 4099):
        rom = _random_rom_with_plants(rng, n, 1, [ord(c) for c in kw], False, nplants=30)
        gpu_engine.upload(rom)
        assert (gpu_engine.download(0, n) == rom).all()
        plan, oplan = mm.plan_relative(1, kw), oracle.plan(1, kw)
        assert _scan_both(gpu_engine, plan, block_bytes=65536).tolist() == oracle.engine(oplan, rom, 65536).tolist()
        assert gpu_engine.scan(plan).tolist() == oracle.search(oplan, rom).tolist()
        gpu_engine.poke(100, np.array([1, 2, 3], np.uint8))                   # poke works on either kind of ROM
        assert gpu_engine.download(100, 3).tolist() == [1, 2, 3]
        assert gpu_engine.gather([100, n - 2], 3).tolist() == [[1, 2, 3], [int(rom[n - 2]), int(rom[n - 1]), 0]]


def test_no_matches_in_the_padding_behind_the_rom(mm, gpu_engine, oracle):
    """Regression (found by the tiny vectors): the streaming code looks at whole 16-byte chunks, so it
    sees the bytes behind the ROM; when those continue the pattern (stale bytes of a larger ROM that
    was resident before, or a borrowed buffer) a survivor there must not be taken for an alignment
    of a block that does not exist.  Sizes that are exact multiples of the block size are the trap."""
    for elem, kw, n, block in ((1, "eee", 7 * 76, 7), (1, "ddddddddd", 129 * 7, 129), (2, "uuu", 26 * 36, 26), (1, "nnnn", 4096, 512),
                               (1, "relativesrch", 65536, 4096)):
        big = np.full(n + 4096, 0x41, np.uint8)
        if kw == "relativesrch":
            vals = np.array([ord(c) for c in kw], np.uint8)
            for at in range(100, n + 4000, 1000):             # plants inside AND behind the ROM
                big[at:at + len(kw)] = vals
        gpu_engine.upload(big)                                 # leaves its bytes in the device buffer ...
        gpu_engine.upload(big[:n])                             # ... behind the ROM that counts
        plan = mm.plan_relative(elem, kw)
        want = oracle.engine(oracle.plan(elem, kw), big[:n], block)
        got = _scan_both(gpu_engine, plan, block_bytes=block)
        assert got.tolist() == want.tolist(), (kw, n, block)
        assert len(got) == 0 or int(got.max()) + len(kw) * elem <= n


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


def test_many_matches_stay_on_the_resolver_path(mm, gpu_engine, oracle):
    # tens of thousands of candidates: still filter + per-candidate resolvers (not the forward
    # engine), ordered by the device radix sort instead of the rank kernels
    rng = np.random.default_rng(5)
    kw = "words"
    rom = _random_rom_with_plants(rng, 32 << 20, 1, [ord(c) for c in kw], False, nplants=40000)
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(1, kw), oracle.plan(1, kw)
    got = gpu_engine.scan(plan, block_bytes=524288, cap=1 << 17)
    c = gpu_engine.counters()
    assert c["path"] in (0, 2) and c["candidates"] > 30000
    want = oracle.engine(oplan, rom, 524288)
    assert got.tolist() == want.tolist() and len(want) > 16384
    assert gpu_engine.scan(plan, cap=1 << 17).tolist() == oracle.search(oplan, rom).tolist()
    assert gpu_engine.scan(plan, block_bytes=524288, cap=1000).tolist() == want.tolist()   # MMH_E_CAPACITY, retried bigger


@pytest.mark.parametrize("nplants,elem", [(9000, 1), (70000, 1), (150000, 1), (50000, 2), (700000, 1)])
def test_bucketed_store_list_lengths(mm, gpu_engine, oracle, nplants, elem):
    """Big ROMs drop candidates into buckets of their ROM neighbourhood and mm_scan_tail2 ranks them from there
    (csrc/mm_tail2.h): lists of up to 8192 slots are written straight into pinned memory, longer ones (up to 2^20) are
    fetched from the device-side copy -- through mmh_scan, through the submit lanes, and one chain over the whole
    buffer; a few plants wrap around (candidates that are no matches: holes)."""
    rng = np.random.default_rng(nplants)
    kw = "bucket"
    rom = _random_rom_with_plants(rng, 64 << 20, elem, [ord(c) for c in kw], False, nplants=nplants)
    if elem == 1:
        # plants whose bytes wrap around 255 -> 0: SWAR survivors the signed compare loop rejects
        for pos in rng.integers(1000, (64 << 20) - 1000, 300):
            rom[pos:pos + 6] = (np.array([ord(c) for c in kw]) + 150) & 0xFF
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(elem, kw), oracle.plan(elem, kw)
    want = oracle.engine(oplan, rom, 524288)
    got = gpu_engine.scan(plan, block_bytes=524288, cap=1 << 20)
    ctr = gpu_engine.counters()
    assert got.tolist() == want.tolist() and len(want) > 0.8 * nplants
    assert ctr["path"] in (0, 2) and ctr["candidates"] >= len(want), ctr
    tickets = [gpu_engine.submit(plan, block_bytes=524288) for _ in range(3)]
    for t in tickets:
        assert gpu_engine.collect(t, cap=1 << 20).tolist() == want.tolist()
    data = rom if elem == 1 else rom.view("<u2")
    assert gpu_engine.scan(plan, cap=1 << 20).tolist() == oracle.search(oplan, data).tolist()
    assert gpu_engine.scan(plan, block_bytes=524288, cap=100).tolist() == want.tolist()      # MMH_E_CAPACITY, retried bigger


def test_crowded_bucket_takes_the_list_based_kernels(mm, gpu_engine, oracle):
    """More candidates in one ROM neighbourhood than a bucket holds (4096 per 128 KiB of a 512 MiB ROM: a 3-symbol keyword
    planted every 16 bytes over 256 KiB): mm_scan_tail2 resolves nothing, the scan starts over with the list-based
    kernels and still reports exactly the reference's offsets -- through mmh_scan and through the submit lanes."""
    from _oracle import oracle_engine_parallel
    rng = np.random.default_rng(4096)
    n = 512 << 20
    rom = rng.integers(0, 256, n, dtype=np.uint8)
    at = (300 << 20) + 776          # (even: with this keyword every jump of the reference's chain is 2 -- odd positions are never visited)
    for k in range(16384):
        base = int(rng.integers(0, 200))
        rom[at + 16 * k: at + 16 * k + 3] = [base + 10, base + 11, base + 12]          # 'klm' shifted: deltas +1, +1
    kw = "klm"
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(1, kw), oracle.plan(1, kw)
    want = oracle_engine_parallel(oracle, oplan, rom, 524288)
    got = gpu_engine.scan(plan, block_bytes=524288, cap=1 << 17)
    ctr = gpu_engine.counters()
    assert got.tolist() == want.tolist() and len(want) > 16384
    assert ctr["path"] in (0, 2, 4, 5), ctr
    t = gpu_engine.submit(plan, block_bytes=524288)
    assert gpu_engine.collect(t, cap=1 << 17).tolist() == want.tolist()


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


def _random_rom_with_plants(rng, n, elem, kw_vals, be, nplants=40):
    hi = 256 if elem == 1 else 65536
    d = rng.integers(0, hi, n // elem).astype(np.int64)
    L = len(kw_vals)
    lits = [v for v in kw_vals if v is not None]
    for _ in range(nplants):
        pos = int(rng.integers(0, d.size - L))
        sh = int(rng.integers(-min(lits), hi - max(lits)))
        for j, v in enumerate(kw_vals):
            if v is not None:
                d[pos + j] = v + sh
    arr = d.astype(np.uint8 if elem == 1 else (">u2" if be else "<u2"))
    return arr.view(np.uint8)


@pytest.mark.parametrize("L", [2, 3, 17, 24, 32])
def test_keyword_lengths(mm, gpu_engine, oracle, L):
    # L-1 > 16 takes the byte-packed phase maps, L = 2 the degenerate single phase
    rng = np.random.default_rng(100 + L)
    kw = [int(c) for c in rng.integers(97, 123, L)]
    rom = _random_rom_with_plants(rng, 4 << 20, 1, kw, False)
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(1, kw), oracle.plan(1, kw)
    want = oracle.engine(oplan, rom, 524288)
    assert _scan_both(gpu_engine, plan, block_bytes=524288).tolist() == want.tolist()
    assert len(want) >= 20 or L == 2
    assert gpu_engine.scan(plan).tolist() == oracle.search(oplan, rom).tolist()
    with pytest.raises(mm.MMError):
        mm.plan_relative(1, [97 + (i % 5) for i in range(129)])   # longer than MMH_MAX_KEYWORD: refused, loudly


@pytest.mark.parametrize("L", [33, 48, 64, 65, 127, 128])
@pytest.mark.parametrize("path", ["simple", "wildcard", "mixed-case"])
@pytest.mark.parametrize("elem,be", [(1, False), (2, True)])
def test_long_keywords(mm, gpu_engine, oracle, L, path, elem, be):
    """Keywords beyond 32 symbols, up to the 128 the reference's char-sized tables allow (monkey_moore.cpp:250-253).  Up to
    64 symbols the per-candidate resolvers take them with a phase set of one 64-bit ballot (D = L - 1 <= 63); beyond that
    (round 6) the first resolver follows two phases per lane and what its look-back windows leave open goes to the forward
    engine (phase maps of 128 bytes, lane e and lane e + 64).  Both reference loops, both element sizes, engine and
    whole-buffer semantics, the forward engine forced on every one of them as well, against the oracle."""
    rng = np.random.default_rng(1000 + L + 7 * elem)
    wildcard = 0
    if path == "simple":
        kw = [int(c) for c in rng.integers(97, 123, L)]
        vals = list(kw)
    elif path == "wildcard":
        wildcard = ord("*")
        kw = [int(c) for c in rng.integers(97, 123, L)]
        for i in rng.choice(np.arange(1, L - 1), size=max(2, L // 9), replace=False):
            kw[int(i)] = wildcard
        vals = [None if c == wildcard else c for c in kw]
    else:
        wildcard = ord("*")
        kw = [int(c) for c in rng.integers(97, 123, L)]
        for i in rng.choice(np.arange(0, L), size=max(2, L // 7), replace=False):
            kw[int(i)] -= 32                                  # upper case: the minority case becomes wildcards
        vals = [None if c < 97 else c for c in kw]
    n = (2 << 20) + 4099
    rom = _random_rom_with_plants(rng, n, elem, vals, be, nplants=60)
    # low-entropy stretches: short jumps, chains that do not merge
    rom[300000:400000] = rng.integers(0, 3, 100000).astype(np.uint8) + 0x40
    rom[700000:760000] = 0x41
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(elem, kw, wildcard), oracle.plan(elem, kw, wildcard)
    for block in (524288, 65536 + 2 * elem):
        want = oracle.engine(oplan, rom, block, be)
        got = gpu_engine.scan(plan, block_bytes=block, big_endian=be)
        # (the candidate path -- 2: its hard resolver, 3 / 4 / 5: the low-entropy stretches to the forward engine)
        assert gpu_engine.counters()["path"] in (0, 2, 3, 4, 5), gpu_engine.counters()
        assert got.tolist() == want.tolist(), (L, path, elem, block)
        if True:
            gpu_engine.set_engine(2)                           # the forward engine's wide maps on the same keyword
            assert gpu_engine.scan(plan, block_bytes=block, big_endian=be).tolist() == want.tolist(), (L, path, elem, block, "forward")
            gpu_engine.set_engine(0)
            whole = gpu_engine.scan(plan)                      # one chain over the whole buffer (elements in little-endian order)
            assert whole.tolist() == oracle.search(oplan, rom if elem == 1 else rom[: n // 2 * 2].view("<u2")).tolist()
        if block == 524288:
            # (of the 60 plants the reference itself reports few when wildcards cap its skips: its chain
            # walks past most of them -- SURVEY fact 1; what matters is that the lists are identical)
            assert len(want) >= (10 if path == "simple" else 1)
    gpu_engine.set_engine(1)                                   # and the sequential engine agrees
    assert gpu_engine.scan(plan, block_bytes=524288, big_endian=be).tolist() == oracle.engine(oplan, rom, 524288, be).tolist()
    gpu_engine.set_engine(0)
    if not be:
        data = rom[: (n // elem) * elem].view(np.uint8 if elem == 1 else "<u2")
        assert gpu_engine.scan(plan).tolist() == oracle.search(oplan, data).tolist()


def test_pattern_without_swar_key_uses_dense_engine(mm, gpu_engine, oracle):
    # no literal with a literal one to four places to its left -> nothing for the streaming filter to key on
    # (runs of two and three wildcards have the wide shapes since round 6: `a**d**g` is in WIDE_KEYWORDS' family below)
    rng = np.random.default_rng(7)
    kw = "a****f****k"
    vals = [None if ch == "*" else ord(ch) for ch in kw]
    rom = _random_rom_with_plants(rng, 2 << 20, 1, vals, False)
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(1, kw, ord("*")), oracle.plan(1, kw, ord("*"))
    got = gpu_engine.scan(plan, block_bytes=65536)
    assert gpu_engine.counters()["path"] == 3
    assert got.tolist() == oracle.engine(oplan, rom, 65536).tolist()
    assert len(got) >= 10


def _wildcard_shapes(L):
    # every placement of wildcards over an L-symbol keyword that leaves >= 2 literals
    out = []
    for mask in range(1 << L):
        kw = "".join("*" if (mask >> i) & 1 else "abcdefghijkl"[(i * 5) % 12] for i in range(L))
        if L - bin(mask).count("1") >= 2:
            out.append(kw)
    return out


# runs of two and three wildcards (and longer ones, and every pair of gaps of the wide 8-bit shapes): `But**er` is the
# reference's own (tests/test_monkey_moore.cpp:194-221; mixed case: the capital becomes a wildcard as well)
WIDE_KEYWORDS = ["qz**mb", "q**k**x", "bu***er", "abc**de", "ab*c*d", "a***b***c", "ab***cd**ef", "a**bc***d*e", "a****bc", "a*b**c***d",
                 "ab**c*d", "a***bc**d", "a**b*c", "a***b**c", "a***b*c", "a**b***c", "a*b***c", "ab***c", "a***bc", "a**b****c*d"]


@pytest.mark.parametrize("elem,be", [(1, False), (2, False), (2, True)])
def test_every_wildcard_placement(mm, gpu_engine, oracle, elem, be):
    # Sweeps the SWAR condition shapes of the streaming filter (adjacent and over-a-wildcard
    # deltas, run-time shifts, 1..4 conditions; choose_filter in mm_kernels.hip) through the
    # span kernel: ROMs of whole 4 KiB groups plus a ragged tail, matches planted all over.
    rng = np.random.default_rng(2024 + elem + int(be))
    paths = set()
    for n, kw in enumerate(_wildcard_shapes(6) + ["ab*de*gh", "abc*efgh*jkl", "a*c*e*g", "*b*d*f*h", "ab**ef*hi"] + WIDE_KEYWORDS):
        vals = [None if ch == "*" else ord(ch) for ch in kw]
        rom = _random_rom_with_plants(rng, (96 << 10) + 4 * (n % 7) + 2, elem, vals, be)
        gpu_engine.upload(rom)
        plan, oplan = mm.plan_relative(elem, kw, ord("*")), oracle.plan(elem, kw, ord("*"))
        got = gpu_engine.scan(plan, block_bytes=16384, big_endian=be)
        paths.add(gpu_engine.counters()["path"])
        want = oracle.engine(oplan, rom, 16384, big_endian=be)
        assert got.tolist() == want.tolist(), kw
        assert len(want) >= 3, kw
        # whole-buffer mode reads the bytes as little-endian elements whatever they were written as
        assert gpu_engine.scan(plan).tolist() == oracle.search(oplan, rom if elem == 1 else rom.view("<u2")).tolist(), kw
    assert 0 in paths or 2 in paths


def test_hard_candidates(mm, gpu_engine, oracle):
    # 'abcde' never leaves its residue class (every jump is 4): no look-back window
    # can certify a candidate, so they all go through mm_hard_resolve ...
    rng = np.random.default_rng(9)
    kw = [ord(c) for c in "abcde"]
    rom = _random_rom_with_plants(rng, 8 << 20, 1, kw, False, nplants=24)
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(1, "abcde"), oracle.plan(1, "abcde")
    got = gpu_engine.scan(plan, block_bytes=4 << 20)
    assert gpu_engine.counters()["path"] == 2
    assert got.tolist() == oracle.engine(oplan, rom, 4 << 20).tolist()
    # ... and when a prefix is longer than it maps (whole-buffer chain over 32 MiB), through the dense engine
    rom = _random_rom_with_plants(rng, 32 << 20, 1, kw, False, nplants=24)
    gpu_engine.upload(rom)
    got = gpu_engine.scan(plan)
    assert gpu_engine.counters()["path"] == 3
    assert got.tolist() == oracle.search(oplan, rom).tolist()
    assert len(got) >= 4


def test_value_scan_and_custom_sequence_at_scale(mm, gpu_engine, oracle):
    rng = np.random.default_rng(21)
    values = [60, 61, 62, 63, 64, 71, 40]
    rom = _random_rom_with_plants(rng, 4 << 20, 2, values, True)
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_value_scan(2, values), oracle.plan_values(2, values)
    assert _scan_both(gpu_engine, plan, block_bytes=524288, big_endian=True).tolist() == oracle.engine(oplan, rom, 524288, True).tolist()
    seq = "aiueobcdfghjklmnpqrstvwxyz"
    kw = "matchbox"
    vals = [seq.index(c) for c in kw]
    rom = _random_rom_with_plants(rng, 4 << 20, 1, vals, False)
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(1, kw, 0, seq), oracle.plan(1, kw, 0, seq)
    want = oracle.engine(oplan, rom, 524288)
    assert _scan_both(gpu_engine, plan, block_bytes=524288).tolist() == want.tolist()
    assert len(want) >= 20


@pytest.mark.parametrize("elem", [1, 2])
def test_c1_reference_benchmark_input_vs_compiled_reference(mm, gpu_engine, elem):
    """BASELINE config C1: the reference's own benchmark (bench_search.cpp) -- mt19937(42)
    data, keyword abcde and its three wildcard variants -- GPU whole-buffer scan against
    MonkeyMoore<T>::search of the COMPILED REFERENCE (oracle/_ref, prebuilt) on the same bytes."""
    from _oracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libmmref.so was not shipped")
    ref = Ref()
    n = 16 << 20
    raw = ref.bench_data(elem, n)
    data = raw.view(np.uint8 if elem == 1 else "<u2")
    gpu_engine.upload(raw)
    total = 0
    for kw, wc in (("abcde", 0), ("*bcde", ord("*")), ("ab*de", ord("*")), ("abcd*", ord("*")), ("monkey", 0)):
        want = ref.search(elem, kw, data, wc)
        got = gpu_engine.scan(mm.plan_relative(elem, kw, wc))
        assert got.tolist() == want.tolist(), kw
        total += len(want)
    assert total == (2 if elem == 1 else 0)   # what the reference finds in its own benchmark buffer


def test_rom_from_file_and_gather(mm, gpu_engine, oracle, tmp_path):
    # mmh_rom_load_file (parallel readers, overlapped copies) against the bytes of the file, at
    # offsets and sizes that are not multiples of the 4 MiB staging piece; mmh_rom_gather against numpy
    rng = np.random.default_rng(77)
    kw = "relativesrch"
    data = _random_rom_with_plants(rng, (21 << 20) + 12345, 1, [ord(c) for c in kw], False, nplants=200)
    path = tmp_path / "rom.bin"
    data.tofile(path)
    for off, n, threads in [(0, data.size, 0), (4099, (9 << 20) + 7, 3), (data.size - 100, 100, 1), (5, 0, 2)]:
        stats = gpu_engine.load_file(str(path), off, n, threads)
        assert stats["bytes"] == n
        if n:
            assert gpu_engine.download(0, n).tobytes() == data[off:off + n].tobytes()
    gpu_engine.load_file(str(path), 0, data.size)
    plan, oplan = mm.plan_relative(1, kw), oracle.plan(1, kw)
    got = gpu_engine.scan(plan, block_bytes=524288)
    assert got.tolist() == oracle.engine(oplan, data, 524288).tolist() and len(got) >= 150
    under = gpu_engine.gather(got, len(kw))
    assert under.tolist() == [data[o:o + len(kw)].tolist() for o in got.tolist()]
    tail = gpu_engine.gather([data.size - 3], 8)           # bytes behind the ROM read as zero
    assert tail[0].tolist() == data[-3:].tolist() + [0] * 5
    with pytest.raises(mm.MMError):
        gpu_engine.load_file(str(tmp_path / "missing.bin"), 0, 16)
    with pytest.raises(mm.MMError):
        gpu_engine.load_file(str(path), data.size - 10, 100)  # runs off the end of the file: short read
    # the watched load (mmh_rom_load_file_watched): bytes landed, and an abort word that is already up
    import ctypes
    done, stop = ctypes.c_uint64(123), ctypes.c_int32(0)
    gpu_engine.load_file(str(path), 0, data.size, 4, abort_word=stop, bytes_done=done)
    assert done.value == data.size
    assert gpu_engine.scan(plan, block_bytes=524288).tolist() == got.tolist()
    stop.value = 1
    with pytest.raises(mm.MMError) as err:
        gpu_engine.load_file(str(path), 0, data.size, 4, abort_word=stop, bytes_done=done)
    assert err.value.code == mm.MMH_E_ABORTED and done.value < data.size
    # ... after which the context loads and scans as before (copies the aborted load left in flight are drained first)
    gpu_engine.load_file(str(path), 0, data.size)
    assert gpu_engine.scan(plan, block_bytes=524288).tolist() == got.tolist()


@pytest.mark.parametrize("depth", [2, 3])
def test_scans_in_flight(mm, gpu_engine, oracle, depth):
    # mmh_scan_submit / mmh_scan_collect: same offsets as mmh_scan, `depth` tickets outstanding at a
    # time, different plans interleaved on the same ROM; the paths the lanes do not run themselves
    # (no SWAR key -> dense engine, long match lists) come back through the synchronous rescan
    rng = np.random.default_rng(31)
    kws = [("relativesrch", 0), ("re*at*vesrch", ord("*")), ("a**d**g", ord("*")), ("words", 0)]
    rom = rng.integers(0, 256, 24 << 20).astype(np.uint8)
    for n, (kw, wc) in enumerate(kws):
        vals = [None if ch == "*" else ord(ch) for ch in kw]
        planted = _random_rom_with_plants(rng, 1 << 20, 1, vals, False, nplants=300 if kw != "words" else 20000)
        at = (5 * n + 2) << 20
        rom[at:at + planted.size] = planted
    gpu_engine.upload(rom)
    plans = [mm.plan_relative(1, kw, wc) for kw, wc in kws]
    want = [gpu_engine.scan(p, block_bytes=524288, cap=1 << 16).tolist() for p in plans]
    assert want[0] == oracle.engine(oracle.plan(1, kws[0][0]), rom, 524288).tolist()
    # (more than MM_DIRECT_PUBLISH slots, the device to itself: the list comes over through mm_publish_list + a polled word)
    assert want[3] == oracle.engine(oracle.plan(1, kws[3][0]), rom, 524288).tolist()
    assert len(want[3]) > 16384 and all(len(w) > 20 for w in want), [len(w) for w in want]
    order = [0, 1, 2, 3, 3, 0, 2, 1, 0, 0, 1, 0, 0, 0, 1, 1, 2]
    tickets, got = [], []
    for k in order:
        tickets.append((k, gpu_engine.submit(plans[k], block_bytes=524288)))
        if len(tickets) == depth:
            i, t = tickets.pop(0)
            got.append((i, gpu_engine.collect(t, cap=1000).tolist()))      # cap too small now and then: retried
    while len(tickets) < mm.MMH_MAX_IN_FLIGHT:
        tickets.append((0, gpu_engine.submit(plans[0], block_bytes=524288)))
        order = order + [0]
    with pytest.raises(mm.MMError):
        gpu_engine.submit(plans[0], block_bytes=524288)                    # one more outstanding scan is refused
    while tickets:
        i, t = tickets.pop(0)
        got.append((i, gpu_engine.collect(t).tolist()))
    with pytest.raises(mm.MMError):
        gpu_engine.collect(12345)
    assert [i for i, _ in got] == order
    for i, offs in got:
        assert offs == want[i], kws[i]


@pytest.mark.parametrize("mib", [2, 16, 80])
def test_scans_without_timing_events(mm, gpu_engine, oracle, mib):
    """mmh_set_timing(0): no start event on a scan's first dispatch (~4.5 us less per scan -- what the include/mmoore facade runs
    with).  Same offsets through the single-launch kernel (2 MiB), the streaming + tail kernels (16 / 80 MiB) and the lanes; the
    timing calls then report the kernel's own clock for single-launch scans and 0 otherwise, and everything again once switched
    back on."""
    rng = np.random.default_rng(500 + mib)
    rom = _random_rom_with_plants(rng, mib << 20, 1, [ord(ch) for ch in "relativesrch"], False, nplants=40 * mib)
    gpu_engine.upload(rom)
    plan = mm.plan_relative(1, "relativesrch")
    want = oracle.engine(oracle.plan(1, "relativesrch"), rom, 524288).tolist()
    assert len(want) >= 30 * mib                               # (a few plants overwrite each other)
    assert gpu_engine.scan(plan, block_bytes=524288).tolist() == want
    timed = gpu_engine.timings()
    assert timed["total_ms"] > 0 and timed["filter_ms"] > 0
    gpu_engine.set_timing(False)
    try:
        for _ in range(3):
            assert gpu_engine.scan(plan, block_bytes=524288).tolist() == want
        t = gpu_engine.timings()
        if mib <= 4:                                           # one launch: the kernel's own stamps
            assert 0 < t["filter_ms"] <= t["total_ms"], t
        else:
            assert t["total_ms"] == 0 and t["filter_ms"] == 0, t
        tickets = [gpu_engine.submit(plan, block_bytes=524288) for _ in range(3)]
        for tk in tickets:
            assert gpu_engine.collect(tk).tolist() == want
    finally:
        gpu_engine.set_timing(True)
    assert gpu_engine.scan(plan, block_bytes=524288).tolist() == want
    t = gpu_engine.timings()
    assert t["total_ms"] > 0 and t["filter_ms"] > 0, t


def test_offset_gather_on_rccl_world_of_one():
    # The collective plumbing bench.py uses at N > 1 (pinned staging, all_gather_into_tensor on
    # RCCL, one device-to-host copy), exercised on the one GPU this suite has.  In its own
    # process, torch first -- as in bench.py: torch brings its own HIP runtime, which finds no
    # device once the system runtime behind libmmoore_hip.so has initialised the GPU.
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_gather_check.py")], capture_output=True, text=True,
                       timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "gather ok" in r.stdout


def test_synthetic_rom_goldens_on_device(mm, gpu_engine):
    # ROMs generated ON THE DEVICE (mm_synth_fill + pokes) hash to the golden file's
    # SHA-256 and scan to the offsets the compiled reference reported for them
    import hashlib
    from _oracle import Ref
    for c in load_golden("synth_roms.json"):
        if c.get("bench_data"):
            # C1: the reference benchmark's own buffer (its generator lives in the compiled reference shim)
            if not Ref.available():
                continue
            data = Ref().bench_data(c["elem_bytes"], c["nbytes"])
            assert hashlib.sha256(data.tobytes()).hexdigest() == c["sha256"], c["name"]
            gpu_engine.upload(data)
            for kw, want in c["search"].items():
                assert gpu_engine.scan(mm.plan_relative(c["elem_bytes"], kw)).tolist() == want, (c["name"], kw)
            continue
        spec = mm.synth.RomSpec(c["seed"], c["nbytes"], c["keyword"], c["elem_bytes"], c["wildcard"], c["big_endian"], 524288)
        gpu_engine.alloc(c["nbytes"])
        spec.apply_device(gpu_engine)
        assert hashlib.sha256(gpu_engine.download(0, c["nbytes"]).tobytes()).hexdigest() == c["sha256"], c["name"]
        plan = mm.plan_relative(c["elem_bytes"], c["keyword"], c["wildcard"] or 0)
        for block, want in c["engine"].items():
            got = _scan_both(gpu_engine, plan, block_bytes=int(block), big_endian=c["big_endian"])
            assert got.tolist() == want, (c["name"], block)
        if "whole_buffer" in c:
            assert gpu_engine.scan(plan).tolist() == c["whole_buffer"], c["name"]


@pytest.mark.parametrize("elem", [1, 2])
def test_flood_of_undecidable_candidates_uses_flagged_domains(mm, gpu_engine, oracle, elem):
    # 'abcde' candidates can only be settled from the start of their domain (test_hard_candidates).
    # More of them than the hard list takes: the forward engine then runs on the flagged domains
    # only (path 4), everything else keeps the resolvers' verdicts.
    rng = np.random.default_rng(17 + elem)
    block, nblocks = 65536, 512
    vals = [ord(c) for c in "abcde"]
    rom = _random_rom_with_plants(rng, block * nblocks, elem, vals, False, nplants=150)
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(elem, "abcde"), oracle.plan(elem, "abcde")
    got = gpu_engine.scan(plan, block_bytes=block)
    ctr = gpu_engine.counters()
    want = oracle.engine(oplan, rom, block)
    assert got.tolist() == want.tolist()
    assert len(want) >= 20
    assert ctr["path"] == 4 and 100 <= ctr["tiles_walked"] <= 160, ctr      # [2] = flagged domains on this path
    # the forced engines agree, and so do two scans in flight (collect falls back to the same path)
    assert _scan_both(gpu_engine, plan, block_bytes=block).tolist() == want.tolist()
    t = gpu_engine.submit(plan, block_bytes=block)
    assert gpu_engine.collect(t).tolist() == want.tolist()


def test_flagged_domains_with_a_long_list_from_the_forward_engine(mm, oracle):
    # Found by a fuzz soak (seed 303 of test_gpu_fuzz.py): 'bbbb' on a two-symbol alphabet -- every domain is
    # flagged (path 4) and the forward engine reports tens of thousands of matches.  Packing those for the sort
    # reallocated the buffer the domain list lived in, and the engine's second attempt (its lists had to grow)
    # read the list from freed memory: 527 of 43538 matches.  A FRESH context, so that every buffer starts small.
    rng = np.random.default_rng(303)
    rom = (rng.integers(0, 2, 1 << 20) + 0x60).astype(np.uint8)
    plan, oplan = mm.plan_relative(1, "bbbb"), oracle.plan(1, "bbbb")
    for block in (65536, 524288):
        want = oracle.engine(oplan, rom, block)
        assert len(want) > 30000
        with mm.Engine(0) as eng:
            eng.upload(rom)
            got = eng.scan(plan, block_bytes=block, cap=1 << 12)
            assert eng.counters()["path"] in (3, 4, 5), eng.counters()
            assert got.tolist() == want.tolist(), block
            assert eng.scan(plan, block_bytes=block).tolist() == want.tolist()      # and again, buffers grown


@pytest.mark.parametrize("kw,elem,hard", [("abcde", 1, 0), ("aaaa", 1, 0), ("aaaa", 2, 0), ("abcde", 1, 360)])
def test_candidate_flood_in_a_padding_run(mm, gpu_engine, oracle, kw, elem, hard):
    # The bench ROM has 1 MiB runs of 0x00, 0xFF and a +1 ramp.  A keyword that matches a whole run
    # ('abcde' on the ramp, 'aaaa' on the constants) floods the candidate lists from two or three
    # domains: the forward engine takes those domains, the per-candidate path the rest (path 5).
    # hard > 0: on top of that, more undecidable candidates ('abcde' plants, test_hard_candidates)
    # than the hard list takes, in 12 more domains -- those domains join the flooded ones.
    n, block = 64 << 20, 524288
    spec = mm.synth.RomSpec(42, n, kw, elem, None, False, block, plants_per_mib=0 if kw in ("abcde", "aaaa") else 4)   # (plants of these two are undecidable one by one)
    gpu_engine.alloc(n)
    spec.apply_device(gpu_engine)
    word = np.array([ord(c) + 20 for c in kw], dtype=np.uint8 if elem == 1 else "<u2").view(np.uint8)
    for k in range(hard):
        gpu_engine.poke((40 + k % 12) * block + 3000 + 7777 * (k // 12), word)
    rom = gpu_engine.download(0, n)
    plan, oplan = mm.plan_relative(elem, kw), oracle.plan(elem, kw)
    got = gpu_engine.scan(plan, block_bytes=block, cap=1 << 21)
    ctr = gpu_engine.counters()
    want = oracle.engine(oplan, rom, block)
    assert got.tolist() == want.tolist()
    assert len(want) > 100000                                  # the run(s) matched wholesale
    lo, hi = (1, 8) if not hard else (10, 40)
    assert ctr["path"] == 5 and lo <= ctr["tiles_walked"] <= hi, ctr      # [2] = domains given to the forward engine
    gpu_engine.set_engine(2)
    assert gpu_engine.scan(plan, block_bytes=block, cap=1 << 21).tolist() == want.tolist()
    gpu_engine.set_engine(0)
    t = gpu_engine.submit(plan, block_bytes=block)
    assert gpu_engine.collect(t, cap=1 << 21).tolist() == want.tolist()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MM_SWEEP_SEEDS", "8"))))       # raise for a soak
def test_forward_engine_sweep_on_mixed_roms(mm, gpu_engine, oracle, seed):
    """The forward engine forced (mmh_set_engine 2) on ROMs that make its sparse sweep take every turn (csrc/mm_forward.h,
    pass 1): random stretches (maps constant after 1 - 4 tiles: swept batches), sparse and clustered plants (loud tiles
    next to each other, at a batch's first and last tile), low-entropy stretches and padding (chains that never merge:
    sweep -> fill, look-back over many batches), 8-bit plain / wildcard keywords (loud tiles found in the kernel), 16-bit
    wildcard keywords (the bitmap pre-pass) and 16-bit plain ones (no sweep) -- blocks of 512 KiB, 64 KiB, 8191 bytes and
    one chain over the whole buffer, against the oracle."""
    rng = np.random.default_rng(31000 + seed)
    elem = 1 if seed % 4 != 3 else 2
    be = elem == 2 and seed % 8 == 7
    kw, wc = [("relative", 0), ("re*at*ve", ord("*")), ("abcab", 0), ("te*ts*h", ord("*"))][seed % 4]
    n = (20 << 20) + int(rng.integers(0, 5000)) * elem
    hi = 256 if elem == 1 else 65536
    d = rng.integers(0, hi, n // elem).astype(np.int64)
    vals = [None if (wc and ord(c) == wc) else ord(c) for c in kw]
    lits = [v for v in vals if v is not None]

    def plant(pos):
        sh = int(rng.integers(-min(lits), hi - max(lits)))
        for j, v in enumerate(vals):
            if v is not None and pos + j < d.size:
                d[pos + j] = v + sh
    for _ in range(300):                                      # sparse
        plant(int(rng.integers(0, d.size - 16)))
    for c in range(12):                                       # clusters: neighbouring tiles, tile and batch edges
        at = int(rng.integers(0, d.size - 40000))
        at -= at % 2044 if c % 3 == 0 else (at % 32704 if c % 3 == 1 else 0)
        for k in range(int(rng.integers(2, 30))):
            plant(max(0, at + int(rng.integers(-20, 6200))))
    for _ in range(5):                                        # low entropy / padding: chains that do not merge
        at = int(rng.integers(0, d.size - (1 << 20)))
        ln = int(rng.integers(3000, 600000))
        alpha = int(rng.choice([1, 2, 3, 5]))
        d[at:at + ln] = rng.integers(0, alpha, ln) + int(rng.integers(0, hi - 8))
    if seed % 2:
        a = int(rng.integers(0, d.size - 70000))
        d[a:a + 66000] = np.arange(66000) % 251               # a ramp: 'abc..'-like keywords match it wholesale
    rom = d.astype(np.uint8 if elem == 1 else (">u2" if be else "<u2")).view(np.uint8)
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(elem, kw, wc), oracle.plan(elem, kw, wc)
    gpu_engine.set_engine(2)
    try:
        for block in (524288, 65536, 8191):
            got = gpu_engine.scan(plan, block_bytes=block, big_endian=be, cap=1 << 21)
            assert gpu_engine.counters()["path"] == 3
            assert got.tolist() == oracle.engine(oplan, rom, block, be).tolist(), (seed, kw, block)
        if not be:
            data = rom if elem == 1 else rom[: (rom.size // 2) * 2].view("<u2")
            assert gpu_engine.scan(plan, cap=1 << 21).tolist() == oracle.search(oplan, data).tolist(), (seed, kw, "whole")
    finally:
        gpu_engine.set_engine(0)


LOUD_CASES = [(1, "qz", 0, False, 0), (1, "q*", ord("*"), False, 0), (1, "q*v", ord("*"), False, 0), (1, "abc", 0, False, 0),
              (1, "qzvk", 0, False, 300), (1, "relativesrch", 0, False, 250), (1, "re*ative*ear*hxy", ord("*"), False, 250),
              (1, "a keyword of twenty-two", 0, False, 300), (1, "k" * 40, 0, False, 500),
              (2, "qz", 0, False, 700), (2, "q*v", ord("*"), True, 0), (2, "te*ts*h", ord("*"), False, 300)]


@pytest.mark.parametrize("elem,kw,wc,be,every", LOUD_CASES, ids=["%d-%s-%d" % (c[0], c[1][:12], c[4]) for c in LOUD_CASES])
def test_forward_engine_on_loud_batches(mm, gpu_engine, oracle, elem, kw, wc, be, every):
    """The forward engine forced on ROMs where most tiles of a batch have something to report (csrc/mm_forward.h, round 6):
    two-symbol keywords (one phase: no maps, no look-back, the finds straight off the flags), three-symbol wildcard keywords
    (chains merge: the batch sweeps for its exit phase only and WALKS its tiles with the entry phase from the look-back;
    the walk on a bit mask in registers), 'abc' (chains never merge: sweep -> fill), keywords planted every `every`
    elements (the walk with super-group tables, with the one-lane threading beyond 13 phases, with wide maps beyond 32),
    16-bit keywords on the bitmap pre-pass -- blocks of 512 KiB, 8191 bytes and one chain over the whole buffer, a ragged
    end, against the oracle."""
    rng = np.random.default_rng(4242 + len(kw) + 7 * elem + every)
    hi = 256 if elem == 1 else 65536
    n = (6 << 20) // elem + int(rng.integers(1, 3000))
    d = rng.integers(0, hi, n).astype(np.int64)
    vals = [None if (wc and ord(c) == wc) else ord(c) for c in kw]
    lits = [v for v in vals if v is not None]
    if every:
        for pos in range(int(rng.integers(0, every)), n - len(kw), every):
            pos += int(rng.integers(0, every // 2))
            sh = int(rng.integers(-min(lits), hi - max(lits)))
            for j, v in enumerate(vals):
                if v is not None and pos + j < n:
                    d[pos + j] = v + sh
    quiet = int(rng.integers(0, n - 400000))
    d[quiet:quiet + 300000] = rng.integers(0, hi, 300000) & ~1 if len(lits) < 4 else d[quiet:quiet + 300000]   # (a stretch with fewer hits)
    rom = d.astype(np.uint8 if elem == 1 else (">u2" if be else "<u2")).view(np.uint8)
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(elem, kw, wc), oracle.plan(elem, kw, wc)
    gpu_engine.set_engine(2)
    try:
        for block in (524288, 8191):
            got = gpu_engine.scan(plan, block_bytes=block, big_endian=be, cap=1 << 23)
            assert gpu_engine.counters()["path"] == 3
            want = oracle.engine(oplan, rom, block, be)
            assert got.size == want.size and np.array_equal(got, want), (kw, block, got.size, want.size)
        if not be:
            data = rom if elem == 1 else rom[: (rom.size // 2) * 2].view("<u2")
            got, want = gpu_engine.scan(plan, cap=1 << 23), oracle.search(oplan, data)
            assert got.size == want.size and np.array_equal(got, want), (kw, "whole", got.size, want.size)
    finally:
        gpu_engine.set_engine(0)


RUN_KEYWORDS = [("aaaa", 0), ("abcd", 0), ("dcba", 0), ("aceg", 0), ("aa", 0), ("abc", 0), ("a*a*a", ord("*")), ("ab*d", ord("*")), ("d*ba", ord("*")),
                ("a" * 40, 0), ("abcdefghijklmnopqrstuvwxyz", 0)]


@pytest.mark.parametrize("kw,wc", RUN_KEYWORDS, ids=[k[:8] + str(len(k)) for k, _ in RUN_KEYWORDS])
def test_forward_engine_on_runs_and_ramps(mm, gpu_engine, oracle, kw, wc):
    """Keywords that match padding and ramps wholesale, the forward engine forced and the library's own routing: runs of one
    byte, ramps up and down and in steps of two -- modulo 256, so that a ramp's wrap (255 -> 0) separates the plain loop's
    exact differences from the wildcard loop's modular ones -- from three bytes to hundreds of KiB long, at tile, batch and
    block edges.  Every position of such a stretch passes the first compare: dwords whose four positions all hit take the
    rest of the compare loop four at a time (csrc/mm_forward.h, mm_fwd_jumps), the others one by one; both against the
    oracle, blocks of 512 KiB, 8191 bytes and one chain over the whole buffer."""
    rng = np.random.default_rng(777 + len(kw) + wc)
    n = (5 << 20) + int(rng.integers(1, 3000))
    d = rng.integers(0, 256, n).astype(np.int64)
    at = int(rng.integers(0, 5000))
    kind = 0
    while at < n - 700000:
        ln = int(rng.choice([3, 4, 5, 7, 17, 40, 300, 2044, 2045, 5000, 40000, 600000], p=[.1, .1, .1, .1, .1, .1, .1, .05, .05, .1, .07, .03]))
        base = int(rng.integers(0, 256))
        if kind % 4 == 0:
            d[at:at + ln] = base
        elif kind % 4 == 1:
            d[at:at + ln] = (base + np.arange(ln)) % 256
        elif kind % 4 == 2:
            d[at:at + ln] = (base - np.arange(ln)) % 256
        else:
            d[at:at + ln] = (base + 2 * np.arange(ln)) % 256
        kind += 1
        gap = int(rng.choice([0, 1, 2, 3, 50, 2044 - (at + ln) % 2044, 30000]))
        at += ln + gap
    rom = d.astype(np.uint8)
    gpu_engine.upload(rom)
    plan, oplan = mm.plan_relative(1, kw, wc), oracle.plan(1, kw, wc)
    for engine in (2, 0):
        gpu_engine.set_engine(engine)
        try:
            for block in (524288, 8191):
                got, want = gpu_engine.scan(plan, block_bytes=block, cap=1 << 22), oracle.engine(oplan, rom, block, False)
                assert got.size == want.size and np.array_equal(got, want), (kw, engine, block, got.size, want.size)
            got, want = gpu_engine.scan(plan, cap=1 << 22), oracle.search(oplan, rom)
            assert got.size == want.size and np.array_equal(got, want), (kw, engine, "whole", got.size, want.size)
        finally:
            gpu_engine.set_engine(0)
