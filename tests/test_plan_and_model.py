# SPDX-License-Identifier: GPL-3.0-or-later
"""CPU tests of the product's host logic and of the device ALGORITHM (via the
Python model in tests/_model.py), against the golden vectors and the oracle.
No GPU needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import _model
from conftest import ROOT, load_golden


def _plan(mm, c):
    if c.get("values") is not None:
        return mm.plan_value_scan(c["elem_bytes"], c["values"])
    return mm.plan_relative(c["elem_bytes"], c["keyword"], c["wildcard"], c.get("char_seq"))


def _rom(c, key="data"):
    if key == "data":
        dt = np.uint8 if c["elem_bytes"] == 1 else "<u2"
        return np.array(c["data"], dtype=dt).view(np.uint8)
    return np.array(c["file"], dtype=np.uint8)


def test_library_exports_every_declared_symbol(mm):
    """The C-ABI library loads (no GPU needed) and exports all of include/mmoore_hip.h."""
    header = open(os.path.join(ROOT, "include", "mmoore_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(mmh_[a-z_]+)\s*\(", header)))
    assert len(declared) >= 18
    lib = C.CDLL(mm.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(mm.EXPORTS) == declared


def test_no_gpu_means_loud_failure(mm):
    if mm.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(mm.MMError) as e:
        mm.Engine(0)
    assert e.value.code == mm.MMH_E_DEVICE


def test_plan_rejects_what_the_reference_rejects(mm):
    for args in ((1, "a"), (1, "*a", ord("*")), (1, [0x3042, 0x41]), (1, "x" * 129), (1, "***", ord("*"))):
        with pytest.raises(mm.MMError) as e:
            mm.plan_relative(*args)
        assert e.value.code == mm.MMH_E_PLAN
    with pytest.raises(mm.MMError):
        mm.plan_value_scan(1, [5])
    with pytest.raises(mm.MMError):
        mm.plan_relative(3, "abc")


@pytest.mark.parametrize("case", load_golden("kat_matcher.json"), ids=lambda c: c["name"])
def test_plan_sequential_model_matcher_kats(mm, case):
    pl = _plan(mm, case)
    rom = _rom(case)
    g = _model.Geom(rom.size, case["elem_bytes"], pl.L)
    assert _model.chain_seq(pl, rom, g) == case["expect"]
    assert _model.fast_path(pl, rom, g, tile=8, seg=2) == case["expect"]


@pytest.mark.parametrize("case", load_golden("kat_engine.json"), ids=lambda c: c["name"])
def test_plan_models_engine_kats(mm, case):
    pl = _plan(mm, case)
    rom = _rom(case, "file")
    for bs in case["block_sizes"]:
        g = _model.Geom(rom.size, case["elem_bytes"], pl.L, bs, case["big_endian"])
        assert _model.chain_seq(pl, rom, g) == case["expect"], bs
        assert _model.fast_path(pl, rom, g, tile=8, seg=2) == case["expect"], bs


def test_models_against_reference_vectors(mm):
    """Host plan + sequential model + filter/resolver model == compiled reference."""
    cases = load_golden("diff_search.json")
    rng = np.random.default_rng(5)
    n_fast = 0
    for idx, c in enumerate(cases):
        pl = _plan(mm, c)
        rom = _rom(c)
        g = _model.Geom(rom.size, c["elem_bytes"], pl.L)
        assert _model.chain_seq(pl, rom, g) == c["expect"], c
        if idx % 3 == 0:
            tile = int(rng.choice([4, 8, 16, 64]))
            seg = int(rng.choice([1, 2, 4]))
            assert _model.fast_path(pl, rom, g, tile=tile, seg=seg) == c["expect"], (c, tile, seg)
            n_fast += 1
    assert n_fast > 500


def test_models_against_tiny_reference_vectors(mm):
    """A sample of the 11 000 tiny vectors (tests/golden/diff_tiny.json.gz) through the host plan and the
    Python model of the device algorithm (every 8th case; the oracle and the GPU take all of them)."""
    from conftest import load_tiny
    search, engine = load_tiny()
    for c in search[::8]:
        pl = _plan(mm, c)
        rom = c["data"].view(np.uint8)
        g = _model.Geom(rom.size, c["elem_bytes"], pl.L)
        assert _model.chain_seq(pl, rom, g) == c["expect"], c
        assert _model.fast_path(pl, rom, g, tile=16, seg=2) == c["expect"], c
    for c in engine[::8]:
        pl = _plan(mm, c)
        g = _model.Geom(c["file"].size, c["elem_bytes"], pl.L, c["block_size"], c["big_endian"])
        assert _model.chain_seq(pl, c["file"], g) == c["expect"], c
        assert _model.fast_path(pl, c["file"], g, tile=16, seg=4) == c["expect"], c


def test_models_against_reference_engine_vectors(mm):
    cases = load_golden("diff_engine.json")
    for idx, c in enumerate(cases):
        pl = _plan(mm, c)
        rom = _rom(c, "file")
        g = _model.Geom(rom.size, c["elem_bytes"], pl.L, c["block_size"], c["big_endian"])
        assert _model.chain_seq(pl, rom, g) == c["expect"], c
        if idx % 2 == 0:
            assert _model.fast_path(pl, rom, g, tile=16, seg=4) == c["expect"], c


@pytest.mark.parametrize("elem,kw,conds,verify", [
    # (keyword position, gap) of the SWAR conditions the streaming filter keys on
    (1, "relativesrch", [(11, 1), (10, 1), (9, 1), (8, 1)], False),     # BASELINE C2: the contiguous shape
    (1, "abcde", [(4, 1), (3, 1), (2, 1), (1, 1)], False),
    # bench_search.cpp Wildcard/Middle: the streaming loop tests conditions 0 and 1 on every byte -- two of the same gap
    # cost it one SWAR subtraction per dword instead of two, so the gap-1 condition three positions back moves up
    (1, "ab*de", [(4, 1), (1, 1), (3, 2)], False),
    (1, "ab*defg", [(6, 1), (5, 1), (4, 1)], False),                    # (the look-back ends 4 bytes behind the anchor)
    # (two conditions is all one dword of look-back gives these two: the wide shapes take them, with the places beyond it)
    (1, "a*cd*f", [(5, 2), (3, 1), (2, 2)], False),
    (1, "a*c*ef*h", [(7, 2), (5, 1), (4, 2), (2, 2)], False),
    (1, "*bcde", [(4, 1), (3, 1), (2, 1)], False),
    (1, "abcd*", [(3, 1), (2, 1), (1, 1)], False),
    (1, "re*ative*ear*hxy", [(7, 1), (6, 1), (5, 1), (4, 1)], False),   # BASELINE C3
    # the wide shapes (round 6): gaps up to 4 on the hot path, every further condition with run-time places
    (1, "a*c*e*g", [(6, 2), (4, 2), (2, 2)], False),
    (1, "a**d**g", [(6, 3), (3, 3)], False),
    (1, "qz**mb", [(5, 1), (4, 3), (1, 1)], False),
    (1, "abc**de", [(6, 1), (5, 3), (2, 1), (1, 1)], False),
    (1, "a***b***c", [(8, 4), (4, 4)], False),
    (1, "ab***cd**ef", [(10, 1), (9, 3), (6, 1), (5, 4)], False),
    (1, "a**b", [], None),                                              # (two literals: one 7-bit condition is no filter)
    (1, "a****bc", [(6, 1)], True),                                     # (a gap of 5: the one-dword shape's single condition)
    (2, "q**k", [(3, 3)], False),
    (2, "ab**cd", [(5, 1), (4, 3), (1, 1)], False),
    (2, "textsrch", [(7, 1), (6, 1)], False),                           # BASELINE C4
    (2, "ab*de", [(4, 1), (3, 2)], False),
    (2, "abc*e", [(2, 1), (1, 1)], False),                              # ties go to the contiguous run
    (2, "*bc*e", [(4, 2), (2, 1)], False),
    (2, "a*c*e", [(4, 2), (2, 2)], False),
])
def test_filter_conditions_chosen(mm, elem, kw, conds, verify):
    info = mm.filter_shape(mm.plan_relative(elem, kw, ord("*")))
    assert info["conditions"] == conds
    assert info["ncond"] == len(conds)
    if conds:
        assert info["anchor"] == conds[0][0]
        assert info["verify_in_filter"] == verify


def test_mixed_gaps_on_the_per_byte_stage_take_a_wide_shape(mm):
    """Two per-byte conditions of different gaps cost the one-dword shapes two exact SWAR subtractions per dword; the wide
    shapes' seven-bit stage takes them where it has as many conditions (csrc/mm_kernels.hip choose_filter)."""
    for kw in ("qz*k", "q*vk"):
        info = mm.filter_shape(mm.plan_relative(1, kw, ord("*")))
        assert info["ncond"] == 2 and info["shape"] & 0x200, (kw, info)
    info = mm.filter_shape(mm.plan_relative(1, "ab*de", ord("*")))       # (two conditions of the SAME gap moved up: stays)
    assert info["shape"] & 0x200 == 0 and info["conditions"][:2] == [(4, 1), (1, 1)], info


def test_filter_conditions_are_necessary_for_a_match(mm, oracle):
    """Each chosen condition (position i, gap g, expected delta of the plan) must hold at every
    match the oracle reports: the filter may only over-approximate the reference."""
    rng = np.random.default_rng(99)
    for elem in (1, 2):
        for kw in ("ab*de", "a*c*e*g", "abc*efgh*jkl", "*b*d*f*h", "zyxwv", "ab**ef*hi", "qz**mb", "q**k**x", "a***b***c",
                   "ab***cd**ef", "a**bc***d*e", "a*b**c***d"):
            plan = mm.plan_relative(elem, kw, ord("*"))
            info = mm.filter_shape(plan)
            assert info["ncond"] >= 1
            hi = 256 if elem == 1 else 65536
            d = rng.integers(0, hi, 4096).astype(np.int64)
            for pos in range(8, 4000, 97):
                sh = int(rng.integers(-90, hi - 130))
                for j, ch in enumerate(kw):
                    if ch != "*":
                        d[pos + j] = ord(ch) + sh
            data = d.astype(np.uint8 if elem == 1 else np.uint16)
            hits = oracle.search(oracle.plan(elem, kw, ord("*")), data)
            assert len(hits) >= 10
            for h in hits:
                for i, g in info["conditions"]:
                    delta = (int(data[h + i]) - int(data[h + i - g])) % hi
                    assert delta == plan.expected[i] % hi, (kw, i, g)
