# SPDX-License-Identifier: GPL-3.0-or-later
"""AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of the product, in the build container
(GPU sanitizers and XNACK are not available on the pool):

* the plan builder (csrc/mm_plan.cpp) on 60 000 random keywords of every mode, also malformed ones;
* the C++ facade (host/monkey_moore.cpp, host/search_engine.cpp: partition rounds, progress / abort,
  equivalency maps, preview windows) driven by the facade tests of tests/cpp/facade_tests.cpp and -- when
  the reference tree is present -- by the reference's OWN unit tests, compiled from where they lie.

The facade needs a device behind the C ABI; here that is tests/cpp/cpu_backend_double.cpp, a CPU TEST DOUBLE
of the dozen entries the facade calls (it walks the real plan as include/mmoore_hip.h documents it).  The
double exists for these sanitizer runs only: it is not part of the package, nothing outside tests/ builds it,
and the product still fails loudly without a GPU (test_facade_fails_loudly_without_gpu)."""
import os
import subprocess

import pytest

from conftest import ROOT

CPP = os.path.join(ROOT, "tests", "cpp")
BUILD = os.path.join(CPP, "build")
CSRC = os.path.join(ROOT, "monkey-moore_amd", "csrc")
HOST = os.path.join(ROOT, "monkey-moore_amd", "host")
REF_TESTS = "/root/reference/tests"
SAN = ["g++", "-std=c++17", "-O0", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread",
       "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + HOST]
FACADE = [os.path.join(CPP, "cpu_backend_double.cpp"), os.path.join(CSRC, "mm_plan.cpp"), os.path.join(HOST, "monkey_moore.cpp"),
          os.path.join(HOST, "search_engine.cpp")]
# (MMOORE_TEST_ABORT_MS: the facade tests hold an aborted run() to 10 ms on the device; instrumented builds of the CPU double
# need far longer for the 4 MiB piece they are in the middle of)
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", MMOORE_TEST_ABORT_MS="2000")


def _run(exe, **env):
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=dict(ENV, **env))
    assert r.returncode == 0, r.stdout[-4000:]
    assert "Sanitizer" not in r.stdout and "runtime error" not in r.stdout, r.stdout[-4000:]
    return r.stdout


def test_plan_builder_under_sanitizers():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "plan_sanitize")
    subprocess.check_call(SAN[:3] + ["-O1"] + SAN[4:] + [os.path.join(CPP, "plan_sanitize.cpp"), os.path.join(CSRC, "mm_plan.cpp"), "-o", exe])
    out = _run(exe)
    assert "plans built" in out and "no sanitizer report" in out


def test_facade_under_sanitizers():
    os.makedirs(BUILD, exist_ok=True)
    subprocess.check_call(["python3", os.path.join(CPP, "gen_cases.py"), os.path.join(BUILD, "cases.inc")])
    exe = os.path.join(BUILD, "facade_tests_asan")
    subprocess.check_call(SAN + ["-I" + BUILD, os.path.join(CPP, "facade_tests.cpp")] + FACADE + ["-o", exe])
    # (MMOORE_DOUBLE_PIECE_US: the double's ingest takes 2 ms per 4 MiB piece, so that the 8192-block file of the facade
    # tests streams in over ~30 ms -- ticks arrive meanwhile, and the aborts raised 1 / 6 / 15 ms in hit the ingest)
    out = _run(exe, MMOORE_DOUBLE_PIECE_US="2000")
    assert " 0 failures" in out
    assert out.count("run() back") >= 2, out[-2000:]
    # run()'s multi-device rounds: three "devices", partitions dealt out over them
    assert " 0 failures" in _run(exe, MMOORE_DOUBLE_DEVICES="3", MMOORE_HIP_MULTI="1", MMOORE_DOUBLE_PIECE_US="500")


def test_reference_unit_tests_on_the_facade_under_sanitizers():
    if not os.path.exists(os.path.join(REF_TESTS, "test_search_engine.cpp")):
        pytest.skip("reference sources not present")
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "ref_tests_asan")
    srcs = [os.path.join(REF_TESTS, f) for f in ("test_monkey_moore.cpp", "test_search_engine.cpp", "test_text_utils.cpp")]
    subprocess.check_call(SAN + ["-I" + os.path.join(ROOT, "tests", "shim"), "-I" + REF_TESTS] + srcs +
                          [os.path.join(ROOT, "tests", "shim", "catch_main.cpp")] + FACADE + ["-o", exe])
    assert " 0 failures" in _run(exe)
    assert " 0 failures" in _run(exe, MMOORE_DOUBLE_DEVICES="2", MMOORE_HIP_MULTI="1")


def test_facade_under_thread_sanitizer():
    """concurrent search() calls on one MonkeyMoore, run()'s loader threads and serialized callbacks"""
    os.makedirs(BUILD, exist_ok=True)
    subprocess.check_call(["python3", os.path.join(CPP, "gen_cases.py"), os.path.join(BUILD, "cases.inc")])
    exe = os.path.join(BUILD, "facade_tests_tsan")
    tsan = [a if not a.startswith("-fsanitize=") else "-fsanitize=thread" for a in SAN if a != "-fno-sanitize-recover=all"]
    subprocess.check_call(tsan + ["-I" + BUILD, os.path.join(CPP, "facade_tests.cpp")] + FACADE + ["-o", exe])
    for env in ({}, {"MMOORE_DOUBLE_DEVICES": "3", "MMOORE_HIP_MULTI": "1"}):
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                           env=dict(os.environ, MMOORE_TEST_ABORT_MS="5000", **env))
        assert r.returncode == 0 and "ThreadSanitizer" not in r.stdout, r.stdout[-4000:]
        assert " 0 failures" in r.stdout


def test_facade_differential_against_the_reference_core():
    """tests/cpp/facade_diff_*.cpp with the CPU test double behind the facade: MonkeyMoore<T>::search (positions +
    equivalency maps) and SearchEngine<T>::run (byte offsets, maps, decoded previews, callback counts) of this
    repository's facade against the reference core -- its own sources, class and namespace renamed on the command
    line so that both fit one binary -- on 400 random cases.  The facade's side runs under ASan + UBSan (the
    reference's side does not: it reads 16-bit elements at odd addresses by design, byteswap.hpp:75)."""
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "src", "core", "search_engine.cpp")):
        pytest.skip("reference sources not present")
    os.makedirs(BUILD, exist_ok=True)
    rename = ["-DMonkeyMoore=MonkeyMooreRef", "-Dmmoore=mmoore_ref"]
    plain = ["g++", "-std=c++17", "-O1", "-pthread"]
    objs = []
    for name, src, extra in (("diff_ref_mm.o", os.path.join(ref, "src", "core", "monkey_moore.cpp"), ["-DNDEBUG"]),
                             ("diff_ref_se.o", os.path.join(ref, "src", "core", "search_engine.cpp"), ["-DNDEBUG"]),
                             ("diff_side_ref.o", os.path.join(CPP, "facade_diff_side.cpp"), ["-DDIFF_SIDE=ref", "-I" + CPP])):
        obj = os.path.join(BUILD, name)
        subprocess.check_call(plain + rename + extra + ["-I" + os.path.join(ref, "include"), "-I" + os.path.join(ref, "src", "core"), "-c", src, "-o", obj])
        objs.append(obj)
    side = os.path.join(BUILD, "diff_side_gpu.o")
    san = [a for a in SAN if a != "-O0"] + ["-O1"]
    subprocess.check_call(san + ["-DDIFF_SIDE=gpu", "-I" + CPP, "-c", os.path.join(CPP, "facade_diff_side.cpp"), "-o", side])
    exe = os.path.join(BUILD, "facade_diff_cpu")
    subprocess.check_call(san + ["-I" + CPP, os.path.join(CPP, "facade_diff_main.cpp"), side] + objs + FACADE + ["-o", exe])
    r = subprocess.run([exe, "400"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=ENV)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 failures" in r.stdout and "Sanitizer" not in r.stdout and "runtime error" not in r.stdout, r.stdout[-3000:]
    compared = int(r.stdout.split(" matches compared")[0].split()[-1])
    assert compared > 50000, r.stdout[-500:]
