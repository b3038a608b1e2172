# SPDX-License-Identifier: GPL-3.0-or-later
"""The C++17 facade (include/mmoore: MonkeyMoore<T>, SearchEngine<T>) over the C ABI.

CPU part: it builds, exports the reference's symbol set and refuses to run without a GPU.
GPU part: tests/cpp/facade_tests.cpp replays the reference's own Catch2 vectors
(matcher KATs, engine KATs with previews, progress, abort, missing file) through it."""
import os
import subprocess

import pytest

from conftest import ROOT

CPP = os.path.join(ROOT, "tests", "cpp")
BUILD = os.path.join(CPP, "build")


def _build_tests(mm):
    mm.build.build_all()
    os.makedirs(BUILD, exist_ok=True)
    subprocess.check_call(["python3", os.path.join(CPP, "gen_cases.py"), os.path.join(BUILD, "cases.inc")])
    exe = os.path.join(BUILD, "facade_tests")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + BUILD,
                           os.path.join(CPP, "facade_tests.cpp"), "-L" + mm.build.LIB_DIR, "-lmonkey-core", "-lmmoore_hip",
                           "-Wl,-rpath," + mm.build.LIB_DIR, "-pthread", "-o", exe])
    return exe


def test_facade_builds_and_exports_reference_symbols(mm):
    mm.build.build_all()
    out = subprocess.check_output(["nm", "-DC", "--defined-only", mm.build.CORE_SO], text=True)
    for ty in ("unsigned char", "unsigned short"):
        assert "MonkeyMoore<%s>::search(" % ty in out
        assert "MonkeyMoore<%s>::MonkeyMoore(std::vector<char32_t" % ty in out
        assert "MonkeyMoore<%s>::MonkeyMoore(std::vector<short" % ty in out
        assert "mmoore::SearchEngine<%s>::run(" % ty in out


def test_facade_fails_loudly_without_gpu(mm):
    if mm.device_count() > 0:
        pytest.skip("a GPU is present")
    exe = _build_tests(mm)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stdout


REF = "/root/reference"
REF_BIN = os.path.join(ROOT, "oracle", "_ref")


def test_reference_harness_compiles(mm):
    """the reference's own harness -- tests/test_monkey_moore.cpp, tests/test_search_engine.cpp,
    tests/test_text_utils.cpp, benchmarks/bench_search.cpp -- compiles UNMODIFIED, from where it lies,
    against include/mmoore + libmonkey-core.so (oracle/Makefile: harness; Catch2 / google-benchmark
    stand-ins in tests/shim/).  Build container only: the sources do not travel."""
    if not os.path.exists(os.path.join(REF, "tests", "test_monkey_moore.cpp")):
        pytest.skip("reference sources not present (GPU box): the prebuilt binaries are run by the gpu test below")
    mm.build.build_all()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "harness"])
    for exe in ("ref_unit_tests", "ref_bench_search", "ref_unit_tests_refcore"):
        assert os.access(os.path.join(REF_BIN, exe), os.X_OK), exe
    # the binaries resolve the facade's symbols (undefined in them, defined in libmonkey-core.so)
    und = subprocess.check_output(["nm", "-DC", "--undefined-only", os.path.join(REF_BIN, "ref_unit_tests")], text=True)
    assert "MonkeyMoore<unsigned char>::search(" in und and "mmoore::SearchEngine<unsigned short>::run(" in und
    ldd = subprocess.check_output(["ldd", os.path.join(REF_BIN, "ref_bench_search")], text=True)
    assert "libmonkey-core.so" in ldd and "not found" not in ldd
    # the Catch2 stand-in itself: the same test objects linked with the REFERENCE core pass on the CPU
    r = subprocess.run([os.path.join(REF_BIN, "ref_unit_tests_refcore")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=300)
    assert r.returncode == 0 and " 0 failures" in r.stdout, r.stdout[-2000:]
    assert "14 test cases" in r.stdout
    if mm.device_count() == 0:
        # and without a GPU the facade build of the same tests fails loudly, not silently
        r = subprocess.run([os.path.join(REF_BIN, "ref_unit_tests")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           timeout=300)
        assert r.returncode != 0 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_reference_own_tests_pass_on_the_gpu_facade(mm):
    """The reference's unmodified Catch2 test sources, linked against the MI355X facade (prebuilt in
    the build container by oracle/Makefile: harness), run on the GPU: every assertion holds."""
    exe = os.path.join(REF_BIN, "ref_unit_tests")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_unit_tests was not prebuilt (needs /root/reference in the build container)")
    mm.build.build_all()
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert "14 test cases" in r.stdout and " 0 failures" in r.stdout


@pytest.mark.gpu
def test_reference_suites_through_the_facade(mm):
    exe = _build_tests(mm)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 failures" in r.stdout
