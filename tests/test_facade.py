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


@pytest.mark.gpu
def test_reference_suites_through_the_facade(mm):
    exe = _build_tests(mm)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 failures" in r.stdout
