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


@pytest.mark.gpu
def test_progress_and_abort_on_a_4gib_file(mm):
    """SearchEngine<T>::run on a 4 GiB file in 512 KiB blocks (8192 of them -- BASELINE C2's shape as a file): exactly
    8192 + 3 callbacks, ticks arriving while the file streams to HBM, and an abort raised 1 / 6 / 15 ms into the ingest
    (~80 ms) brings run() back empty-handed within 10 ms (search_engine.cpp:161-187)."""
    exe = _build_tests(mm)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900,
                       env=dict(os.environ, MMOORE_TEST_BIGFILE_MIB="4096"))
    print(r.stdout[-3000:])
    assert r.returncode == 0 and " 0 failures" in r.stdout, r.stdout[-3000:]
    assert "big file (4096 MiB)" in r.stdout
    assert r.stdout.count("run() back") == 3, r.stdout[-2000:]       # all three aborts hit the ingest


@pytest.mark.gpu
def test_facade_differential_against_the_reference_core_on_the_gpu(mm):
    """oracle/_ref/facade_diff (tests/cpp/facade_diff_*.cpp, prebuilt in the build container): MonkeyMoore<T>::search and
    SearchEngine<T>::run of the MI355X facade against the reference core in ONE process through the API both share --
    positions, equivalency maps, previews, callback counts -- on 200 random cases.  The reference side can be made to
    allocate without bound by keywords the generator avoids: the program watches its own resident size, and
    so does this test from outside (a second, independent watchdog), besides a wall-clock limit.  The same 200 cases
    run against the CPU double in the CPU suite (tests/test_sanitize.py), so the reference side's behaviour on them is
    known before they ever reach a GPU box."""
    import time
    exe = os.path.join(REF_BIN, "facade_diff")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/facade_diff was not prebuilt (needs /root/reference in the build container)")
    mm.build.build_all()
    cases = os.environ.get("MM_FACADE_DIFF_CASES", "200")
    p = subprocess.Popen([exe, cases], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    t0, killed = time.time(), None
    page = os.sysconf("SC_PAGESIZE")
    while p.poll() is None:
        time.sleep(0.05)
        try:
            with open("/proc/%d/statm" % p.pid) as f:
                resident = int(f.read().split()[1]) * page
        except (OSError, IndexError, ValueError):
            resident = 0
        if resident > (8 << 30) or time.time() - t0 > 900:
            killed = "resident %d MiB after %.0f s" % (resident >> 20, time.time() - t0)
            p.kill()
            break
    out = p.communicate()[0]
    print(out[-3000:])
    assert killed is None, killed
    assert p.returncode == 0, out[-3000:]
    assert "%s cases" % cases in out and " 0 failures" in out
    compared = int(out.split(" matches compared")[0].split()[-1])
    assert compared > 20000, out[-500:]
