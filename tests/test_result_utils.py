# SPDX-License-Identifier: GPL-3.0-or-later
"""what a front end does with a result list -- include/mmoore/result_utils.hpp (de-dup by
equivalency map, "%c=%0NX " values with the GUI's byte order rule, hex / decimal offsets, table export with
the 26-letter expansion and wrap) and the command line over the facade, tools/mmoore_search.cpp.

CPU: the header on hand-derived vectors (tests/cpp/result_utils_tests.cpp, citing monkey_frame.cpp:1215-1273 and
table_creator.cpp:164-194); the command line built against the CPU test double of tests/test_sanitize.py under
ASan + UBSan, on the reference's own engine vectors.  GPU: the same command line on the real facade."""
import json
import os
import subprocess

import pytest

from conftest import ROOT, load_golden

CPP = os.path.join(ROOT, "tests", "cpp")
BUILD = os.path.join(CPP, "build")
CLI_SRC = os.path.join(ROOT, "tools", "mmoore_search.cpp")


def test_result_utils_on_hand_derived_vectors():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "result_utils_tests")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(CPP, "result_utils_tests.cpp"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and " 0 failures" in r.stdout, r.stdout[-3000:]
    assert int(r.stdout.split(" checks")[0].split()[-1]) >= 50


def _kat(name):
    return next(c for c in load_golden("kat_engine.json") if c["name"] == name)


def _write(tmp_path, case):
    p = tmp_path / "rom.bin"
    p.write_bytes(bytes(case["file"]))
    return str(p)


def _cli_cases(run, tmp_path):
    """the command line against the reference's own engine vectors; run(args) -> CompletedProcess"""
    # 8-bit 'text' (test_search_engine.cpp:26-81): five matches, three distinct encodings
    case = _kat("u8 text")
    path = _write(tmp_path, case)
    r = run(["--block", "23", "--all", "--dec", path, "text"])
    assert r.returncode == 0, r.stderr
    rows = [ln.split("\t") for ln in r.stdout.splitlines()]
    assert [int(x[0]) for x in rows] == case["expect"]
    data = case["file"]
    for off, row in zip(case["expect"], rows):
        base = data[off] - ord("t")                               # 'a' + dist, monkey_moore.cpp:381-391
        assert row[1] == "A=%02X a=%02X " % ((65 + base) & 0xFF, (97 + base) & 0xFF)
    r = run(["--block", "23", path, "text"])                      # default: hex offsets, one row per distinct map
    shown = [ln.split("\t") for ln in r.stdout.splitlines()]
    assert len({x[1] for x in rows}) == len(shown) < len(rows)
    assert shown[0][0] == "0x0" and all(x[0].startswith("0x") for x in shown)
    assert "%d shown, 5 matches" % len(shown) in r.stderr
    # 16-bit big endian (:139-158), previews (:245-261), a saved table
    case = _kat("u16 BE text")
    path = _write(tmp_path, case)
    r = run(["--bits", "16", "--be", "--block", "24", "--all", "--dec", path, "text"])
    assert [int(ln.split("\t")[0]) for ln in r.stdout.splitlines()] == case["expect"]
    case = _kat("u16 preview theater")
    path = _write(tmp_path, case)
    table = tmp_path / "out.tbl"
    r = run(["--bits", "16", "--block", "32", "--all", "--preview", "25", "--table", str(table), path, "theater"])
    rows = [ln.split("\t") for ln in r.stdout.splitlines()]
    assert [int(x[0], 16) for x in rows] == case["expect"] and [x[2] for x in rows] == case["previews"]
    text = table.read_bytes().decode("utf-8")
    lines = text.split("\r\n")
    assert lines[-1] == "" and len(lines) == 53                   # A..Z, a..z
    a = int(rows[0][1].split("a=")[1][:4], 16)                    # as displayed: byte-reversed after a little-endian search
    assert "%04X=a" % a in lines
    # custom wildcard (:429-447), value scan, nothing found, bad arguments
    case = _kat("u8 custom wildcard $atch")
    path = _write(tmp_path, case)
    r = run(["--wildcard", "$", "--block", "20", "--all", "--dec", path, "$atch"])
    assert [int(ln.split("\t")[0]) for ln in r.stdout.splitlines()] == case["expect"]
    r = run(["--values", "1,2,3,5", "--all", "--dec", path])
    assert r.returncode in (0, 1)
    r = run(["--all", path, "zzzzqqqq"])
    assert r.returncode in (0, 1)
    assert run([path, "ab"]).returncode == 2                      # the GUI's three-literal rule (monkey_frame.cpp:1040)
    assert run(["--bits", "12", path, "text"]).returncode == 2
    assert run([str(tmp_path / "missing.bin"), "text"]).returncode == 2


def test_command_line_on_the_cpu_double_under_sanitizers(tmp_path):
    from test_sanitize import ENV, FACADE, SAN
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "mmoore_search_asan")
    subprocess.check_call(SAN + [CLI_SRC] + FACADE + ["-o", exe])

    def run(args):
        r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, env=ENV)
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        return r
    _cli_cases(run, tmp_path)


def test_command_line_fails_loudly_without_gpu(mm, tmp_path):
    if mm.device_count() > 0:
        pytest.skip("a GPU is present")
    exe = _build_cli(mm)
    p = tmp_path / "rom.bin"
    p.write_bytes(bytes(range(64)))
    r = subprocess.run([exe, str(p), "text"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr and r.stdout == ""


def _build_cli(mm):
    mm.build.build_all()
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "mmoore_search")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), CLI_SRC, "-L" + mm.build.LIB_DIR,
                           "-lmonkey-core", "-lmmoore_hip", "-Wl,-rpath," + mm.build.LIB_DIR, "-pthread", "-o", exe])
    return exe


@pytest.mark.gpu
def test_command_line_on_the_gpu(mm, tmp_path):
    exe = _build_cli(mm)
    _cli_cases(lambda args: subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300), tmp_path)
