#!/usr/bin/env python3
# SPDX-License-Identifier: GPL-3.0-or-later
"""Turns tests/golden/kat_*.json (the reference's own known-answer vectors) into a C++
include for tests/cpp/facade_tests.cpp.  Usage: gen_cases.py <out.inc>"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "..", "golden")


def vec(xs, suffix=""):
    return "{" + ",".join(str(int(x)) + suffix for x in xs) + "}"


def cstr(s):
    return '"' + s.encode("utf-8").decode("latin-1").encode("unicode_escape").decode("ascii").replace('"', '\\"') + '"'


def main(out_path):
    km = json.load(open(os.path.join(GOLDEN, "kat_matcher.json"), encoding="utf-8"))
    ke = json.load(open(os.path.join(GOLDEN, "kat_engine.json"), encoding="utf-8"))
    o = []
    o.append("static const std::vector<MatcherCase> matcher_cases = {")
    for c in km:
        maps = "{" + ",".join("{" + ",".join("{%d,%d}" % (ord(k), v) for k, v in m.items()) + "}" for m in (c["maps"] or [])) + "}"
        o.append("  {%s, %d, %s, %s, %s, %s, %d, %s, %s, %s}," % (
            cstr(c["name"]), c["elem_bytes"], vec(c["data"]), "true" if c["values"] is not None else "false",
            vec(c["values"] or []), vec(c["keyword"] or []), c["wildcard"], vec(c.get("char_seq") or []),
            vec(c["expect"], "ull"), maps))
    o.append("};")
    o.append("static const std::vector<EngineCase> engine_cases = {")
    for c in ke:
        prev = "{" + ",".join(cstr(p) for p in (c["previews"] or [])) + "}"
        o.append("  {%s, %d, %s, %s, %d, %s, %s, %s, %s, %d, %s, %s, %s}," % (
            cstr(c["name"]), c["elem_bytes"], vec(c["file"]), vec(c["keyword"]), c["wildcard"],
            vec(c.get("char_seq") or []), "true" if c["big_endian"] else "false", vec(c["block_sizes"]),
            vec(c["threads"]), c["preview_width"], vec(c["expect"], "ull"),
            "true" if c["previews"] is not None else "false", prev))
    o.append("};")
    open(out_path, "w").write("\n".join(o) + "\n")


if __name__ == "__main__":
    main(sys.argv[1])
