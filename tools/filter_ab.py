# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: the streaming kernel of several revisions of the device code, A/B in ONE process on ONE
box -- box-to-box spread (1.5-3 %) is as large as the differences in question.

    tools/build_variant.sh r03 434eef0 ; tools/build_variant.sh r04 1097c1c      (once, in the build container)
    python tools/filter_ab.py [--config C2|C3|C4] [--rounds 200] [TAG ...]        (GPU box; default tags: all of tools/ab/ + cur)

Every variant is its own libmmoore_hip_TAG.so (own kernels, own context) attached to the SAME ROM in HBM; the variants
take turns, one synchronous scan each per round, and report the streaming kernel's own duration (HIP events riding on the
kernel's dispatch, mmh_last_timings) -- median, mean, min over the rounds.  -> profiles/r05_filter_ab.log"""
import argparse
import ctypes as C
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

BLOCK = 524288
CONFIGS = {"C2": (4 << 30, 1, "relativesrch", 0, False), "C3": (4 << 30, 1, "re*ative*ear*hxy", ord("*"), False),
           "C4": (8 << 30, 2, "textsrch", 0, False), "C4BE": (8 << 30, 2, "textsrch", 0, True),
           # the reference's Wildcard/Middle benchmark keyword and two short wildcard keywords, on C2's ROM
           "WM": (4 << 30, 1, "mo*ke", ord("*"), False), "THIS": (4 << 30, 1, "th*s", ord("*"), False),
           "ACDF": (4 << 30, 1, "a*cd*f", ord("*"), False)}


class Variant:
    """the handful of C-ABI entry points every revision has"""

    def __init__(self, tag, path, mm):
        self.tag = tag
        self.lib = L = C.CDLL(path)
        L.mmh_last_error.restype = C.c_char_p
        L.mmh_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.mmh_destroy.argtypes = [C.c_void_p]
        L.mmh_destroy.restype = None
        L.mmh_rom_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.mmh_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64)]
        L.mmh_last_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.mmh_plan_relative.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p]
        self.h = C.c_void_p()
        self.check(L.mmh_create(0, C.byref(self.h)))
        # one launch over the whole ROM per scan: since round 5 mmh_scan runs a ROM of >= 1 GiB as a pipeline of parts
        # (MMH_ROUTE_NO_SPLIT = 16; older revisions have no such route and refuse the mask: they never split)
        if hasattr(L, "mmh_set_route"):
            L.mmh_set_route.argtypes = [C.c_void_p, C.c_uint32]
            L.mmh_set_route(self.h, 16)
        self.plan = mm.PlanDesc()                     # (the plan's layout has not changed since round 1)
        self.out = np.zeros(1 << 20, np.uint64)
        self.count = C.c_uint64(0)
        self.t = (C.c_float * 4)()

    def check(self, rc):
        if rc != 0:
            raise RuntimeError("%s: %s" % (self.tag, self.lib.mmh_last_error().decode()))

    def set_plan(self, elem, keyword, wildcard):
        kw = np.array([ord(c) for c in keyword], np.uint32)
        self.check(self.lib.mmh_plan_relative(elem, kw.ctypes.data_as(C.POINTER(C.c_uint32)), len(kw), wildcard, None, 0, C.byref(self.plan)))

    def attach(self, ptr, nbytes):
        self.check(self.lib.mmh_rom_attach(self.h, C.c_void_p(ptr), nbytes))

    def scan(self, big_endian):
        self.check(self.lib.mmh_scan(self.h, C.byref(self.plan), BLOCK, int(big_endian), 0, self.out.ctypes.data_as(C.POINTER(C.c_uint64)),
                                     self.out.size, C.byref(self.count)))
        self.check(self.lib.mmh_last_timings(self.h, self.t))
        return self.t[0], self.t[3], int(self.count.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--rounds", type=int, default=200)
    ap.add_argument("tags", nargs="*")
    args = ap.parse_args()
    mm = load_package()
    nbytes, elem, kw, wc, be = CONFIGS[args.config]
    # the ROM: owned by the shipped library's context, built like bench.py's
    eng = mm.Engine(0)
    hip = C.CDLL("libamdhip64.so")
    ptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(ptr), C.c_size_t(nbytes + 64)) == 0
    eng.attach(ptr.value, nbytes)
    mm.synth.RomSpec(42, nbytes, kw, elem, wc or None, be, BLOCK).apply_device(eng)
    eng.scan(mm.plan_relative(elem, kw, wc), block_bytes=BLOCK, big_endian=be)
    paths = {os.path.basename(p)[len("libmmoore_hip_"):-3]: p for p in sorted(glob.glob(os.path.join(ROOT, "tools", "ab", "libmmoore_hip_*.so")))}
    paths["cur"] = mm.LIB_PATH
    tags = args.tags or list(paths)
    variants = []
    for tag in tags:
        v = Variant(tag, paths[tag], mm)
        v.set_plan(elem, kw, wc)
        v.attach(ptr.value, nbytes)
        variants.append(v)
    print("# filter A/B: %s (%d GiB, %d-bit, '%s'), %d rounds, one synchronous scan per variant and round, variants: %s" % (
        args.config, nbytes >> 30, 8 * elem, kw, args.rounds, " ".join(tags)), flush=True)
    t_end = time.time() + 1.0
    while time.time() < t_end:                                # clocks up
        for v in variants:
            v.scan(be)
    filt = {v.tag: [] for v in variants}
    tot = {v.tag: [] for v in variants}
    counts = {}
    for r in range(args.rounds):
        order = variants if r % 2 == 0 else variants[::-1]    # nobody always runs behind the same neighbour
        for v in order:
            f, t, n = v.scan(be)
            filt[v.tag].append(f)
            tot[v.tag].append(t)
            counts[v.tag] = n
    assert len(set(counts.values())) == 1, counts
    base = float(np.median(filt[tags[0]]))
    for tag in tags:
        f, t = np.array(filt[tag]), np.array(tot[tag])
        print("%-8s streaming kernel: median %.4f ms  mean %.4f  min %.4f  (%.0f GB/s, %.3f of 8 TB/s; %+.2f %% against %s) | device time of the scan: median %.4f ms | %d matches" % (
            tag, np.median(f), f.mean(), f.min(), nbytes / np.median(f) / 1e6, nbytes / np.median(f) / 1e6 / 8000,
            (np.median(f) / base - 1) * 100, tags[0], np.median(t), counts[tag]), flush=True)


if __name__ == "__main__":
    main()
