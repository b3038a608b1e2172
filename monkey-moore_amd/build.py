# SPDX-License-Identifier: GPL-3.0-or-later
"""Builds libmmoore_hip.so (HIP kernels + C ABI) and libmonkey-core.so (the C++17
facade with the reference's include/mmoore API) in-tree, for gfx950 only.

hipcc cross-compiles without a GPU, so this also runs in the build container.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIB_DIR = os.path.join(PKG, "lib")
CAPI_SO = os.path.join(LIB_DIR, "libmmoore_hip.so")
CORE_SO = os.path.join(LIB_DIR, "libmonkey-core.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
ROCM_LIB = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
ARCH = "gfx950"
SHAPE_UNITS = 7          # MM_SHAPE_UNITS of csrc/mm_filter_shapes.h


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def _stamp_matches(target, digest):
    """The built library next to a stamp file holding the hash of the sources it was built from: nothing to do.
    (File times do not survive the copy to the GPU box, and lib/obj/ does not travel at all: without the stamp every
    process there rebuilt the library -- two minutes of hipcc per pytest run.)"""
    try:
        with open(target + ".stamp") as f:
            return os.path.exists(target) and f.read().strip() == digest
    except OSError:
        return False


def _write_stamp(target, digest):
    with open(target + ".stamp", "w") as f:
        f.write(digest + "\n")


def build_capi(force=False):
    """libmmoore_hip.so: every csrc translation unit to its own object (in parallel), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB_DIR, exist_ok=True)
    digest = device_source_sha16()
    if not force and _stamp_matches(CAPI_SO, digest):
        return CAPI_SO
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    # (mm_filter_shapes.hip is compiled SHAPE_UNITS times, -DMM_FILTER_SHAPE_UNIT=k: the streaming kernels of one group of
    # filter shapes each -- in one unit they took hipcc four minutes, side by side the slowest takes one)
    units = [("mm_filter_shapes.hip", k) for k in range(SHAPE_UNITS)]
    units += [(name, None) for name in ("mm_kernels.hip", "mm_capi.hip", "mm_multi.hip", "mm_ingest.hip", "mm_sort.hip", "mm_probe.hip", "mm_plan.cpp")]
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "mmoore_hip.h")]
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]

    def compile_unit(unit):
        name, shape_unit = unit
        src = os.path.join(CSRC, name)
        obj = os.path.join(obj_dir, name + (".o" if shape_unit is None else ".%d.o" % shape_unit))
        if force or _newer(obj, [src] + headers):
            define = [] if shape_unit is None else ["-DMM_FILTER_SHAPE_UNIT=%d" % shape_unit]
            _run([HIPCC, *flags, *define, "-x", "hip", "-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_unit, units))
    if force or _newer(CAPI_SO, objs):
        # librccl: the multi-GPU offset gather (mm_multi.hip) calls RCCL itself
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-pthread", "-L" + ROCM_LIB, "-lrccl", "-o", CAPI_SO])
    _write_stamp(CAPI_SO, digest)
    return CAPI_SO


def build_core(force=False):
    """The C++17 facade (MonkeyMoore<T>, SearchEngine<T>) over the C ABI."""
    srcs = [os.path.join(HOST, f) for f in ("monkey_moore.cpp", "search_engine.cpp", "c_bindings.cpp")]
    if not all(os.path.exists(s) for s in srcs):
        return None
    inc = os.path.join(ROOT, "include", "mmoore")
    deps = srcs + [os.path.join(inc, f) for f in os.listdir(inc)] + [CAPI_SO]
    digest = _files_sha16(sorted(deps[:-1]) + [os.path.join(ROOT, "include", "mmoore_hip.h")]) + "-" + device_source_sha16()
    if not force and _stamp_matches(CORE_SO, digest):
        return CORE_SO
    if force or _newer(CORE_SO, deps) or not _stamp_matches(CORE_SO, digest):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(ROOT, "include"),
              *srcs, "-L" + LIB_DIR, "-lmmoore_hip", "-Wl,-rpath,$ORIGIN", "-o", CORE_SO])
    _write_stamp(CORE_SO, digest)
    return CORE_SO


def _files_sha16(paths):
    import hashlib
    h = hashlib.sha256()
    for path in paths:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def device_source_sha16():
    """SHA-256 (first 16 hex digits) over the sources libmmoore_hip.so is built from -- csrc/ and the C-ABI header --
    in file-name order.  profiles/rNN_pmc_summary.json carries it: bench.py reports the PMC traffic figure only
    while the device code is the code the counters were taken with (the binary's own hash differs from build
    directory to build directory)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".cpp")))
    files.append(os.path.join(ROOT, "include", "mmoore_hip.h"))
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_all(force=False):
    build_capi(force)
    build_core(force)
    return CAPI_SO


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
