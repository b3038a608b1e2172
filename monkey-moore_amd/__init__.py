# SPDX-License-Identifier: GPL-3.0-or-later
"""monkey-moore_amd -- MI355X-native relative-search engine (host-side Python binding).

Thin ctypes layer over the C ABI of include/mmoore_hip.h (libmmoore_hip.so).
PyTorch is not needed here; bench.py uses it only for device memory, streams and
torch.distributed plumbing.  There is no CPU fallback: without the built HIP
library, or without a GPU, the device entry points raise.

The directory name contains a hyphen, so import it with `load_package()` from
`__graft_entry__` / tests/conftest.py (importlib, module name `monkey_moore_amd`).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build
from . import build, synth  # noqa: F401

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = _build.CAPI_SO

MMH_MAX_KEYWORD = 128
MMH_OK, MMH_E_ARG, MMH_E_PLAN, MMH_E_DEVICE, MMH_E_CAPACITY, MMH_E_STATE, MMH_E_ABORTED = 0, -1, -2, -3, -4, -5, -6

EXPORTS = [
    "mmh_last_error", "mmh_plan_relative", "mmh_plan_value_scan", "mmh_device_count", "mmh_create", "mmh_destroy",
    "mmh_set_stream", "mmh_rom_upload", "mmh_rom_attach", "mmh_rom_download", "mmh_rom_alloc", "mmh_rom_synth",
    "mmh_rom_poke", "mmh_rom_fill", "mmh_scan", "mmh_set_engine", "mmh_last_timings", "mmh_last_counters",
    "mmh_timing_history", "mmh_filter_shape", "mmh_rom_load_file", "mmh_last_load_stats", "mmh_rom_gather",
    "mmh_scan_submit", "mmh_scan_collect", "mmh_rom_load_file_watched",
    "mmh_partition", "mmh_comm_unique_id", "mmh_comm_init_rank", "mmh_comm_init_all", "mmh_comm_info", "mmh_comm_destroy",
    "mmh_gather_start", "mmh_gather_finish", "mmh_last_gather_timings", "mmh_scan_multi", "mmh_selftest_gather_pack",
    "mmh_set_route", "mmh_set_timing", "mmh_health", "mmh_selftest_kat", "mmh_selftest_run", "mmh_debug_inject", "mmh_selftest_read_probe",
]
ROUTE_NO_SINGLE_LAUNCH, ROUTE_NO_ZERO_COPY, ROUTE_NO_BUCKETS, ROUTE_NO_POLLED, ROUTE_NO_SPLIT = 1, 2, 4, 8, 16
FB_NONE, FB_HEADER, FB_CAPACITY, FB_STALE_SLOT, FB_ORDER, FB_RANGE, FB_SELFTEST = range(7)
MMH_GATHER_RECORD_WORDS = 8 + 16384
MMH_MAX_IN_FLIGHT = 3
MMH_COMM_ID_BYTES = 128


class PlanDesc(C.Structure):
    """mmh_plan_desc (include/mmoore_hip.h)."""
    _fields_ = [
        ("elem_bytes", C.c_uint32), ("mode", C.c_uint32), ("L", C.c_uint32), ("match_jump", C.c_uint32),
        ("lead_wildcards", C.c_uint32), ("first_literal", C.c_uint32), ("default_skip", C.c_int32),
        ("n_skip", C.c_uint32),
        ("expected", C.c_int32 * MMH_MAX_KEYWORD), ("cmp_mask", C.c_uint32 * MMH_MAX_KEYWORD),
        ("skip_diff", C.c_int32 * MMH_MAX_KEYWORD), ("skip_val", C.c_int32 * MMH_MAX_KEYWORD),
        ("bridge", C.c_int8 * MMH_MAX_KEYWORD), ("wst", C.c_uint8 * MMH_MAX_KEYWORD),
    ]


class MMError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (code %d)" % (msg, code))
        self.code = code


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels ship their own libamdhip64.so / librccl.so under the same SONAMEs as the
    system ROCm.  Whichever copy is loaded first serves the whole process: loading this library
    first would hand torch the SYSTEM runtime, under which its kernels find no device.  With
    MMOORE_PRELOAD_TORCH_HIP=1 the torch copies are loaded (globally) before libmmoore_hip.so, so
    that either import order works; without it, import torch first (as bench.py does)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("MMOORE_PRELOAD_TORCH_HIP", "0") in ("", "0"):
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libamdhip64.so", "librccl.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """Load libmmoore_hip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMError(MMH_E_DEVICE, "libmmoore_hip.so is not built (run __graft_entry__.build()): " + LIB_PATH)
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        u32p, u64p, i16p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int16)
        L.mmh_last_error.restype = C.c_char_p
        L.mmh_plan_relative.argtypes = [C.c_uint32, u32p, C.c_uint32, C.c_uint32, u32p, C.c_uint32, C.POINTER(PlanDesc)]
        L.mmh_plan_value_scan.argtypes = [C.c_uint32, i16p, C.c_uint32, C.POINTER(PlanDesc)]
        L.mmh_device_count.argtypes = [C.POINTER(C.c_int)]
        L.mmh_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.mmh_destroy.argtypes = [C.c_void_p]
        L.mmh_destroy.restype = None
        L.mmh_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.mmh_set_engine.argtypes = [C.c_void_p, C.c_int]
        L.mmh_rom_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.mmh_rom_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.mmh_rom_download.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.mmh_rom_alloc.argtypes = [C.c_void_p, C.c_uint64]
        L.mmh_rom_synth.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.mmh_rom_poke.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.mmh_rom_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
        L.mmh_scan.argtypes = [C.c_void_p, C.POINTER(PlanDesc), C.c_uint64, C.c_int, C.c_uint64, u64p, C.c_uint64, u64p]
        L.mmh_scan_submit.argtypes = [C.c_void_p, C.POINTER(PlanDesc), C.c_uint64, C.c_int, C.c_uint64, C.POINTER(C.c_int)]
        L.mmh_scan_collect.argtypes = [C.c_void_p, C.c_int, u64p, C.c_uint64, u64p]
        L.mmh_last_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.mmh_last_counters.argtypes = [C.c_void_p, u64p]
        L.mmh_filter_shape.argtypes = [C.POINTER(PlanDesc), C.POINTER(C.c_uint32)]
        L.mmh_rom_load_file.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_int]
        L.mmh_rom_load_file_watched.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_int32), u64p]
        L.mmh_last_load_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), u64p, C.POINTER(C.c_int)]
        L.mmh_rom_gather.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_uint32, C.c_void_p]
        L.mmh_timing_history.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
        L.mmh_partition.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, u64p, u64p]
        L.mmh_comm_unique_id.argtypes = [C.c_void_p]
        L.mmh_comm_init_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.mmh_comm_init_all.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.mmh_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mmh_comm_destroy.argtypes = [C.c_void_p]
        L.mmh_comm_destroy.restype = None
        L.mmh_gather_start.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_int]
        L.mmh_gather_finish.argtypes = [C.c_void_p, u64p, C.c_uint64, u64p]
        L.mmh_last_gather_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.mmh_selftest_gather_pack.argtypes = [C.c_void_p, u64p, C.c_int, u64p, C.c_uint64, u64p, u64p]
        L.mmh_scan_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(PlanDesc), C.c_uint64, C.c_int, u64p, u64p, C.c_uint64, u64p]
        L.mmh_set_route.argtypes = [C.c_void_p, C.c_uint32]
        L.mmh_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.mmh_health.argtypes = [C.c_void_p, u64p]
        L.mmh_selftest_kat.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, u64p, u64p, C.c_uint64, u64p]
        L.mmh_selftest_run.argtypes = [C.c_int, u32p]
        L.mmh_debug_inject.argtypes = [C.c_void_p, C.c_uint32]
        L.mmh_selftest_read_probe.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _check(rc):
    if rc != MMH_OK:
        raise MMError(rc, lib().mmh_last_error().decode("utf-8", "replace"))


def _codepoints(s):
    if s is None:
        return np.zeros(0, np.uint32)
    if isinstance(s, str):
        return np.array([ord(ch) for ch in s], dtype=np.uint32)
    return np.array(list(s), dtype=np.uint32)


def _p(a, typ):
    return a.ctypes.data_as(C.POINTER(typ)) if a.size else C.cast(None, C.POINTER(typ))


def plan_relative(elem_bytes, keyword, wildcard=0, char_seq=None):
    """MonkeyMoore<Ty>(keyword, wildcard, char_seq) -> flattened plan (host only)."""
    kw, seq = _codepoints(keyword), _codepoints(char_seq)
    d = PlanDesc()
    _check(lib().mmh_plan_relative(elem_bytes, _p(kw, C.c_uint32), len(kw), int(wildcard), _p(seq, C.c_uint32), len(seq), C.byref(d)))
    return d


def plan_value_scan(elem_bytes, values):
    """MonkeyMoore<Ty>(reference_values) -> flattened plan (host only)."""
    v = np.array(list(values), dtype=np.int16)
    d = PlanDesc()
    _check(lib().mmh_plan_value_scan(elem_bytes, _p(v, C.c_int16), len(v), C.byref(d)))
    return d


def filter_shape(plan):
    """How the streaming filter keys on a plan (host only): number of SWAR conditions, anchor,
    kernel shape, where survivors are verified, (keyword position, gap) of every condition."""
    info = (C.c_uint32 * 12)()
    _check(lib().mmh_filter_shape(C.byref(plan), info))
    n = info[0]
    return {"ncond": n, "anchor": info[1], "shape": info[2], "verify_in_filter": bool(info[3]),
            "conditions": [(info[4 + 2 * k], info[5 + 2 * k]) for k in range(n)]}


def partition_range(total_bytes, block_bytes, keyword_len, elem_bytes, rank, nranks):
    """(first_byte, nbytes) of rank's block-aligned partition incl. the pattern-length overlap (host only)."""
    first, n = C.c_uint64(0), C.c_uint64(0)
    _check(lib().mmh_partition(total_bytes, block_bytes, keyword_len, elem_bytes, rank, nranks, C.byref(first), C.byref(n)))
    return first.value, n.value


def comm_unique_id():
    """Rank 0: the id every rank passes to Engine.comm_init_rank (distribute it through the launcher's rendezvous)."""
    buf = (C.c_uint8 * MMH_COMM_ID_BYTES)()
    _check(lib().mmh_comm_unique_id(buf))
    return bytes(buf)


def comm_init_all(engines):
    """One process, several GPUs: engine i becomes rank i of one communicator."""
    arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
    _check(lib().mmh_comm_init_all(arr, len(engines)))


def scan_multi(engines, plan, block_bytes, base_offsets, big_endian=False, cap=1 << 16):
    """mmh_scan_multi: every engine scans its attached partition, the lists are gathered over RCCL."""
    arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
    bases = np.ascontiguousarray(base_offsets, dtype=np.uint64)
    count = C.c_uint64(0)
    while True:
        out = np.empty(cap, np.uint64)
        rc = lib().mmh_scan_multi(arr, len(engines), C.byref(plan), block_bytes, int(big_endian), _p(bases, C.c_uint64),
                                  _p(out, C.c_uint64), out.size, C.byref(count))
        if rc == MMH_E_CAPACITY:
            cap = int(count.value) + 16
            continue
        _check(rc)
        return out[: count.value].copy()


def selftest_kat():
    """(rom bytes, keyword, block_bytes, expected offsets) of the library's first-use self-test (host only)."""
    rom = np.zeros(8192, np.uint8)
    exp = np.zeros(64, np.uint64)
    nb, ne = C.c_uint64(0), C.c_uint64(0)
    _check(lib().mmh_selftest_kat(rom.ctypes.data_as(C.POINTER(C.c_uint8)), rom.size, C.byref(nb), _p(exp, C.c_uint64), exp.size, C.byref(ne)))
    return rom[: nb.value].copy(), "abcde", 1024, exp[: ne.value].copy()


def selftest_run(device=0):
    """Runs the known-answer self-test on a device; returns the MMH_ROUTE_* mask it would switch off (0: all routes fine)."""
    off = C.c_uint32(0)
    _check(lib().mmh_selftest_run(device, C.byref(off)))
    return off.value


def device_count():
    n = C.c_int(0)
    rc = lib().mmh_device_count(C.byref(n))
    return n.value if rc == MMH_OK else 0


class Engine:
    """One device context: a ROM resident in HBM plus scan workspace."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().mmh_create(device, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().mmh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- ROM ------------------------------------------------------------
    def upload(self, data):
        a = np.ascontiguousarray(data)
        _check(lib().mmh_rom_upload(self._h, a.ctypes.data, a.nbytes))

    def attach(self, device_ptr, nbytes):
        _check(lib().mmh_rom_attach(self._h, C.c_void_p(device_ptr), nbytes))

    def alloc(self, nbytes):
        _check(lib().mmh_rom_alloc(self._h, nbytes))

    def synth(self, seed, base_offset=0):
        _check(lib().mmh_rom_synth(self._h, seed, base_offset))

    def poke(self, first_byte, data):
        a = np.ascontiguousarray(data, dtype=np.uint8)
        _check(lib().mmh_rom_poke(self._h, first_byte, a.ctypes.data, a.nbytes))

    def fill(self, first_byte, nbytes, value, ramp=0):
        _check(lib().mmh_rom_fill(self._h, first_byte, nbytes, value, ramp))

    def download(self, first_byte, nbytes):
        out = np.empty(nbytes, np.uint8)
        _check(lib().mmh_rom_download(self._h, first_byte, out.ctypes.data, nbytes))
        return out

    def load_file(self, path, file_offset, nbytes, threads=0, abort_word=None, bytes_done=None):
        """ROM <- bytes [file_offset, file_offset + nbytes) of a file (parallel readers + overlapped copies).
        abort_word / bytes_done: ctypes c_int32 / c_uint64 another thread raises / polls (mmh_rom_load_file_watched);
        an aborted load raises MMError with code MMH_E_ABORTED."""
        if abort_word is None and bytes_done is None:
            _check(lib().mmh_rom_load_file(self._h, os.fsencode(path), file_offset, nbytes, threads))
        else:
            _check(lib().mmh_rom_load_file_watched(self._h, os.fsencode(path), file_offset, nbytes, threads,
                                                   C.byref(abort_word) if abort_word is not None else None,
                                                   C.byref(bytes_done) if bytes_done is not None else None))
        s, b, t = C.c_double(0), C.c_uint64(0), C.c_int(0)
        _check(lib().mmh_last_load_stats(self._h, C.byref(s), C.byref(b), C.byref(t)))
        return {"seconds": s.value, "bytes": b.value, "threads": t.value}

    def gather(self, rom_offsets, bytes_each):
        """bytes_each bytes at every ROM offset, as an (n, bytes_each) uint8 array."""
        offs = np.ascontiguousarray(rom_offsets, dtype=np.uint64)
        out = np.zeros((offs.size, bytes_each), np.uint8)
        _check(lib().mmh_rom_gather(self._h, _p(offs, C.c_uint64), offs.size, bytes_each, out.ctypes.data))
        return out

    def set_stream(self, hip_stream):
        _check(lib().mmh_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_engine(self, engine):
        _check(lib().mmh_set_engine(self._h, engine))

    # -- scan -----------------------------------------------------------
    def _result_buffer(self, cap):
        # one result buffer reused from scan to scan; its ctypes pointer is cached (ndarray.ctypes
        # builds a new object on every access, which costs more than the call itself)
        cap = max(cap, getattr(self, "_long", 0))            # (a long list was handed out last time: room for one like it)
        out = getattr(self, "_out", None)
        if out is None or out.size < cap:
            out = self._out = np.empty(cap, np.uint64)
            self._out_ptr = out.ctypes.data_as(C.POINTER(C.c_uint64))
            self._count = C.c_uint64(0)
            self._count_ref = C.byref(self._count)
        return out

    def scan(self, plan, block_bytes=0, big_endian=False, base_offset=0, cap=1 << 16):
        """Returns ascending np.uint64 offsets (element indices when block_bytes == 0)."""
        scan = lib().mmh_scan
        plan_ref = C.byref(plan)
        while True:
            out = self._result_buffer(cap)
            rc = scan(self._h, plan_ref, block_bytes, int(big_endian), base_offset, self._out_ptr, out.size, self._count_ref)
            if rc == MMH_E_CAPACITY:
                cap = int(self._count.value) + 16
                continue
            if rc != MMH_OK:
                _check(rc)
            return self._take(out)

    def _take(self, out):
        """The call's offsets: a copy of the reusable buffer's head -- or, for a long list, the buffer itself (the next
        call gets a new one: copying 100 MB a second time costs more than the allocation)."""
        n = self._count.value
        if n > (1 << 18):
            self._out = None
            self._long = n + 16
            return out[:n]
        self._long = 0
        return out[:n].copy()

    def submit(self, plan, block_bytes=0, big_endian=False, base_offset=0):
        """Enqueue a scan (at most MMH_MAX_IN_FLIGHT = 3 outstanding); returns the ticket for collect()."""
        t = C.c_int(0)
        _check(lib().mmh_scan_submit(self._h, C.byref(plan), block_bytes, int(big_endian), base_offset, C.byref(t)))
        return t.value

    def collect(self, ticket, cap=1 << 16):
        """Wait for a submitted scan; returns what scan() would have returned."""
        while True:
            out = self._result_buffer(cap)
            rc = lib().mmh_scan_collect(self._h, ticket, self._out_ptr, out.size, self._count_ref)
            if rc == MMH_E_CAPACITY:
                cap = int(self._count.value) + 16
                continue
            if rc != MMH_OK:
                _check(rc)
            return self._take(out)

    # -- multi-GPU ------------------------------------------------------
    def comm_init_rank(self, unique_id, nranks, rank):
        """One process per GPU: join the communicator (collective over all ranks)."""
        buf = (C.c_uint8 * MMH_COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(lib().mmh_comm_init_rank(self._h, buf, nranks, rank))

    def comm_info(self):
        r, n = C.c_int(0), C.c_int(0)
        _check(lib().mmh_comm_info(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def gather_start(self, offsets=None, want_list=True):
        """Enqueue the RCCL gather of the offset lists: offsets None = the last scan's list, straight from HBM."""
        if offsets is None:
            _check(lib().mmh_gather_start(self._h, C.cast(None, C.POINTER(C.c_uint64)), 0, int(want_list)))
        else:
            a = np.ascontiguousarray(offsets, dtype=np.uint64)
            n = a.size
            if n == 0:
                a = np.zeros(1, np.uint64)                   # (NULL, 0) would mean "the last scan's list": an empty list needs a pointer
            self._gather_keep = a                            # stays alive until the copy is enqueued (synchronous for pageable memory)
            _check(lib().mmh_gather_start(self._h, a.ctypes.data_as(C.POINTER(C.c_uint64)), n, int(want_list)))

    def gather_finish(self, want_list=True, cap=1 << 16):
        """Wait for the oldest outstanding gather; the merged ascending list (or just its length when not want_list)."""
        count = C.c_uint64(0)
        if not want_list:
            _check(lib().mmh_gather_finish(self._h, C.cast(None, C.POINTER(C.c_uint64)), 0, C.byref(count)))
            return int(count.value)
        while True:
            out = getattr(self, "_gout", None)
            if out is None or out.size < cap:
                out = self._gout = np.empty(cap, np.uint64)
            rc = lib().mmh_gather_finish(self._h, _p(out, C.c_uint64), out.size, C.byref(count))
            if rc == MMH_E_CAPACITY:
                cap = int(count.value) + 16
                continue
            if rc != MMH_OK:
                _check(rc)
            return out[: count.value].copy()

    def selftest_gather_pack(self, records):
        """records: (nranks, MMH_GATHER_RECORD_WORDS) uint64 -- the table an all-gather would have left; returns
        (merged list or None when the long-list phase would follow, longest list)."""
        rec = np.ascontiguousarray(records, dtype=np.uint64)
        out = np.empty(rec.shape[0] * 16384, np.uint64)
        n, longest = C.c_uint64(0), C.c_uint64(0)
        _check(lib().mmh_selftest_gather_pack(self._h, _p(rec, C.c_uint64), rec.shape[0], _p(out, C.c_uint64), out.size,
                                              C.byref(n), C.byref(longest)))
        return (out[: n.value].copy() if longest.value <= 16384 else None), int(longest.value)

    def gather_timings(self):
        t = (C.c_float * 2)()
        _check(lib().mmh_last_gather_timings(self._h, t))
        return dict(device_ms=t[0], host_ms=t[1])

    def timings(self):
        t = (C.c_float * 4)()
        _check(lib().mmh_last_timings(self._h, t))
        # (parts > 0: the scan ran as a pipeline of that many parts; filter_ms is then the SUM of their overlapping streaming kernels
        # and total_ms the pipeline's wall time, see include/mmoore_hip.h)
        return dict(filter_ms=t[0], post_filter_ms=t[1], total_ms=t[3], parts=int(t[2]))

    def timing_history(self, n=64):
        """(filter_ms, total_ms) arrays of the last <= n scans, oldest first."""
        f, t, k = (C.c_float * n)(), (C.c_float * n)(), C.c_int(0)
        _check(lib().mmh_timing_history(self._h, f, t, n, C.byref(k)))
        return np.array(f[: k.value]), np.array(t[: k.value])

    def set_timing(self, on):
        """False: scans record no start events (~4.5 us less per scan); timings() then reports 0 except for single-launch scans."""
        _check(lib().mmh_set_timing(self._h, int(bool(on))))

    def set_route(self, mask):
        """Switch fast routes off on this context (ROUTE_* bits; 0 = all on)."""
        _check(lib().mmh_set_route(self._h, mask))

    def inject(self, kind):
        """Tests: damage the next polled scan's published block on the host (mmh_debug_inject)."""
        _check(lib().mmh_debug_inject(self._h, kind))

    def health(self):
        h = (C.c_uint64 * 16)()
        _check(lib().mmh_health(self._h, h))
        return dict(fallback_reason=int(h[0]), fallbacks=int(h[1]), late_slots=int(h[2]), last_reason=int(h[3]), routes_off=int(h[4]),
                    process_routes_off=int(h[5]), selftest=int(h[6]), validated=int(h[7]), header=[int(v) for v in h[8:16]])

    def read_probe(self, reps=20):
        """Pure-read passes over the ROM (mmh_selftest_read_probe): the box's measured HBM read ceiling."""
        best, mean, ms = C.c_double(0), C.c_double(0), C.c_double(0)
        _check(lib().mmh_selftest_read_probe(self._h, reps, C.byref(best), C.byref(mean), C.byref(ms)))
        return dict(mean_GBps=mean.value, best_GBps=best.value, ms_per_pass=ms.value, passes=reps,
                    what="pure-read kernels over the same ROM (grid-stride sweeps of 5 / 6 / 8 workgroups per CU and 64 KiB wave spans, 16-byte loads), "
                         "HIP events per pass, the best pattern's mean")

    def counters(self):
        c = (C.c_uint64 * 4)()
        _check(lib().mmh_last_counters(self._h, c))
        return dict(candidates=int(c[0]), matches=int(c[1]), tiles_walked=int(c[2]), path=int(c[3]))
