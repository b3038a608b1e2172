# SPDX-License-Identifier: GPL-3.0-or-later
"""ctypes bindings for the test-only checkers under oracle/.

`Oracle`  -> oracle/liboracle.so       (our C restatement, mm_oracle.c)
`Ref`     -> oracle/_ref/libmmref.so   (the unmodified reference core + our shim;
                                        prebuilt in the build container, optional)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libmmref.so")

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i16p = C.POINTER(C.c_int16)


def codepoints(s):
    """str / list of ints -> np.uint32 array of UTF-32 code points."""
    if isinstance(s, str):
        return np.array([ord(ch) for ch in s], dtype=np.uint32)
    return np.array(list(s), dtype=np.uint32)


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a is not None and a.size else C.cast(None, typ)


def build_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
        os.path.join(ORACLE_DIR, "mm_oracle.c")
    ):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, os.path.join(ORACLE_DIR, "liboracle.so")])


class OraclePlan:
    def __init__(self, lib, handle, elem_bytes):
        self.lib, self.h, self.elem_bytes = lib, handle, elem_bytes

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.mmo_plan_free(self.h)
            self.h = None

    @property
    def keyword_len(self):
        return self.lib.mmo_plan_keyword_len(self.h)

    @property
    def wildcard_path(self):
        return bool(self.lib.mmo_plan_is_wildcard_path(self.h))


class Oracle:
    def __init__(self):
        build_oracle()
        lib = C.CDLL(ORACLE_SO)
        lib.mmo_plan_relative.restype = C.c_void_p
        lib.mmo_plan_relative.argtypes = [C.c_int, u32p, C.c_int, C.c_uint32, u32p, C.c_int, C.c_char_p, C.c_int]
        lib.mmo_plan_value_scan.restype = C.c_void_p
        lib.mmo_plan_value_scan.argtypes = [C.c_int, i16p, C.c_int, C.c_char_p, C.c_int]
        lib.mmo_plan_free.argtypes = [C.c_void_p]
        lib.mmo_plan_keyword_len.argtypes = [C.c_void_p]
        lib.mmo_plan_is_wildcard_path.argtypes = [C.c_void_p]
        lib.mmo_search.restype = C.c_int64
        lib.mmo_search.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, u64p, C.c_uint64]
        lib.mmo_engine.restype = C.c_int64
        lib.mmo_engine.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, u64p, C.c_uint64]
        lib.mmo_values_map.restype = C.c_int
        lib.mmo_values_map.argtypes = [C.c_void_p, C.c_void_p, u32p, u32p, C.c_int]
        lib.mmo_synth_word.restype = C.c_uint64
        lib.mmo_synth_word.argtypes = [C.c_uint64, C.c_uint64]
        lib.mmo_synth_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        self.lib = lib

    def plan(self, elem_bytes, keyword, wildcard=0, char_seq=None):
        kw = codepoints(keyword)
        seq = codepoints(char_seq) if char_seq is not None else np.zeros(0, np.uint32)
        err = C.create_string_buffer(256)
        h = self.lib.mmo_plan_relative(elem_bytes, _ptr(kw, u32p), len(kw), int(wildcard), _ptr(seq, u32p), len(seq), err, 256)
        if not h:
            raise RuntimeError(err.value.decode())
        return OraclePlan(self.lib, h, elem_bytes)

    def plan_values(self, elem_bytes, values):
        v = np.array(values, dtype=np.int16)
        err = C.create_string_buffer(256)
        h = self.lib.mmo_plan_value_scan(elem_bytes, _ptr(v, i16p), len(v), err, 256)
        if not h:
            raise RuntimeError(err.value.decode())
        return OraclePlan(self.lib, h, elem_bytes)

    def search(self, plan, data):
        dt = np.uint8 if plan.elem_bytes == 1 else np.uint16
        data = np.ascontiguousarray(data, dtype=dt)
        cap = data.size + 16
        out = np.zeros(cap, np.uint64)
        n = self.lib.mmo_search(plan.h, data.ctypes.data, data.size, _ptr(out, u64p), cap)
        assert n <= cap
        return out[:n].copy()

    def engine(self, plan, file_bytes, block_size, big_endian=False):
        fb = np.ascontiguousarray(file_bytes, dtype=np.uint8)
        cap = max(16, fb.size + 2)
        out = np.zeros(cap, np.uint64)
        n = self.lib.mmo_engine(plan.h, fb.ctypes.data, fb.size, int(block_size), int(big_endian), _ptr(out, u64p), cap)
        assert n <= cap
        return out[:n].copy()

    def values_map(self, plan, data, pos):
        dt = np.uint8 if plan.elem_bytes == 1 else np.uint16
        data = np.ascontiguousarray(data, dtype=dt)
        keys = np.zeros(512, np.uint32)
        vals = np.zeros(512, np.uint32)
        n = self.lib.mmo_values_map(plan.h, data[int(pos):].ctypes.data, _ptr(keys, u32p), _ptr(vals, u32p), 512)
        return {int(k): int(v) for k, v in zip(keys[:n], vals[:n])}

    def synth(self, first_byte, nbytes, seed):
        buf = np.empty(nbytes, np.uint8)
        self.lib.mmo_synth_fill(buf.ctypes.data, first_byte, nbytes, seed)
        return buf


class Ref:
    """The compiled, unmodified reference (only where oracle/_ref/libmmref.so exists)."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        lib = C.CDLL(REF_SO)
        lib.mmref_last_error.restype = C.c_char_p
        lib.mmref_search.restype = C.c_int64
        lib.mmref_search.argtypes = [C.c_int, u32p, C.c_int, C.c_uint32, u32p, C.c_int, C.c_void_p, C.c_uint64, u64p, C.c_uint64]
        lib.mmref_value_scan.restype = C.c_int64
        lib.mmref_value_scan.argtypes = [C.c_int, i16p, C.c_int, C.c_void_p, C.c_uint64, u64p, C.c_uint64]
        lib.mmref_engine.restype = C.c_int64
        lib.mmref_engine.argtypes = [
            C.c_int, C.c_char_p, C.c_int, u32p, C.c_int, C.c_uint32, u32p, C.c_int, i16p, C.c_int,
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u64p, C.c_uint64,
        ]
        lib.mmref_result_map.argtypes = [C.c_int64, u32p, u32p, C.c_int]
        lib.mmref_result_preview.argtypes = [C.c_int64, C.c_char_p, C.c_int]
        lib.mmref_progress_info.argtypes = [C.POINTER(C.c_int)] * 3
        self.lib = lib

    def _err(self):
        return RuntimeError(self.lib.mmref_last_error().decode())

    def bench_data(self, elem_bytes, nbytes):
        """The reference benchmark's input buffer (bench_search.cpp:11-22), as raw bytes."""
        self.lib.mmref_bench_data.argtypes = [C.c_int, C.c_uint64, C.c_void_p]
        buf = np.empty(nbytes, np.uint8)
        self.lib.mmref_bench_data(elem_bytes, nbytes, buf.ctypes.data)
        return buf

    def search(self, elem_bytes, keyword, data, wildcard=0, char_seq=None):
        kw = codepoints(keyword)
        seq = codepoints(char_seq) if char_seq is not None else np.zeros(0, np.uint32)
        dt = np.uint8 if elem_bytes == 1 else np.uint16
        data = np.ascontiguousarray(data, dtype=dt)
        cap = max(16, data.size + 2)
        out = np.zeros(cap, np.uint64)
        n = self.lib.mmref_search(elem_bytes, _ptr(kw, u32p), len(kw), int(wildcard), _ptr(seq, u32p), len(seq),
                                  data.ctypes.data, data.size, _ptr(out, u64p), cap)
        if n < 0:
            raise self._err()
        return out[:n].copy()

    def value_scan(self, elem_bytes, values, data):
        v = np.array(values, dtype=np.int16)
        dt = np.uint8 if elem_bytes == 1 else np.uint16
        data = np.ascontiguousarray(data, dtype=dt)
        cap = max(16, data.size + 2)
        out = np.zeros(cap, np.uint64)
        n = self.lib.mmref_value_scan(elem_bytes, _ptr(v, i16p), len(v), data.ctypes.data, data.size, _ptr(out, u64p), cap)
        if n < 0:
            raise self._err()
        return out[:n].copy()

    def result_map(self, i):
        keys = np.zeros(512, np.uint32)
        vals = np.zeros(512, np.uint32)
        n = self.lib.mmref_result_map(i, _ptr(keys, u32p), _ptr(vals, u32p), 512)
        return {int(k): int(v) for k, v in zip(keys[:n], vals[:n])}

    def result_preview(self, i):
        buf = C.create_string_buffer(4096)
        self.lib.mmref_result_preview(i, buf, 4096)
        return buf.value.decode("utf-8")

    def progress_info(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.lib.mmref_progress_info(C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, bool(c.value)

    def engine(self, elem_bytes, file_bytes, keyword=None, wildcard=ord("*"), char_seq=None, values=None,
               big_endian=False, threads=1, block_size=524288, preview_width=50, previews=False,
               abort_after=0, path=None):
        kw = codepoints(keyword) if keyword is not None else np.zeros(0, np.uint32)
        seq = codepoints(char_seq) if char_seq is not None else np.zeros(0, np.uint32)
        rv = np.array(values if values is not None else [], dtype=np.int16)
        tmp = None
        if path is None:
            fb = np.ascontiguousarray(file_bytes, dtype=np.uint8)
            fd, tmp = tempfile.mkstemp(prefix="mmref_", suffix=".bin")
            with os.fdopen(fd, "wb") as f:
                f.write(fb.tobytes())
            path = tmp
            cap = fb.size + 16
        else:
            cap = min((os.path.getsize(path) if os.path.exists(path) else 0) + 16, 1 << 24)
        try:
            out = np.zeros(cap, np.uint64)
            n = self.lib.mmref_engine(elem_bytes, path.encode(), int(values is None), _ptr(kw, u32p), len(kw),
                                      int(wildcard), _ptr(seq, u32p), len(seq), _ptr(rv, i16p), len(rv),
                                      int(big_endian), threads, block_size, preview_width, int(previews),
                                      abort_after, _ptr(out, u64p), cap)
            if n < 0:
                raise self._err()
            if n > cap:
                raise RuntimeError("more matches than the binding's buffer holds")
            return out[:n].copy()
        finally:
            if tmp:
                os.unlink(tmp)


def oracle_engine_parallel(oracle, plan, rom, block_size, big_endian=False, workers=None):
    """Oracle engine over a big ROM using several host cores: the ROM is cut on block
    boundaries (every block x alignment is an independent chain), each slice --
    with its (L-1)*S bytes of overlap -- goes through mmo_engine on its own thread
    (ctypes releases the GIL), offsets are shifted and concatenated."""
    from concurrent.futures import ThreadPoolExecutor
    n = rom.size
    nblocks = -(-n // block_size)
    workers = workers or min(os.cpu_count() or 1, 64)
    per = max(1, -(-nblocks // workers))
    overlap = (plan.keyword_len - 1) * plan.elem_bytes

    def run(b0):
        first = b0 * block_size
        end = min((b0 + per) * block_size + overlap, n)
        return oracle.engine(plan, rom[first:end], block_size, big_endian) + np.uint64(first)

    with ThreadPoolExecutor(max_workers=workers) as ex:
        parts = list(ex.map(run, range(0, nblocks, per)))
    return np.concatenate(parts) if parts else np.zeros(0, np.uint64)
