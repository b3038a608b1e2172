# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: host -> HBM rates of mmh_rom_upload (pageable memory) and mmh_rom_load_file."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
for mib in (16, 256, 1024, 4096):
    a = np.random.default_rng(1).integers(0, 256, mib << 20, dtype=np.uint8)
    eng.upload(a)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); eng.upload(a); best = min(best, time.perf_counter() - t0)
    path = "/dev/shm/_mm_upload_probe.bin"
    a.tofile(path)
    eng.load_file(path, 0, a.size)
    bf = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); eng.load_file(path, 0, a.size); bf = min(bf, time.perf_counter() - t0)
    os.unlink(path)
    print("%5d MiB: upload %.1f ms (%.1f GB/s) | load_file %.1f ms (%.1f GB/s)" % (mib, best * 1e3, a.size / best / 1e9, bf * 1e3, a.size / bf / 1e9), flush=True)
