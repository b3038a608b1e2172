# SPDX-License-Identifier: GPL-3.0-or-later
"""Synthetic ROMs for tests and bench.py.

A ROM is defined by (seed, nbytes, global base offset) plus a list of edits:
  * base bytes: byte i = byte (i & 7), little endian, of splitmix64 word (i >> 3)
    -- generated on the device by mm_synth_fill, on the host by the oracle's
    mmo_synth_fill or `splitmix_bytes` below (numpy);
  * runs: 1 MiB each of 0x00, 0xFF and a +1 ramp (real ROMs are full of padding);
  * plants: shifted copies of the keyword (no element wraps, so the plant is a
    match on the signed simple path too), wildcard slots keep the ROM's bytes.
    One per MiB, plus plants straddling every 64th block boundary and every
    1/8 partition boundary, plus -- for 16-bit -- plants at k*B-3, k*B-1, k*B+1.

The edits are plain data (offset, bytes); `apply_host` patches a numpy buffer,
`apply_device` patches the ROM held by an Engine.  Expected offsets are never
derived from the plant list: they come from the oracle.
"""
import numpy as np

MASK64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15


def bench_search_buffer(elem_bytes, nbytes, seed=42):
    """The reference benchmark's own input (benchmarks/bench_search.cpp:11-22, BASELINE config C1): std::mt19937(seed), one
    uniform_int_distribution<unsigned>(0, max(T)) draw per element -- with a 32-bit engine and a power-of-two range that is
    the draw's top 8 / 16 bits.  Little-endian bytes of the elements.  (Checked against the compiled reference's own
    generator in tests/test_oracle.py.)"""
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)                                   # init_genrand(seed), what std::mt19937(seed) does
    raw = bg.random_raw(nbytes // elem_bytes)
    if elem_bytes == 1:
        return (raw >> np.uint64(24)).astype(np.uint8)
    return (raw >> np.uint64(16)).astype("<u2").view(np.uint8)


def splitmix_word(seed, k):
    z = (seed + (k + 1) * GOLDEN) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def splitmix_bytes(seed, first_byte, nbytes):
    """numpy version of the device generator."""
    k0, k1 = first_byte >> 3, (first_byte + nbytes + 7) >> 3
    k = np.arange(k0, k1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (k + np.uint64(1)) * np.uint64(GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    b = z.view(np.uint8)
    lo = first_byte - (k0 << 3)
    return b[lo: lo + nbytes].copy()


class RomSpec:
    """Edits of one shard [base, base + nbytes) of a global ROM of total_bytes."""

    def __init__(self, seed, total_bytes, keyword, elem_bytes=1, wildcard=None, big_endian=False,
                 block_bytes=524288, base=0, nbytes=None, partitions=8, plants_per_mib=1, runs=True):
        self.seed, self.total, self.S = seed, total_bytes, elem_bytes
        self.kw = [None if (wildcard is not None and ord(c) == wildcard) else ord(c) for c in keyword]
        self.be, self.B = big_endian, block_bytes
        self.base = base
        self.nbytes = total_bytes - base if nbytes is None else nbytes
        self.partitions = partitions
        self.plants_per_mib = plants_per_mib
        self.with_runs = runs

    # -- what to write where (global offsets) -------------------------------
    def runs(self):
        out = []
        mib = 1 << 20
        if self.with_runs and self.total >= 8 * mib:
            out = [(3 * mib, mib, 0x00, 0), (5 * mib, mib, 0xFF, 0), (7 * mib, mib, 0, 1)]
        return out

    def plant_offsets(self):
        L, S, mib = len(self.kw), self.S, 1 << 20
        span = L * S
        offs = []
        nm = self.total // mib
        for m in range(nm * self.plants_per_mib):
            r = splitmix_word(self.seed ^ 0x5EED, m)
            offs.append((m // self.plants_per_mib) * mib + r % (mib - span))
        if self.total < mib and self.total > 4 * span:
            for m in range(4):
                offs.append(splitmix_word(self.seed ^ 0x5EED, m) % (self.total - span))
        k = 64
        while self.B and k * self.B < self.total:
            offs.append(k * self.B - span // 2)
            k += 64
        if self.B:
            nblocks = -(-self.total // self.B)
            for g in range(1, self.partitions):
                bnd = (g * nblocks // self.partitions) * self.B
                if span < bnd < self.total - span:
                    offs.append(bnd - 5 * S)
                    offs.append(bnd - span + S)          # ends just past the boundary's first element
            if S == 2:
                for kb in (1, 2, 3, 5):
                    for d in (-3, -1, 1):
                        o = kb * self.B + d
                        if span < o < self.total - span:
                            offs.append(o)
        return sorted(set(o for o in offs if 0 <= o <= self.total - span))

    def plant_bytes(self, o):
        """(byte offsets, byte values) of the plant at global offset o."""
        r = splitmix_word(self.seed ^ 0xB10C, o)
        lits = [v for v in self.kw if v is not None]
        hi = 0xFF if self.S == 1 else 0xFFFF
        lo_shift, hi_shift = -min(lits), hi - max(lits)
        shift = lo_shift + r % (hi_shift - lo_shift + 1)
        idx, val = [], []
        for j, v in enumerate(self.kw):
            if v is None:
                continue
            x = v + shift
            if self.S == 1:
                idx.append(o + j)
                val.append(x)
            else:
                b = x.to_bytes(2, "big" if self.be else "little")
                idx += [o + 2 * j, o + 2 * j + 1]
                val += [b[0], b[1]]
        return idx, val

    # -- apply ------------------------------------------------------------
    def _local(self, first, n):
        a, b = max(first, self.base), min(first + n, self.base + self.nbytes)
        return (a, b) if a < b else None

    def host_rom(self):
        rom = splitmix_bytes(self.seed, self.base, self.nbytes)
        for first, n, value, ramp in self.runs():
            loc = self._local(first, n)
            if loc:
                i = np.arange(loc[0] - first, loc[1] - first)
                rom[loc[0] - self.base: loc[1] - self.base] = ((value + (i & 0xFF) * ramp) & 0xFF).astype(np.uint8)
        for o in self.plant_offsets():
            idx, val = self.plant_bytes(o)
            for i, v in zip(idx, val):
                if self.base <= i < self.base + self.nbytes:
                    rom[i - self.base] = v
        return rom

    def apply_device(self, eng):
        """eng already holds an allocated ROM of self.nbytes bytes."""
        eng.synth(self.seed, self.base)
        for first, n, value, ramp in self.runs():
            loc = self._local(first, n)
            if loc:
                # ramp restarts every 256 bytes, so a clipped run keeps its phase when 256 | offset
                eng.fill(loc[0] - self.base, loc[1] - loc[0], (value + ((loc[0] - first) & 0xFF) * ramp) & 0xFF, ramp)
        for o in self.plant_offsets():
            idx, val = self.plant_bytes(o)
            pairs = [(i, v) for i, v in zip(idx, val) if self.base <= i < self.base + self.nbytes]
            # contiguous runs of bytes -> one poke each
            run_start, run = None, []
            for i, v in pairs + [(None, None)]:
                if run and (i is None or i != run_start + len(run)):
                    eng.poke(run_start - self.base, np.array(run, np.uint8))
                    run = []
                if i is not None:
                    if not run:
                        run_start = i
                    run.append(v)
