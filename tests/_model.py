# SPDX-License-Identifier: GPL-3.0-or-later
"""Pure-Python model of what the HIP kernels compute from an mmh_plan_desc.

Used by the CPU test-suite to check, without a GPU, (a) the host plan builder
(monkey-moore_amd/csrc/mm_plan.cpp) and (b) the ALGORITHM of the device path --
candidate filter + phase-set certificate resolver -- against the oracle and the
golden vectors.  It mirrors mm_kernels.hip structurally (same tile / segment
decomposition, same pull-back), not the reference.
"""
import numpy as np


class Geom:
    def __init__(self, nbytes, S, L, block_bytes=0, big_endian=False):
        self.S, self.L, self.B = S, L, block_bytes
        self.whole = block_bytes == 0
        self.N = (nbytes // S) * S if self.whole else nbytes
        self.be = big_endian and S == 2
        self.nblocks = 1 if self.whole else -(-nbytes // block_bytes)

    def nv(self, b, p):
        if self.whole:
            return self.N // self.S - self.L + 1
        off = b * self.B
        full = self.B + (self.L - 1) * self.S
        size = min(full, self.N - off)
        count = size // self.S
        if p + count * self.S > size:
            count -= 1
        return count - self.L + 1

    def start(self, b, p):
        return 0 if self.whole else b * self.B + p

    def locate(self, o):
        if self.whole:
            if o % self.S:
                return None
            b, p, j = 0, 0, o // self.S
        else:
            b = o // self.B
            r = o - b * self.B
            p, j = r % self.S, r // self.S
        return (b, p, j) if j < self.nv(b, p) else None


def elem(rom, g, start, j):
    a = start + j * g.S
    if g.S == 1:
        return int(rom[a])
    lo, hi = int(rom[a]), int(rom[a + 1])
    return (lo << 8 | hi) if g.be else (hi << 8 | lo)


def skip_of(pl, d):
    s = pl.default_skip
    for k in range(pl.n_skip):
        if pl.skip_diff[k] == d:
            s = pl.skip_val[k]
    return s


def step(pl, rd, j):
    """(jump, matched) of the unified compare loop (mm_step in mm_kernels.hip)."""
    for i in range(pl.L - 1, -1, -1):
        d = rd(j + i) - rd(j + i + pl.bridge[i])
        if ((d ^ pl.expected[i]) & pl.cmp_mask[i]) & 0xFFFFFFFF:
            s = max(skip_of(pl, d), 1)
            return min(s, pl.wst[i]), False
    return pl.match_jump, True


def chain_seq(pl, rom, g, base_offset=0):
    """mm_chain_seq: walk every domain sequentially."""
    out = []
    for b in range(g.nblocks):
        for p in range(1 if g.whole else g.S):
            nv, st = g.nv(b, p), g.start(b, p)
            h = 0
            while h < nv:
                J, m = step(pl, lambda k: elem(rom, g, st, k), h)
                if m:
                    out.append(h if g.whole else st + h * g.S + base_offset)
                h += J
    return sorted(out)


def candidates(pl, rom, g):
    """What mm_filter_* must deliver: every valid alignment where the compare loop matches."""
    out = []
    for o in range(g.N):
        loc = g.locate(o)
        if loc is None:
            continue
        b, p, j = loc
        st = g.start(b, p)
        if step(pl, lambda k: elem(rom, g, st, k), j)[1]:
            out.append(o)
    return out


def resolve(pl, rom, g, o, tile=256, seg=16):
    """mm_resolve for one candidate: pull the acceptable-phase set back through tile maps."""
    b, p, jc = g.locate(o)
    st = g.start(b, p)
    D = pl.L - 1
    full = (1 << D) - 1
    A = 1 << (jc % D)
    hi = jc
    tiles = 0
    while True:
        if hi == 0:
            return bool(A & 1), tiles
        lo = ((hi - 1) // tile) * tile
        npos = hi - lo
        nseg = -(-npos // seg)
        maps = []
        for l in range(nseg):
            m = list(range(D))
            s0, s1 = l * seg, min((l + 1) * seg, npos)
            r = (lo + s0) % D
            for q in range(s0, s1):
                J, _ = step(pl, lambda k: elem(rom, g, st, k), lo + q)
                if J != D:
                    r2 = (r + J) % D
                    m = [r2 if v == r else v for v in m]
                r = (r + 1) % D
            maps.append(m)
        Anew = 0
        for e in range(D):
            v = e
            for m in maps:
                v = m[v]
            if (A >> v) & 1:
                Anew |= 1 << e
        tiles += 1
        if Anew == full:
            return True, tiles
        if Anew == 0:
            return False, tiles
        A, hi = Anew, lo


def fast_path(pl, rom, g, base_offset=0, tile=256, seg=16):
    out = []
    for o in candidates(pl, rom, g):
        ok, _ = resolve(pl, rom, g, o, tile, seg)
        if ok:
            out.append(o // g.S if g.whole else o + base_offset)
    return out
