# SPDX-License-Identifier: GPL-3.0-or-later
"""Test double of the multi-GPU plumbing (test infrastructure: nothing under monkey-moore_amd/ imports it).  The product
path is in the C ABI (csrc/mm_multi.hip: mmh_partition, mmh_comm_*, mmh_gather_start / _finish, mmh_scan_multi -- RCCL
called from the library itself, lists sent from HBM).  This module restates the same protocol on top of torch.distributed
so that the CPU suite can run it with "gloo" and two processes (tests/test_multi_gpu_host.py); bench.py uses it as the
CHECKER of the library's gather behind its timed region (`gather_check`) and, under --torch-gather, for diagnosis only
(never for a reported figure: a run whose native communicator does not come up prints `value: null` and exits 3)."""
import numpy as np

GATHER_WIDTH = 8192     # int64 words per rank in the fixed-width record: [count, offsets...]


def shard_range(total_bytes, block_bytes, keyword_len, elem_bytes, rank, world):
    """(first_byte, nbytes) of rank's partition: whole blocks [rank*nb/world, (rank+1)*nb/world)
    plus the (L-1)*S bytes of pattern-length overlap into the next partition."""
    nblocks = -(-total_bytes // block_bytes)
    b0 = rank * nblocks // world
    b1 = (rank + 1) * nblocks // world
    first = b0 * block_bytes
    if b1 == b0:
        return first, 0                                  # more ranks than blocks: nothing to scan here
    end = min(b1 * block_bytes + (keyword_len - 1) * elem_bytes, total_bytes)
    return first, max(end - first, 0)


class OffsetGather:
    """Gather of the per-rank offset lists to rank 0, split into start() / finish() so that the
    collective of one scan runs while the next scan is under way.

    The payload is tiny and latency-bound (8 B per match), so the common case is ONE
    collective: an all_gather of fixed-width records [count, offsets...] (64 KiB per rank; on
    the 8-GPU xGMI mesh every peer is one hop away) into one contiguous table, followed by ONE
    device-to-host copy per rank.  Every rank sees every count, so all ranks agree without
    further traffic on whether some list did not fit; only then a second, padded all_gather of
    the full lists follows (inside finish(), on every rank alike).

    Two staging sets alternate, so at most two gathers may be outstanding; finish them in the
    order they were started.  Backend: "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests."""

    def __init__(self, rank, world, device, dist, width=None):
        import torch
        self.rank, self.world, self.device, self.dist = rank, world, device, dist
        self.width = width or GATHER_WIDTH
        self.sets = []
        for _ in range(2):
            host = torch.zeros(self.width, dtype=torch.int64)
            if device.type == "cuda":
                host = host.pin_memory()
            self.sets.append(dict(host=host, view=host.numpy(),
                                  rec=torch.zeros(self.width, dtype=torch.int64, device=device),
                                  table=torch.zeros(world * self.width, dtype=torch.int64, device=device),
                                  work=None, mine=None))
        self.flat_ok = True
        self.turn = 0

    def _all_gather(self, b, async_op):
        if self.flat_ok:
            try:
                return self.dist.all_gather_into_tensor(b["table"], b["rec"], async_op=async_op)
            except (RuntimeError, NotImplementedError, AttributeError):
                self.flat_ok = False                     # backend without all_gather_into_tensor
        return self.dist.all_gather(list(b["table"].view(self.world, self.width).unbind(0)), b["rec"], async_op=async_op)

    def start(self, offsets, async_op=True):
        """All ranks: hand in this rank's ascending uint64 offsets (already global)."""
        b = self.sets[self.turn]
        if b["mine"] is not None:
            raise RuntimeError("two gathers are already outstanding: finish() the older one first")
        self.turn ^= 1
        mine = np.ascontiguousarray(offsets).astype(np.int64)
        k = min(mine.size, self.width - 1)
        b["view"][0] = mine.size
        b["view"][1:1 + k] = mine[:k]
        b["rec"].copy_(b["host"], non_blocking=True)      # stream ordered before the collective
        b["mine"] = mine
        b["work"] = self._all_gather(b, async_op)
        return b

    def finish(self, b):
        """All ranks, in start() order.  Rank 0 gets the concatenation in rank order (= globally
        ascending), the others None."""
        import torch
        if b["work"] is not None:
            b["work"].wait()
        mine, b["mine"], b["work"] = b["mine"], None, None
        width, world = self.width, self.world
        # one contiguous device-to-host copy on every rank (a strided read of the counts column
        # alone would cost a gather kernel plus the copy)
        host_table = b["table"].view(world, width).cpu().numpy()
        counts = host_table[:, 0].copy()
        if int(counts.max()) <= width - 1:
            if self.rank != 0:
                return None
            return np.concatenate([host_table[r, 1:1 + counts[r]] for r in range(world)]).astype(np.uint64)
        longest = int(counts.max())
        padded = torch.zeros(longest, dtype=torch.int64, device=self.device)
        padded[: mine.size] = torch.from_numpy(mine).to(self.device)
        full = [torch.empty(longest, dtype=torch.int64, device=self.device) for _ in range(world)]
        self.dist.all_gather(full, padded)
        if self.rank != 0:
            return None
        return torch.cat([f[:c] for f, c in zip(full, counts.tolist())]).cpu().numpy().astype(np.uint64)


_gatherers = {}


def gather_offsets(offsets, rank, world, device, dist):
    """Synchronous form: start() + finish() of a cached OffsetGather."""
    key = (rank, world, str(device), GATHER_WIDTH)
    g = _gatherers.get(key)
    if g is None:
        g = _gatherers[key] = OffsetGather(rank, world, device, dist)
    return g.finish(g.start(offsets, async_op=False))
