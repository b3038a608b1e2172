"""Multi-GPU plumbing: block-aligned partitions of a ROM and the gather of the per-rank
offset lists.  torch.distributed only -- backend "nccl" (= RCCL over xGMI) on
GPUs, "gloo" in the CPU tests; the scan itself never needs a collective because every
reference block (x byte alignment) is an independent chain."""
import numpy as np

GATHER_WIDTH = 8192     # int64 words per rank in the fixed-width record: [count, offsets...]


def shard_range(total_bytes, block_bytes, keyword_len, elem_bytes, rank, world):
    """(first_byte, nbytes) of rank's partition: whole blocks [rank*nb/world, (rank+1)*nb/world)
    plus the (L-1)*S bytes of pattern-length overlap into the next partition."""
    nblocks = -(-total_bytes // block_bytes)
    b0 = rank * nblocks // world
    b1 = (rank + 1) * nblocks // world
    first = b0 * block_bytes
    end = min(b1 * block_bytes + (keyword_len - 1) * elem_bytes, total_bytes)
    return first, max(end - first, 0)


_buffers = {}


def _gather_buffers(world, device, width):
    """Staging reused from step to step: a (pinned, on GPUs) host record, its device copy and
    the world x width table the collective fills."""
    import torch
    key = (world, str(device), width)
    b = _buffers.get(key)
    if b is None:
        host = torch.zeros(width, dtype=torch.int64)
        if device.type == "cuda":
            host = host.pin_memory()
        b = _buffers[key] = dict(host=host, view=host.numpy(),
                                 rec=torch.zeros(width, dtype=torch.int64, device=device),
                                 table=torch.zeros(world * width, dtype=torch.int64, device=device), flat_ok=True)
    return b


def _all_gather(dist, b, world, width):
    """One collective into the contiguous table (falls back to the list form where the
    backend has no all_gather_into_tensor)."""
    if b["flat_ok"]:
        try:
            dist.all_gather_into_tensor(b["table"], b["rec"])
            return
        except (RuntimeError, NotImplementedError, AttributeError):
            b["flat_ok"] = False
    dist.all_gather(list(b["table"].view(world, width).unbind(0)), b["rec"])


def gather_offsets(offsets, rank, world, device, dist):
    """All ranks call this with their ascending uint64 offsets (already global).  Rank 0
    gets the concatenation in rank order (= globally ascending), the others get None.

    The payload is tiny and latency-bound (8 B per match), so the common case is ONE
    collective: an all_gather of fixed-width records [count, offsets...] (64 KiB per rank;
    on the 8-GPU xGMI mesh every peer is one hop away) into one contiguous table, followed by
    ONE device-to-host copy per rank.  Every
    rank sees every count, so all ranks agree without further traffic on whether some list did
    not fit; only then a second, padded all_gather of the full lists follows."""
    import torch
    width = GATHER_WIDTH
    mine = np.ascontiguousarray(offsets).astype(np.int64)
    b = _gather_buffers(world, device, width)
    k = min(mine.size, width - 1)
    b["view"][0] = mine.size
    b["view"][1:1 + k] = mine[:k]
    b["rec"].copy_(b["host"], non_blocking=True)          # stream ordered before the collective
    _all_gather(dist, b, world, width)
    # one contiguous device-to-host copy on every rank (a strided read of the counts column
    # alone would cost a gather kernel plus the copy)
    host_table = b["table"].view(world, width).cpu().numpy()
    counts = host_table[:, 0].copy()
    if int(counts.max()) <= width - 1:
        if rank != 0:
            return None
        return np.concatenate([host_table[r, 1:1 + counts[r]] for r in range(world)]).astype(np.uint64)
    longest = int(counts.max())
    padded = torch.zeros(longest, dtype=torch.int64, device=device)
    padded[: mine.size] = torch.from_numpy(mine).to(device)
    full = [torch.empty(longest, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(full, padded)
    if rank != 0:
        return None
    return torch.cat([f[:c] for f, c in zip(full, counts.tolist())]).cpu().numpy().astype(np.uint64)
