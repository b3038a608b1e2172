"""Multi-GPU plumbing: block-aligned partitions of a ROM and the gather of the per-rank
offset lists.  torch.distributed only -- backend "nccl" (= RCCL over xGMI) on
GPUs, "gloo" in the CPU tests; the scan itself never needs a collective because every
reference block (x byte alignment) is an independent chain."""
import numpy as np


def shard_range(total_bytes, block_bytes, keyword_len, elem_bytes, rank, world):
    """(first_byte, nbytes) of rank's partition: whole blocks [rank*nb/world, (rank+1)*nb/world)
    plus the (L-1)*S bytes of pattern-length overlap into the next partition."""
    nblocks = -(-total_bytes // block_bytes)
    b0 = rank * nblocks // world
    b1 = (rank + 1) * nblocks // world
    first = b0 * block_bytes
    end = min(b1 * block_bytes + (keyword_len - 1) * elem_bytes, total_bytes)
    return first, max(end - first, 0)


def gather_offsets(offsets, rank, world, device, dist):
    """All ranks call this with their ascending uint64 offsets (already global).  Rank 0
    gets the concatenation in rank order (= globally ascending), the others get None."""
    import torch
    mine = torch.from_numpy(np.ascontiguousarray(offsets).astype(np.int64))
    n = torch.tensor([mine.numel()], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    padded = torch.zeros(width, dtype=torch.int64, device=device)
    padded[: mine.numel()] = mine.to(device)
    gathered = [torch.empty(width, dtype=torch.int64, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, gathered, dst=0)
    if rank != 0:
        return None
    return torch.cat([g[:c] for g, c in zip(gathered, counts)]).cpu().numpy().astype(np.uint64)
