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


def gather_offsets(offsets, rank, world, device, dist):
    """All ranks call this with their ascending uint64 offsets (already global).  Rank 0
    gets the concatenation in rank order (= globally ascending), the others get None.

    The payload is tiny and latency-bound (8 B per match), so the common case is ONE
    collective: an all_gather of fixed-width records [count, offsets...] (64 KiB per rank;
    on the 8-GPU xGMI mesh every peer is one hop away).  Every rank sees every count, so
    all ranks agree without further traffic on whether some list did not fit; only then a
    second, padded all_gather of the full lists follows."""
    import torch
    mine = np.ascontiguousarray(offsets).astype(np.int64)
    record = np.zeros(GATHER_WIDTH, np.int64)
    record[0] = mine.size
    k = min(mine.size, GATHER_WIDTH - 1)
    record[1:1 + k] = mine[:k]
    rec = torch.from_numpy(record).to(device)
    records = [torch.empty(GATHER_WIDTH, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(records, rec)
    counts = torch.stack([r[0] for r in records]).cpu().numpy()      # every rank: 8 B per peer
    if int(counts.max()) <= GATHER_WIDTH - 1:
        if rank != 0:
            return None
        table = torch.stack(records).cpu().numpy()
        return np.concatenate([table[r, 1:1 + counts[r]] for r in range(world)]).astype(np.uint64)
    width = int(counts.max())
    padded = torch.zeros(width, dtype=torch.int64, device=device)
    padded[: mine.size] = torch.from_numpy(mine).to(device)
    full = [torch.empty(width, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(full, padded)
    if rank != 0:
        return None
    return torch.cat([f[:c] for f, c in zip(full, counts.tolist())]).cpu().numpy().astype(np.uint64)
