# SPDX-License-Identifier: GPL-3.0-or-later
"""Run by test_gpu_parity.py::test_offset_gather_on_rccl_world_of_one in a fresh process."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from conftest import load_package  # noqa: E402
import _gather_double  # noqa: E402

mm = load_package()
try:
    rng = np.random.default_rng(3)
    for n in (0, 1, 4223, _gather_double.GATHER_WIDTH - 1, _gather_double.GATHER_WIDTH + 5000):
        offs = np.sort(rng.choice(1 << 40, size=n, replace=False)).astype(np.uint64)
        got = _gather_double.gather_offsets(offs, 0, 1, dev, dist)
        assert got.dtype == np.uint64 and got.tolist() == offs.tolist(), n
    # the overlapped form bench.py uses at N > 1: start k, finish k-1
    g = _gather_double.OffsetGather(0, 1, dev, dist)
    lists = [np.sort(rng.choice(1 << 40, size=n, replace=False)).astype(np.uint64) for n in (4223, 0, 9000, 17, 4223)]
    done, pending = [], None
    for l in lists:
        h = g.start(l)
        if pending is not None:
            done.append(g.finish(pending))
        pending = h
    done.append(g.finish(pending))
    assert [d.tolist() for d in done] == [l.tolist() for l in lists]
    # and the engine next to torch, the way bench.py sets it up
    eng = mm.Engine(0)
    buf = torch.zeros((1 << 20) + 32, dtype=torch.uint8, device=dev)
    eng.attach(buf.data_ptr(), 1 << 20)
    eng.synth(7)
    assert len(eng.scan(mm.plan_relative(1, "ab"), block_bytes=65536)) > 0
    print("gather ok")
finally:
    dist.destroy_process_group()
