# SPDX-License-Identifier: GPL-3.0-or-later
"""Run by test_gpu_multi.py::test_native_gather_next_to_torch_nccl in a fresh process: bench.py's
N > 1 set-up with a world of one."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29573")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from conftest import load_package  # noqa: E402
from _oracle import Oracle  # noqa: E402

mm = load_package()
try:
    box = [mm.comm_unique_id()]
    dist.broadcast_object_list(box, src=0)
    assert len(box[0]) == mm.MMH_COMM_ID_BYTES
    eng = mm.Engine(0)
    n, block, kw = 8 << 20, 524288, "relativesrch"
    buf = torch.zeros(n + 32, dtype=torch.uint8, device=dev)
    eng.attach(buf.data_ptr(), n)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    spec = mm.synth.RomSpec(42, n, kw, 1, None, False, block)
    spec.apply_device(eng)
    torch.cuda.synchronize()
    eng.comm_init_rank(box[0], 1, 0)
    assert eng.comm_info() == (0, 1)
    plan = mm.plan_relative(1, kw)
    orc = Oracle()
    want = orc.engine(orc.plan(1, kw), eng.download(0, n), block).tolist()
    pending, results = 0, []
    for _ in range(6):
        eng.scan(plan, block_bytes=block)
        eng.gather_start(None)
        pending += 1
        if pending == 2:
            results.append(eng.gather_finish())
            pending -= 1
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)                                  # torch's own communicator keeps working next to it
    results.append(eng.gather_finish())
    assert all(r.tolist() == want for r in results) and len(want) >= 8
    dist.barrier()
    eng.close()
    print("native gather ok", eng.__class__.__name__, len(want))
finally:
    dist.destroy_process_group()
