# SPDX-License-Identifier: GPL-3.0-or-later
"""bench.py's own launcher (CPU): `python bench.py --gpus N` starts its N ranks itself -- before
torch is imported or a GPU touched --, a world size that is not --gpus is refused, and a node
with fewer GPUs than ranks is refused instead of being oversubscribed."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout)


def _lines(out):
    """every JSON object in the text (the ranks of a dry launch write to one inherited pipe: two lines may arrive glued)"""
    found, dec, at = [], json.JSONDecoder(), 0
    while True:
        at = out.find("{", at)
        if at < 0:
            return found
        try:
            obj, end = dec.raw_decode(out, at)
        except json.JSONDecodeError:
            at += 1
            continue
        found.append(obj)
        at = end


def test_gpus_n_launches_n_ranks_by_itself():
    r = _run(["--gpus", "3", "--dry-launch", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    got = sorted(_lines(r.stdout), key=lambda d: d["rank"])
    assert [d["rank"] for d in got] == [0, 1, 2]
    assert [d["local_rank"] for d in got] == [0, 1, 2]
    assert all(d["world"] == 3 and d["gpus"] == 3 and d["dry_launch"] for d in got)
    assert "torch.distributed.run" in r.stderr and "--master-addr 127.0.0.1" in r.stderr


def test_single_gpu_default_does_not_spawn():
    r = _run(["--dry-launch"])
    assert r.returncode == 0
    got = _lines(r.stdout)
    assert len(got) == 1 and {k: got[0][k] for k in ("dry_launch", "rank", "local_rank", "world", "gpus")} == {
        "dry_launch": True, "rank": 0, "local_rank": 0, "world": 1, "gpus": 1}
    assert "launching" not in r.stderr


def test_world_size_that_is_not_gpus_is_refused():
    # the way round 2's bench would have lied: --gpus 8 under a world of 1 (or no launcher at all)
    r = _run(["--gpus", "8", "--dry-launch"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not _lines(r.stdout)
    r = _run(["--gpus", "1", "--dry-launch"], env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and not _lines(r.stdout)


def test_under_a_launcher_no_second_launch():
    r = _run(["--gpus", "2", "--dry-launch"], env={"RANK": "1", "WORLD_SIZE": "2", "LOCAL_RANK": "1"})
    assert r.returncode == 0 and "launching" not in r.stderr
    got = _lines(r.stdout)
    assert len(got) == 1 and {k: got[0][k] for k in ("dry_launch", "rank", "local_rank", "world", "gpus")} == {
        "dry_launch": True, "rank": 1, "local_rank": 1, "world": 2, "gpus": 2}


def test_more_ranks_than_gpus_is_refused():
    # (this container has no GPU at all; a 1-GPU box refuses --gpus 2 the same way: tests/test_gpu_multi.py)
    import torch
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 2), "--steps", "2"])
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "only %d GPU(s) visible" % have in r.stderr and "refusing" in r.stderr
    assert not _lines(r.stdout)


def test_strong_and_weak_partitions_for_1_2_4_8_ranks():
    """--scaling strong deals ONE ROM of the config's size over the ranks with mmh_partition (the reference's dispatcher
    deals one file over its workers, search_engine.cpp:66-188 / :218-253); weak scaling gives every rank a ROM of that
    size.  The dry launch prints both partitions of every rank: whole 512 KiB blocks, consecutive, (L-1)*S bytes of
    overlap into the next one, together the whole ROM."""
    for cfg, gib, L, S in (("C2", 4, 12, 1), ("C4", 8, 8, 2)):
        for n in (1, 2, 4, 8):
            r = _run(["--gpus", str(n), "--dry-launch", "--config", cfg, "--scaling", "strong"], timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            got = sorted(_lines(r.stdout), key=lambda d: d["rank"])
            assert [d["rank"] for d in got] == list(range(n)) and all(d["scaling"] == "strong" for d in got)
            block, overlap = 524288, (L - 1) * S
            for mode, total in (("strong", gib << 30), ("weak", n * (gib << 30))):
                parts = [d[mode] for d in got]
                assert all(p["total"] == total for p in parts)
                at = 0
                for i, p in enumerate(parts):
                    assert p["base"] == at and p["base"] % block == 0, (cfg, n, mode, i, p)
                    last = i == n - 1
                    own = p["bytes"] - (0 if last else overlap)          # what the next rank does not start in
                    assert own > 0 and own % block == 0 and own == total // n, (cfg, n, mode, i, p)
                    at += own
                assert at == total
            assert all(d["strong"]["overlap"] == overlap for d in got)
