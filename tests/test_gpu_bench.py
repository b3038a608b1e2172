# SPDX-License-Identifier: GPL-3.0-or-later
"""bench.py on the GPU box the way the driver runs it: the default line carries every BASELINE
configuration (`other_configs`: C3, C4, C4BE with parity against the oracle), `--config` selects
any of them as the timed one, and `--gpus N` beyond the node's GPUs is refused, not faked."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

BENCH = os.path.join(ROOT, "bench.py")


def _bench(args, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_default_line_carries_the_other_baseline_configs():
    r, lines = _bench(["--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert len(lines) == 1
    res = lines[0]
    assert res["n_gpus"] == 1 and res["config"]["name"] == "C2" and res["dtype"] == "u8"
    # (what the line says is checked for shape and consistency here; how fast the box was is the line's business --
    # only a fraction above the data-sheet peak would be a wrong clock, whatever the box)
    assert res["roofline"]["kernel"] == "mm_filter_u8<4>" and 0.0 < res["roofline"]["frac"] < 1.0
    assert "valu_lane_ops_per_byte" in res["roofline"] and "lds_lane_reads_per_byte" in res["roofline"]
    assert "synchronous.without_timing_events" in res["config"]["step"]
    other = res["other_configs"]
    assert sorted(other) == ["C1", "C3", "C4", "C4BE"]
    c1 = other.pop("C1")                                       # bench_search.cpp's own buffer and keywords, whole-buffer chain
    assert sorted(c1["keywords"]) == ["abcde", "monkey"]
    for kw, o in c1["keywords"].items():
        assert "identical to" in o["parity"] and o["kernel_ms"] > 0 and o["synchronous"]["ms_per_scan"] > 0, (kw, o)
        assert o["facade_search"]["ms_per_call"] > 0
    assert c1["keywords"]["abcde"]["matches"] + c1["keywords"]["monkey"]["matches"] == 0   # (what the reference finds there)
    for name, o in other.items():
        assert 0.0 < o["frac"] < 1.0, (name, o)
        assert "identical to the oracle" in o["parity"]
        assert o["path"] == 0 and o["matches"] >= 4096
        assert o["synchronous"]["ms_per_scan"] > 0 and o["kernel_ms"] > 0
    assert other["C3"]["kernel"].startswith("mm_filter_u8<") and other["C4"]["kernel"].startswith("mm_filter_u16<")


@pytest.mark.parametrize("name,dtype", [("C3", "u8"), ("C4", "u16")])
def test_config_flag_times_that_configuration(name, dtype):
    r, lines = _bench(["--config", name, "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--gib-per-gpu", "1"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = lines[0]
    assert res["dtype"] == dtype and res["value"] > 0 and "other_configs" not in res
    assert res["roofline"]["kernel"].startswith("mm_filter_" + dtype)
    assert res["roofline"]["traffic"] is None                 # the PMC passes are C2's kernel's
    assert res["synchronous"]["same_offsets"] is True


def test_config_flag_at_full_size_counts_its_own_traffic():
    """`--config C3` at BASELINE's size: roofline.traffic and the instruction mix are counted for C3's own kernel in this run
    (rocprofv3 --pmc children; where a box has no counters the line says why and carries null)."""
    r, lines = _bench(["--config", "C3", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    roof = lines[0]["roofline"]
    assert roof["kernel"].startswith("mm_filter_u8<")
    if roof["traffic"] is None:
        assert roof.get("traffic_not_counted_in_this_run"), roof
        return
    assert "counted in this run" in roof["traffic_source"] and "--config C3" in roof["traffic_source"]
    assert 0.9 < roof["traffic_over_algorithmic"] < 1.5, roof            # (counters, not clocks: every ROM byte is read once)
    assert roof["valu_lane_ops_per_byte"] and roof["valu_lane_ops_per_byte"] > 1.0
    assert roof["lds_lane_reads_per_byte"] is not None


def test_more_ranks_than_gpus_is_refused_not_faked():
    import torch
    have = torch.cuda.device_count()
    r, lines = _bench(["--gpus", str(have + 1), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode == 2 and not lines, (r.returncode, r.stdout[-500:])
    assert "only %d GPU(s) visible" % have in r.stderr
