"""bench.py keeps the driver's contract: ONE JSON line on stdout with the agreed keys, the roofline object of the
dominant kernel and (when enabled) the CPU baseline -- checked on a small frame so that it costs seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys(gpu):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
           "--gaussians", "30000", "--width", "640", "--height", "368", "--no-cpu-baseline", "--extras", "config2_300k"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "frames/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["config"]["schedule"] == "dropin"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]          # one view per step at N = 1
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algorithmic_bytes", "avg_ms"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(rf["achieved"] - rf["algorithmic_bytes"] / (rf["avg_ms"] * 1e-3) / 1e9) <= 1e-6 * rf["achieved"]
    assert d["batched"]["value"] > 0 and d["dropin"]["value"] == d["value"]
    assert d["config"]["capacity_misses"] == 0 and len(d["kernel_source_sha"]) == 16
    ex = d["extra"]["config2_300k"]                 # the other BASELINE configurations ride in the same line
    assert "error" not in ex, ex
    for k in ("value", "unit", "ms_per_step", "stages_ms", "roofline", "config"):
        assert k in ex, k
    assert ex["config"]["gaussians"] == 300_000 and ex["config"]["tile_instances"] > 500_000
    assert abs(ex["roofline"]["frac"] - ex["roofline"]["achieved"] / 8000.0) < 1e-12
