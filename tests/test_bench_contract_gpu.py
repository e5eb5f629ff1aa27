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
    # the PMC counters behind `traffic` are collected by rocprofv3 child processes of the run itself; where the
    # profiler cannot run, the line must say why (and falls back to a committed summary of the same build, or null)
    assert isinstance(rf["traffic_source"], str) and rf["traffic_source"], rf
    if rf["traffic_source"].startswith("measured in this run"):
        assert rf["traffic"] > 0 and rf["traffic_upper"] >= rf["traffic"]
        assert {"render_bwd", "render_fwd", "preprocess_fwd", "preprocess_bwd"} <= set(rf["traffic_by_stage"])
        assert rf["valu"]["insts_per_launch"] > 0 and "measured in this run" in rf["valu"]["source"]
    else:
        import warnings
        warnings.warn("bench.py could not collect PMC counters in this run: " + rf["traffic_source"][:300])
    assert d["batched"]["value"] > 0 and d["dropin"]["value"] == d["value"]
    assert d["config"]["capacity_misses"] == 0 and len(d["kernel_source_sha"]) == 16
    assert 0.0 < d["host_ms_per_step"] <= d["ms_per_step"] * 1.05 and 0.0 < d["host_floor"]["ms_per_step"] < 5.0
    ex = d["extra"]["config2_300k"]                 # the other BASELINE configurations ride in the same line
    assert "error" not in ex, ex
    for k in ("value", "unit", "ms_per_step", "stages_ms", "roofline", "config"):
        assert k in ex, k
    assert ex["config"]["gaussians"] == 300_000 and ex["config"]["tile_instances"] > 500_000
    assert abs(ex["roofline"]["frac"] - ex["roofline"]["achieved"] / 8000.0) < 1e-12


@pytest.mark.gpu
def test_bench_two_ranks_as_the_driver_launches_them(gpu):
    """The N > 1 path of bench.py, launched exactly as the driver launches it (torch.distributed.run, one process per
    rank) -- here with the two ranks SHARING the one GPU and talking over gloo (HGS_DP_BACKEND), because no multi-GPU node
    is available to the tests: rank 0 prints the one line, value is the whole-job aggregate of the batched schedule, and
    the line carries the per-rank exchange times the first SCALE record is supposed to explain itself with."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HGS_DP_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--gaussians", "20000", "--width", "320", "--height", "192", "--views-per-step", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["schedule"] == "batched"
    assert d["steps"] == 3 and d["value"] > 0
    assert abs(d["value"] - 2 * 2 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]     # ranks x views per step / step time
    ex = d["exchange"]
    assert ex["backend"] == "gloo" and len(ex["exchange_ms_per_rank"]) == 2 and all(t > 0 for t in ex["exchange_ms_per_rank"])
    assert ex["bytes_per_step"] == 20000 * 59 * 4 and ex["busbw_GBps"] > 0
    # north_star's own partition beside it: one view per rank per step, the bucket exchanged every step
    pv = d["per_view_dp"]
    assert pv["global_batch"] == 2 and pv["views_per_step_per_gpu"] == 1 and pv["value"] > 0
    assert abs(pv["value"] - 2 * 1e3 / pv["ms_per_step"]) <= 1e-6 * pv["value"] and 0 < pv["exchange_share_of_step"]


@pytest.mark.gpu
def test_bench_gpus_2_launched_plainly_means_two_ranks(gpu):
    """`python bench.py --gpus 2` with no torchrun environment launches itself under torch.distributed.run: the line says
    n_gpus 2.  (Two ranks on the one GPU over gloo, as above; without HGS_DP_BACKEND=gloo a node with fewer GPUs than
    ranks is refused with a non-zero exit and no line.)"""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    small = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--gaussians", "20000", "--width", "320", "--height", "192",
             "--views-per-step", "2"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + small, capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=dict(env, HGS_DP_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["exchange"]["backend"] == "gloo"
    if torch.cuda.device_count() < 2:
        env.pop("HGS_DP_BACKEND", None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + small, capture_output=True, text=True,
                           timeout=300, cwd=ROOT, env=env)
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
        assert "GPU" in r.stderr
