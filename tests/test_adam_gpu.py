"""hgs.optim.Adam (csrc/adam.hip through the C ABI) against the reference optimiser's golden outputs and against
the float64 oracle at a realistic size.  Tolerance 2e-6 of the tensor max (float32 arithmetic, op order free)."""
import json
import os
import time

import numpy as np
import pytest
import torch

import adam_cases as ac
from oracle import adam_oracle as ao

pytestmark = pytest.mark.gpu
TOL = 2e-6
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _make_opt(params, wd=0.0):
    from hgs.optim import Adam
    return Adam([dict(params=[params[k]], lr=ac.LRS[k], name=k) for k in ac.KEYS], lr=0.0, eps=ac.EPS,
                weight_decay=wd)


@pytest.mark.parametrize("name", ["sparse", "dense", "mixed", "decay"])
@pytest.mark.parametrize("masked", [False, True])
def test_matches_reference_optimizer_golden(gpu, name, masked):
    z = ac.load()
    P, steps, wd, _ = ac.meta(z, name)
    params = {k: torch.nn.Parameter(torch.from_numpy(z[f"{name}.{k}.init"]).to(gpu)) for k in ac.KEYS}
    opt = _make_opt(params, wd)
    for it in range(steps):
        for k in ac.KEYS:
            params[k].grad = torch.from_numpy(z[f"{name}.{k}.grad{it}"]).to(gpu)
        rel = torch.from_numpy(z[f"{name}.relevant{it}"]).to(gpu)
        if masked and rel.numel() > 0:
            opt.step_masked(params["opacity"].grad)
        else:
            opt.step(rel)
    torch.cuda.synchronize()
    for k in ac.KEYS:
        st = opt.state[params[k]]
        assert float(st["step"]) == steps
        assert ac.rel_err(params[k].detach().cpu().numpy(), z[f"{name}.{k}.final"]) <= TOL, k
        assert ac.rel_err(st["exp_avg"].cpu().numpy(), z[f"{name}.{k}.exp_avg"]) <= TOL, k
        assert ac.rel_err(st["exp_avg_sq"].cpu().numpy(), z[f"{name}.{k}.exp_avg_sq"]) <= TOL, k


def test_million_rows_against_oracle_and_throughput(gpu):
    P = 1_000_000
    g = torch.Generator().manual_seed(21)
    shapes = dict(xyz=(3,), f_dc=(1, 3), f_rest=(15, 3), opacity=(1,), scaling=(3,), rotation=(4,))
    init = {k: torch.randn(P, *s, generator=g) for k, s in shapes.items()}
    grads = {k: torch.randn(P, *s, generator=g) * 1e-3 for k, s in shapes.items()}
    keep = torch.rand(P, generator=g) < 0.7
    grads["opacity"][~keep] = 0
    rel = (grads["opacity"].flatten() != 0).nonzero().flatten()
    params = {k: torch.nn.Parameter(v.clone().to(gpu)) for k, v in init.items()}
    for k in ac.KEYS:
        params[k].grad = grads[k].to(gpu)
    opt = _make_opt(params)
    rel_d = rel.to(gpu)
    opt.step(rel_d)
    opt.step_masked(params["opacity"].grad)
    torch.cuda.synchronize()
    for k in ("xyz", "opacity", "f_rest"):
        p = init[k].numpy().astype(np.float64)
        m, v = np.zeros_like(p), np.zeros_like(p)
        for it in range(2):
            p, m, v = ao.adam_rows(p, grads[k].numpy(), m, v, it + 1, rel.numpy(), lr=ac.LRS[k], eps=ac.EPS)
        assert ac.rel_err(params[k].detach().cpu().numpy(), p) <= TOL, k
        assert ac.rel_err(opt.state[params[k]]["exp_avg_sq"].cpu().numpy(), v) <= TOL, k
    # untouched rows are bit-identical to their initial values
    untouched = (~keep).nonzero().flatten()[:1000]
    assert torch.equal(params["f_rest"].detach().cpu()[untouched], init["f_rest"][untouched])

    # throughput of one fused step (rows listed / masked / dense) vs the bytes it has to move
    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n
    res = {}
    n_sel = int(rel.numel())
    for name, fn, rows in (("rows", lambda: opt.step(rel_d), n_sel),
                           ("masked", lambda: opt.step_masked(params["opacity"].grad), n_sel),
                           ("dense", lambda: opt.step(None), P)):
        dt = timed(fn)
        res[name] = dict(ms=dt * 1e3, rows=rows, GBps=rows * 59 * 28 / dt / 1e9)
    print("adam step:", json.dumps(res))
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "parity_log.jsonl"), "a") as f:
            f.write(json.dumps({"case": "adam_step_1M", **res}) + "\n")
    except OSError:
        pass


def test_rejects_cpu_parameters_and_unsupported_modes(gpu):
    from hgs.optim import Adam
    with pytest.raises(NotImplementedError):
        Adam([torch.nn.Parameter(torch.zeros(4, 3, device=gpu))], amsgrad=True)
    p = torch.nn.Parameter(torch.zeros(4, 3))
    p.grad = torch.ones(4, 3)
    with pytest.raises(RuntimeError):
        Adam([p], lr=1e-3).step(None)
