"""PSNR parity (BASELINE.json metric, second half: 'PSNR delta vs reference'; north_star: within 0.01 dB).

The same miniature optimisation (tests/train_loop.py, modelled on train_single.py) is run twice from the same
perturbed start against the same targets: once with the HIP op on the GPU, once with the float64 CPU oracle.
The two final PSNRs must agree within 0.01 dB and both must have improved substantially.
"""
import json
import os

import pytest
import torch

import train_loop as tl

PSNR_TOL_DB = 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["toy_l1", "multi_tile_l1_dssim"])
def test_short_optimisation_psnr_matches_oracle(gpu, case):
    """toy_l1: 1 000 Gaussians, 128x128, 40 steps, L1 + inverse depth.  multi_tile_l1_dssim: 8 000 Gaussians, 320x192
    (240 tiles, lists of ~100 instances), 6 views, 30 steps, the reference's colour loss 0.8 L1 + 0.2 (1 - SSIM)
    (train_single.py:101-108) + inverse depth."""
    if case == "toy_l1":
        cams, scene = tl.make_problem(P=1000, size=128, n_views=4, seed=0)
        steps, dssim = 40, 0.0
    else:
        cams, scene = tl.make_problem(P=8000, size=320, height=192, n_views=6, seed=1)
        steps, dssim = 30, 0.2
    bg = torch.zeros(3)
    oracle = tl.oracle_render_fn(bg, 3, torch.float64)
    hip = tl.hip_render_fn(bg, 3, gpu)
    with torch.no_grad():
        gt = {k: v.detach() for k, v in tl.activate(tl.raw_params_from_scene(scene, "cpu")).items()}
        targets = [oracle(c, gt) for c in cams]
    raw_o = tl.raw_params_from_scene(scene, "cpu", jitter_seed=5)
    raw_h = tl.raw_params_from_scene(scene, gpu, jitter_seed=5)
    p0 = tl.evaluate(oracle, raw_o, cams, targets)
    loss_o = tl.optimise(oracle, raw_o, cams, targets, steps, lambda_dssim=dssim)
    loss_h = tl.optimise(hip, raw_h, cams, targets, steps, lambda_dssim=dssim)
    p_o = tl.evaluate(oracle, raw_o, cams, targets)
    p_h = tl.evaluate(hip, raw_h, cams, targets)
    # the HIP-trained parameters rendered by the oracle: renderer-independent PSNR
    raw_hc = {k: v.detach().cpu() for k, v in raw_h.items()}
    p_h_via_oracle = tl.evaluate(oracle, raw_hc, cams, targets)
    print(f"PSNR start {p0:.4f} dB; oracle-trained {p_o:.4f}; hip-trained {p_h:.4f} "
          f"(rendered by the oracle: {p_h_via_oracle:.4f}); loss {loss_o[0]:.5f}->{loss_o[-1]:.5f} / "
          f"{loss_h[0]:.5f}->{loss_h[-1]:.5f}")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_log.jsonl"), "a") as f:
            f.write(json.dumps({"case": "psnr_short_optimisation_" + case, "steps": steps, "psnr_start_db": p0,
                                "psnr_oracle_db": p_o, "psnr_hip_db": p_h, "psnr_hip_via_oracle_db": p_h_via_oracle,
                                "delta_db": p_h - p_o}) + "\n")
    except OSError:
        pass
    assert p_o > p0 + 5.0 and p_h > p0 + 5.0
    assert abs(loss_o[0] - loss_h[0]) <= 1e-5 * abs(loss_o[0])
    assert abs(p_h - p_o) <= PSNR_TOL_DB
    assert abs(p_h - p_h_via_oracle) <= PSNR_TOL_DB / 10
