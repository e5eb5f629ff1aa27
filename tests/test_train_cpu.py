"""The miniature optimisation loop itself (tests/train_loop.py), exercised on CPU with the oracle as renderer:
it must reduce the loss, and the float32-geometry and float64 oracles must stay within the PSNR tolerance used by
the GPU parity test."""
import torch

import train_loop as tl


def test_training_loop_converges_and_is_precision_stable():
    cams, scene = tl.make_problem(P=300, size=64, n_views=2, seed=0)
    bg = torch.zeros(3)
    f64 = tl.oracle_render_fn(bg, 3, torch.float64)
    f32 = tl.oracle_render_fn(bg, 3, torch.float32)
    with torch.no_grad():
        gt = {k: v.detach() for k, v in tl.activate(tl.raw_params_from_scene(scene, "cpu")).items()}
        targets = [f64(c, gt) for c in cams]
    out = {}
    for name, fn in (("f64", f64), ("f32", f32)):
        raw = tl.raw_params_from_scene(scene, "cpu", jitter_seed=5)
        p0 = tl.evaluate(fn, raw, cams, targets)
        losses = tl.optimise(fn, raw, cams, targets, 12)
        out[name] = (p0, tl.evaluate(fn, raw, cams, targets), losses)
    assert out["f64"][2][-1] < 0.6 * out["f64"][2][0]
    assert out["f64"][1] > out["f64"][0] + 3.0
    assert abs(out["f64"][1] - out["f32"][1]) <= 0.01
