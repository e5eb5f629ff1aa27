"""Size-independent properties of the op, checked on the GPU at sizes no dense oracle
reaches: permutation of the input order, Gaussians that cannot contribute, rigid motion of scene + camera, linearity
of the backward in the upstream gradient."""
import numpy as np
import pytest
import torch

import parity as pa
from hgs import synth

pytestmark = pytest.mark.gpu


def _case(P, W, H, seed):
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(P, cam, seed=seed)
    gc, gd = synth.upstream_grads(H, W, seed=seed + 1)
    return cam, scene, gc, gd


def test_permuting_the_gaussians_permutes_the_result(gpu):
    """The image does not depend on the order the Gaussians are passed in, and every gradient row follows its
    Gaussian -- BIT FOR BIT: depths are distinct, so the per-tile blend order is the same, each Gaussian's instance
    partials are summed in the same (tile) order, and nothing in the backward is an atomic."""
    P, W, H = 200_000, 1280, 720
    cam, scene, gc, gd = _case(P, W, H, 3)
    # distinct depths (random float32 depths collide: ~0.7 % of 200 k draws from [2, 20])
    z = torch.linspace(2.0, 20.0, P)[torch.from_numpy(np.random.default_rng(1).permutation(P))]
    scene.means3D[:, :2] *= (z / scene.means3D[:, 2])[:, None]        # keep the screen position
    scene.means3D[:, 2] = z
    assert np.unique(scene.means3D[:, 2].numpy()).size == P, "the case needs distinct depths"
    bg = torch.tensor([0.1, 0.0, 0.2])
    perm = torch.from_numpy(np.random.default_rng(0).permutation(P))
    shuffled = synth.Scene(scene.means3D[perm], scene.scales[perm], scene.rotations[perm], scene.opacities[perm],
                           scene.shs[perm], scene.sh_degree)
    a = pa.run_hip(scene, cam, bg, gc, gd, gpu, debug=False)
    b = pa.run_hip(shuffled, cam, bg, gc, gd, gpu, debug=False)
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["invdepth"], b["invdepth"])
    assert torch.equal(a["radii"][perm], b["radii"])
    for k in a["grads"]:
        assert torch.equal(a["grads"][k][perm], b["grads"][k]), k


def test_gaussians_that_cannot_contribute_change_nothing(gpu):
    """Appending Gaussians with opacity below 1/255 (never blended), behind the near plane, or far outside the frustum
    leaves the image and every other gradient row untouched -- bit for bit -- and gives them zero gradients."""
    P, W, H = 100_000, 960, 544
    cam, scene, gc, gd = _case(P, W, H, 5)
    g = torch.Generator().manual_seed(9)
    n = 30_000
    extra = synth.make_scene(n, cam, seed=11)
    kind = torch.randint(0, 3, (n,), generator=g)
    extra.opacities[kind == 0] = 0.003                               # alpha can never reach 1/255
    extra.means3D[kind == 1, 2] = -extra.means3D[kind == 1, 2]        # behind the camera
    extra.means3D[kind == 2, 0] += 500.0                              # far to the side
    both = synth.Scene(*(torch.cat((getattr(scene, f), getattr(extra, f))) for f in
                         ("means3D", "scales", "rotations", "opacities", "shs")), scene.sh_degree)
    bg = torch.zeros(3)
    a = pa.run_hip(scene, cam, bg, gc, gd, gpu, debug=False)
    b = pa.run_hip(both, cam, bg, gc, gd, gpu, debug=False)
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["invdepth"], b["invdepth"])
    for k in a["grads"]:
        assert torch.equal(a["grads"][k], b["grads"][k][:P]), k
        assert float(b["grads"][k][P:].abs().max()) == 0.0, k
    assert int((b["radii"][P:][kind != 0] != 0).sum()) == 0           # culled: radius 0 (faint ones are still "visible")


def test_rigid_motion_of_scene_and_camera(gpu):
    """Rotating and translating the Gaussians together with the camera leaves the image unchanged (to float32
    accuracy of the transformed inputs): the op has no preferred world frame."""
    P, W, H = 50_000, 640, 352
    cam, scene, gc, gd = _case(P, W, H, 7)
    ang = 0.7
    R = torch.tensor([[np.cos(ang), 0.0, np.sin(ang)], [0.0, 1.0, 0.0], [-np.sin(ang), 0.0, np.cos(ang)]], dtype=torch.float64)
    t = torch.tensor([3.0, -1.5, 2.0], dtype=torch.float64)
    # world' = R world + t; camera-to-world rotation R, centre t  ->  world-to-camera translation -R^T t
    cam2 = synth.make_camera(W, H, R=R.numpy(), T=(-R.T @ t).numpy())
    half = ang / 2
    qr = torch.tensor([np.cos(half), 0.0, np.sin(half), 0.0], dtype=torch.float64)      # rotation about y
    q = scene.rotations.double()
    w1, x1, y1, z1 = qr
    w2, x2, y2, z2 = q.unbind(1)
    q2 = torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                      w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], 1)
    moved = synth.Scene((scene.means3D.double() @ R.T + t).float().contiguous(), scene.scales, q2.float().contiguous(),
                        scene.opacities, scene.shs.clone(), 0)
    base = synth.Scene(scene.means3D, scene.scales, scene.rotations, scene.opacities, scene.shs, 0)   # SH degree 0:
    bg = torch.tensor([0.2, 0.3, 0.1])                                                               # no view dependence
    a = pa.run_hip(base, cam, bg, gc, gd, gpu, debug=False)
    b = pa.run_hip(moved, cam2, bg, gc, gd, gpu, debug=False)
    # positions are rounded to float32 after the motion: a handful of pixels may see a blend decision flip
    diff = (a["color"] - b["color"]).abs()
    assert float(diff.median()) < 1e-6 and float((diff > 1e-4).float().mean()) < 1e-3
    assert float((a["radii"] != b["radii"]).float().mean()) < 1e-3


def test_backward_is_linear_in_the_upstream_gradient(gpu):
    """dL/dinputs is linear in (dL/dcolor, dL/dinvdepth): backward(a g1 + b g2) == a backward(g1) + b backward(g2)."""
    P, W, H = 100_000, 960, 544
    cam, scene, gc, gd = _case(P, W, H, 13)
    gc2, gd2 = synth.upstream_grads(H, W, seed=99)
    bg = torch.tensor([0.3, 0.2, 0.1])
    g1 = pa.run_hip(scene, cam, bg, gc, gd, gpu, debug=False)["grads"]
    g2 = pa.run_hip(scene, cam, bg, gc2, gd2, gpu, debug=False)["grads"]
    g12 = pa.run_hip(scene, cam, bg, 0.5 * gc - 2.0 * gc2, 0.5 * gd - 2.0 * gd2, gpu, debug=False)["grads"]
    for k in g1:
        want = 0.5 * g1[k].double() - 2.0 * g2[k].double()
        scale = float(want.abs().max())
        assert float((g12[k].double() - want).abs().max()) <= 2e-5 * scale, k
