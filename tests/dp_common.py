"""Shared pieces of the data-parallel step tests (CPU / gloo and GPU): the toy scene, the views of a step, a CPU
stand-in for hgs.optim.Adam (the fused kernel has no CPU path) built on the pinned Adam oracle."""
import numpy as np
import torch

W, H, P = 96, 64, 400
NAMES = ("means3D", "shs", "opacities", "scales", "rotations")
LRS = dict(means3D=1.6e-4, shs=2.5e-3, opacities=5e-3, scales=5e-4, rotations=1e-3)


def scene_and_cams(n_views):
    from hgs import synth
    base = synth.make_camera(W, H)
    scene = synth.make_scene(P, base, seed=5)
    # half of the Gaussians are moved behind the cameras: they get zero gradients from every view, so the row
    # selection by the REDUCED opacity gradient (train_single.py:170-174) has something to select
    scene.means3D[P // 2:, 2] = -scene.means3D[P // 2:, 2]
    cams = [synth.orbit_camera(W, H, j, n_views, radius=0.4) for j in range(n_views)]
    g = torch.Generator().manual_seed(9)
    targets = [(torch.rand(3, H, W, generator=g), 0.3 * torch.rand(1, H, W, generator=g)) for _ in range(n_views)]
    return scene, cams, targets


class OracleAdam:
    """``step_masked(row_grad, params=None)`` of hgs.optim.Adam on CPU tensors, through oracle/adam_oracle.py
    (float64 arithmetic, float32 storage).  TEST ONLY."""

    def __init__(self, groups, eps=1e-15):
        self.groups = groups                 # [{"params": [tensor], "lr": float}]
        self.eps = eps
        self.state = {}

    def step_masked(self, row_grad, params=None):
        from oracle import adam_oracle
        only = None if params is None else {id(p) for p in params}
        rel = np.nonzero(row_grad.reshape(-1).numpy() != 0)[0]
        for g in self.groups:
            for p in g["params"]:
                if only is not None and id(p) not in only:
                    continue
                st = self.state.setdefault(id(p), dict(step=0, m=np.zeros(p.shape), v=np.zeros(p.shape)))
                st["step"] += 1
                if rel.size == 0:
                    continue                     # nothing visible anywhere: OurAdam's "No grads!" dense step of zeros
                new, st["m"], st["v"] = adam_oracle.adam_rows(p.detach().numpy(), p.grad.numpy(), st["m"], st["v"],
                                                              st["step"], rel, lr=g["lr"], eps=self.eps)
                with torch.no_grad():
                    p.copy_(torch.from_numpy(new).float())
