#!/usr/bin/env python
"""Tuning aid (no GPU): the Python share of one drop-in fwd+bwd.  The compute entry points of libhgs.so are replaced by
fakes that return at once (as in tests/test_host_logic.py), the tensors live on the CPU, and what remains is the glue:
autograd Function, argument struct, arena plan, torch.empty calls, ctypes marshalling.  `--profile` prints the cProfile
top of the loop.  The allocator is the CPU's, so the absolute figure is a lower bound of the real host floor
(scripts/diag_host_floor.py measures that on a GPU box)."""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
import torch

import diff_gaussian_rasterization as dgr
from hgs import _lib

Cm = dgr._C
real = _lib.lib()


class Fake:
    L_next = 5000

    def __getattr__(self, name):
        return getattr(real, name)

    def hgs_raster_fwd(self, a, geom, binb, img, L_cap, radii, color, invd, Lref, stream, dev):
        C.cast(Lref, C.POINTER(C.c_uint32))[0] = self.L_next
        return 0

    def hgs_raster_fwd_stage1(self, a, geom, radii, Lref, stream, dev):
        C.cast(Lref, C.POINTER(C.c_uint32))[0] = self.L_next
        return 0

    def hgs_raster_fwd_stage2(self, *a):
        return 0

    def hgs_raster_bwd(self, *a):
        return 0


fake = Fake()
_lib.lib = lambda: fake
Cm._require_gpu = lambda t, n: t.contiguous()
Cm._small = lambda t, n, k: t.to(torch.float32).contiguous()
Cm._stream = lambda d: None

ap = argparse.ArgumentParser()
ap.add_argument("--profile", action="store_true")
ap.add_argument("-n", type=int, default=3000)
args = ap.parse_args()

P, W, H = 2000, 256, 256
z = lambda *s: torch.zeros(*s)
rs = dgr.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=1.0, tanfovy=1.0, bg=z(3), scale_modifier=1.0,
                                       viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=3, campos=z(3),
                                       prefiltered=False, debug=False, do_depth=True, render_indices=torch.empty(0, dtype=torch.int32),
                                       parent_indices=torch.empty(0, dtype=torch.int32), interpolation_weights=z(0),
                                       num_node_kids=torch.empty(0, dtype=torch.int32))
rast = dgr.GaussianRasterizer(rs)
prm = dict(means3D=z(P, 3), shs=z(P, 16, 3), opacities=z(P, 1), scales=z(P, 3), rotations=z(P, 4))
for t in prm.values():
    t.requires_grad_(True)
gc, gd = z(3, H, W), z(1, H, W)


def step():
    for t in prm.values():
        t.grad = None
    m2 = torch.zeros(P, 3, requires_grad=True)
    c, r, d = rast(means2D=m2, **prm)
    torch.autograd.backward([c, d], [gc, gd])


def fwd_only():
    with torch.no_grad():
        rast(means2D=None, **prm)


class _Null(torch.autograd.Function):
    """Same tensor inputs and outputs as the rasterizer's Function, no work: what torch's autograd charges for one
    custom Function of this arity (apply, graph node, engine start, six AccumulateGrad nodes)."""

    @staticmethod
    def forward(ctx, m3, m2, sh, op, sc, rot):
        ctx.shapes = [t.shape for t in (m3, m2, sh, op, sc, rot)]
        r = torch.empty(P, dtype=torch.int32)
        ctx.mark_non_differentiable(r)
        ctx.set_materialize_grads(False)
        return torch.empty(3, H, W), r, torch.empty(1, H, W)

    @staticmethod
    def backward(ctx, gc_, gr_, gd_):
        return tuple(torch.empty(s) for s in ctx.shapes)


def null_step():
    for t in prm.values():
        t.grad = None
    m2 = torch.zeros(P, 3, requires_grad=True)
    c, r, d = _Null.apply(prm["means3D"], m2, prm["shs"], prm["opacities"], prm["scales"], prm["rotations"])
    torch.autograd.backward([c, d], [gc, gd])


for f, name in ((step, "fwd+bwd"), (fwd_only, "forward only"), (null_step, "autograd alone (empty Function of the same arity)")):
    for _ in range(200):
        f()
    t0 = time.perf_counter()
    for _ in range(args.n):
        f()
    print(f"{name}: {(time.perf_counter() - t0) / args.n * 1e6:.1f} us of glue per call")

if args.profile:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.n):
        step()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
