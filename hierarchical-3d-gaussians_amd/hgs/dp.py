"""Per-view data parallelism for the rasterizer hot path.

The reference trains one camera per step on one GPU (train_single.py:57-59,
utils/general_utils.py:137) and has no collective anywhere.  Views are independent given
the current parameters, so N ranks each rasterize a different view of the SAME (replicated)
Gaussians and the only exchange is one SUM all-reduce of the flat gradient bucket
(59 floats per Gaussian at SH degree 3 = 236 MB at 1 M Gaussians).  One process per GPU,
``torch.distributed`` -- backend "nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests.
One large bucket, one collective: xGMI is point-to-point (7 links per GPU), per-call latency
dominates small messages, so nothing is split into per-tensor all-reduces.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import torch
import torch.distributed as dist

GRAD_ORDER = ("means3D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp")


def init_from_env(backend: str | None = None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun env)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # HGS_DP_BACKEND=gloo lets several ranks share one GPU (functional test of the N>1 path on a
            # single-GPU box); production is "nccl" = RCCL over xGMI, one rank per GPU.
            backend = os.environ.get("HGS_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class GradBucket:
    """Flat float32 bucket holding the gradients of every Gaussian parameter tensor, laid out
    tensor after tensor; ``views`` alias it, so filling the views fills the bucket."""

    def __init__(self, shapes: Dict[str, Sequence[int]], device):
        self.names: List[str] = [n for n in GRAD_ORDER if n in shapes]
        sizes = [int(torch.Size(shapes[n]).numel()) for n in self.names]
        self.flat = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
        self.views: Dict[str, torch.Tensor] = {}
        off = 0
        for n, sz in zip(self.names, sizes):
            self.views[n] = self.flat[off:off + sz].view(*shapes[n])
            off += sz

    def fill(self, grads: Dict[str, torch.Tensor]):
        for n in self.names:
            self.views[n].copy_(grads[n])

    def all_reduce(self, average: bool = False):
        """SUM over ranks (the gradient of the sum of the per-view losses)."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if average:
                self.flat.div_(dist.get_world_size())
        return self.views

    def span(self, names: Sequence[str]) -> torch.Tensor:
        """The contiguous slice of ``flat`` that holds the given tensors (they must be adjacent in the bucket)."""
        idx = sorted(self.names.index(n) for n in names)
        if idx != list(range(idx[0], idx[0] + len(idx))):
            raise ValueError(f"{tuple(names)} are not adjacent in the bucket ({self.names})")
        first, last = self.views[self.names[idx[0]]], self.views[self.names[idx[-1]]]
        a = first.storage_offset()
        return self.flat[a:last.storage_offset() + last.numel()]

    def all_reduce_async(self, names: Sequence[str]):
        """Start the SUM all-reduce of one contiguous group of tensors; returns the work handle (or None on one rank).
        The collective is ordered after everything enqueued so far on the CURRENT stream and runs next to whatever
        the caller enqueues afterwards; ``handle.wait()`` orders the current stream after it."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            return dist.all_reduce(self.span(names), op=dist.ReduceOp.SUM, async_op=True)
        return None


def shard_views(num_views: int, rank: int, world: int) -> List[int]:
    """View indices rendered by ``rank``: view i goes to rank i % world (train_single.py's one-camera
    loop, unrolled across ranks)."""
    return list(range(rank, num_views, world))


class DensifyStats:
    """The three densification statistics the reference updates after every view (train_single.py:146-148,
    scene/gaussian_model.py:686-689), collected over the views a rank renders in one step and reduced over the ranks:

        xyz_gradient_accum[visible] = max(|means2D.grad[visible, :2]|, xyz_gradient_accum[visible])   -> MAX over views
        denom[visible] += 1                                                                          -> SUM over views
        max_radii2D[visible] = max(max_radii2D[visible], radii[visible])                             -> MAX over views

    Rendering N views in one data-parallel step and reducing like this leaves exactly what the reference's loop leaves
    after stepping through the same N views one by one (max and + are associative and commutative)."""

    def __init__(self, P: int, device):
        self.maxes = torch.zeros(2, P, dtype=torch.float32, device=device)   # row 0: |grad| max, row 1: radii max
        self.count = torch.zeros(P, dtype=torch.float32, device=device)      # views that saw the Gaussian

    def reset(self):
        self.maxes.zero_()
        self.count.zero_()

    @torch.no_grad()
    def add_view(self, means2D_grad: torch.Tensor, radii: torch.Tensor):
        vis = radii > 0
        norm = torch.linalg.vector_norm(means2D_grad[:, :2], dim=-1)
        torch.maximum(self.maxes[0], torch.where(vis, norm, torch.zeros_like(norm)), out=self.maxes[0])
        torch.maximum(self.maxes[1], radii.to(torch.float32), out=self.maxes[1])      # radii < 2^24: exact in float32
        self.count += vis

    def all_reduce(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.maxes, op=dist.ReduceOp.MAX)
            dist.all_reduce(self.count, op=dist.ReduceOp.SUM)

    @torch.no_grad()
    def apply(self, xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor):
        """Fold the (reduced) statistics of this step into the model's accumulators ([P,1], [P,1], [P])."""
        seen = self.count > 0
        acc = xyz_gradient_accum.view(-1)
        acc.copy_(torch.where(seen, torch.maximum(acc, self.maxes[0]), acc))
        denom.view(-1).add_(self.count)
        max_radii2D.copy_(torch.maximum(max_radii2D, self.maxes[1].to(max_radii2D.dtype)))


class DataParallelStep:
    """One optimizer step of per-view data parallelism over the rasterizer hot path:

        for every view of this rank:   forward, loss, backward -- gradients ACCUMULATE in the flat bucket
        SUM all-reduce of the bucket, MAX / SUM all-reduce of the densification statistics
        relevant rows = non-zero of the REDUCED opacity gradient (train_single.py:170-174)
        the same Adam step on every rank (hgs.optim.Adam.step_masked: no nonzero(), no host sync)

    so that N ranks holding identical parameters before the step hold identical parameters after it.  The exchange is
    split in two so that it overlaps with compute: the (opacity, scale, rotation) gradients are final as soon as the
    last view's per-Gaussian backward has run and go on the wire while the batched SH backward still runs; the
    (position, SH) gradients follow, and while THEY are on the wire the optimizer already updates opacity / scale /
    rotation.  Works with one rank too (no collective is issued).

    params:     {"means3D", "shs", "opacities", "scales", "rotations"} -> leaf tensors (the op's direct inputs)
    optimizer:  an hgs.optim.Adam over those tensors (or anything with ``step_masked(row_grad, params=None)``)
    """

    EARLY = ("opacities", "scales", "rotations")
    LATE = ("means3D", "shs")

    def __init__(self, params: Dict[str, torch.Tensor], optimizer, backward_stream=None, make_context=None):
        self.params = params
        self.optimizer = optimizer
        dev = params["means3D"].device
        P = params["means3D"].shape[0]
        self.bucket = GradBucket({k: tuple(v.shape) for k, v in params.items()}, dev)
        for k, v in params.items():
            v.grad = self.bucket.views[k]          # the optimizer reads the reduced bucket in place
        self.means2D_grad = torch.zeros(P, 3, dtype=torch.float32, device=dev)
        self.stats = DensifyStats(P, dev)
        self.backward_stream = backward_stream
        if make_context is None:
            import diff_gaussian_rasterization as dgr
            make_context = dgr.RasterContext
        self.context = make_context(grad_buffers=dict(self.bucket.views, means2D=self.means2D_grad),
                                    backward_stream=backward_stream)
        self._views = 0

    def begin(self):
        self._views = 0
        self.context.grad_accumulate = False
        self.stats.reset()

    def _on_backward_stream(self):
        import contextlib
        sb = self.backward_stream
        return torch.cuda.stream(sb) if sb is not None else contextlib.nullcontext()

    def view_done(self, radii: torch.Tensor):
        """Call after the view's backward has been issued: books its densification statistics (on the stream its
        means2D gradient was produced on) and switches the following views to accumulation."""
        with self._on_backward_stream():
            self.stats.add_view(self.means2D_grad, radii)
        self._views += 1
        self.context.grad_accumulate = True

    def finish(self, sh_backward=None):
        """End of the step.  ``sh_backward``: optional callable that issues the batched SH backward of the step's
        views (it completes the means3D / shs gradients); the first all-reduce overlaps with it."""
        with self._on_backward_stream():
            early = self.bucket.all_reduce_async(self.EARLY)     # ordered after the last view's backward
            if sh_backward is not None:
                sh_backward()
            late = self.bucket.all_reduce_async(self.LATE)
            self.stats.all_reduce()
            if early is not None:
                early.wait()
            mask = self.bucket.views["opacities"]
            self.optimizer.step_masked(mask, params=[self.params[k] for k in self.EARLY])
            if late is not None:
                late.wait()
            self.optimizer.step_masked(mask, params=[self.params[k] for k in self.LATE])
        self.context.wait_backward_stream()
        return self._views
