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


def shard_views(num_views: int, rank: int, world: int) -> List[int]:
    """View indices rendered by ``rank``: view i goes to rank i % world (train_single.py's one-camera
    loop, unrolled across ranks)."""
    return list(range(rank, num_views, world))
