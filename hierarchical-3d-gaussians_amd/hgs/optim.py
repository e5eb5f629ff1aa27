"""Fused row-sparse Adam for the Gaussian parameters.

Mirrors the interface of the reference's optimiser, ``scene/OurAdam.py`` (``Adam(params, lr, betas, eps,
weight_decay)``, ``step(relevant)``, per-parameter state ``{'step', 'exp_avg', 'exp_avg_sq'}`` which
``scene/gaussian_model.py:528-620`` edits during densification), but a step is ONE HIP launch over all parameter
tensors (``hgs_adam_step``, ``csrc/adam.hip``) instead of ~12 torch kernels per tensor:

    from hgs.optim import Adam               # instead of: from scene.OurAdam import Adam
    opt.step(relevant)                        # int64 row indices, as train_single.py:171-174
    opt.step(torch.empty(0))                  # dense (OurAdam: relevant.size(0) == 0)
    opt.step_masked(gaussians._opacity.grad)  # same selection without nonzero() and its host sync

Not supported (the reference never uses them): amsgrad, maximize, capturable, foreach.
No CPU fallback: parameters must live on the GPU.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
from torch.optim.optimizer import Optimizer

from . import _lib


class Adam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *,
                 foreach=None, maximize=False, capturable=False):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        if not 0.0 <= weight_decay:
            raise ValueError("Invalid weight_decay value: {}".format(weight_decay))
        if amsgrad or maximize or capturable or foreach:
            raise NotImplementedError("hgs.optim.Adam: amsgrad / maximize / capturable / foreach are not supported")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False)
        super().__init__(params, defaults)

    # -- one fused launch per <= ADAM_MAX_TENSORS tensors with the same row count -------------------------------
    def _collect(self, only=None):
        todo = []
        only = None if only is None else {id(p) for p in only}
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None or (only is not None and id(p) not in only):
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("hgs.optim.Adam needs contiguous float32 GPU parameters (no CPU fallback)")
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                step = state["step"].item()
                bc1 = 1 - beta1 ** step
                bc2 = 1 - beta2 ** step
                rows = p.shape[0] if p.dim() > 0 else 1
                row_len = p.numel() // max(rows, 1)
                t = _lib.AdamTensor(param=p.data_ptr(), grad=grad.data_ptr(), exp_avg=state["exp_avg"].data_ptr(),
                                    exp_avg_sq=state["exp_avg_sq"].data_ptr(), row_len=row_len,
                                    step_size=group["lr"] / bc1, beta1=beta1, one_minus_beta1=1 - beta1, beta2=beta2,
                                    one_minus_beta2=1 - beta2, eps=group["eps"],
                                    weight_decay=group["weight_decay"], bias_correction2_sqrt=math.sqrt(bc2))
                todo.append((rows, p.device, t, grad))
        return todo

    def _launch(self, todo, rows_t, mask_t):
        l = _lib.lib()
        by_rows = {}
        for rows, dev, t, keep in todo:
            by_rows.setdefault((rows, dev), []).append((t, keep))
        for (rows, dev), items in by_rows.items():
            if rows == 0:
                continue
            if mask_t is not None and mask_t.numel() != rows:
                raise RuntimeError(f"row mask has {mask_t.numel()} entries, parameter has {rows} rows")
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for i in range(0, len(items), _lib.ADAM_MAX_TENSORS):
                chunk = items[i:i + _lib.ADAM_MAX_TENSORS]
                arr = (_lib.AdamTensor * len(chunk))(*[t for t, _ in chunk])
                _lib.check(l.hgs_adam_step(arr, len(chunk), rows, _lib.ptr(rows_t),
                                           0 if rows_t is None else rows_t.numel(), _lib.ptr(mask_t), stream,
                                           dev.index if dev.index is not None else torch.cuda.current_device()),
                           "hgs_adam_step")

    @torch.no_grad()
    def step(self, relevant=None, closure=None):
        """relevant: int64 row indices (any shape; flattened) -- empty or None = dense update."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        rows_t = None
        if relevant is not None and relevant.numel() > 0:
            rows_t = relevant.reshape(-1)
            if rows_t.dtype != torch.int64:
                rows_t = rows_t.long()
            if not rows_t.is_cuda:
                raise RuntimeError("relevant must be a GPU tensor")
            rows_t = rows_t.contiguous()
        self._launch(self._collect(), rows_t, None)
        return loss

    @torch.no_grad()
    def step_masked(self, row_grad, params=None):
        """Update row r of every parameter iff row_grad.flatten()[r] != 0 (train_single.py:171 without nonzero()).
        ``params``: restrict the step to these parameters (a step may be issued in several parts, e.g. while the rest of
        a gradient all-reduce is still on the wire; every parameter must be stepped exactly once per optimizer step)."""
        mask = row_grad.reshape(-1)
        if not mask.is_cuda or mask.dtype != torch.float32:
            raise RuntimeError("row_grad must be a float32 GPU tensor")
        self._launch(self._collect(params), None, mask.contiguous())
