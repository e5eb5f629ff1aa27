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
import sys
from typing import Dict, List, Optional, Sequence

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

    def __init__(self, shapes: Dict[str, Sequence[int]], device, direct: Optional[bool] = None):
        """``direct``: exchange through DirectAllReduce (peer pointers) instead of torch.distributed's all-reduce;
        default from the environment (HGS_DP_ALLREDUCE=direct).  The bucket then lives in exportable memory and every
        tensor starts on a multiple of 4 floats (the direct kernels move 16 bytes per lane)."""
        self.names: List[str] = [n for n in GRAD_ORDER if n in shapes]
        sizes = [int(torch.Size(shapes[n]).numel()) for n in self.names]
        if direct is None:
            direct = os.environ.get("HGS_DP_ALLREDUCE", "") == "direct"
        direct = bool(direct) and dist.is_initialized() and dist.get_world_size() > 1
        pad = (lambda x: (x + 3) // 4 * 4) if direct else (lambda x: x)
        total = sum(pad(sz) for sz in sizes)
        self.direct = None
        if direct:
            try:
                self.direct = DirectAllReduce(total, device)
            except DirectRouteUnavailable as e:
                # every rank takes this branch together (DirectAllReduce agrees on the outcome collectively): the
                # exchange falls back to torch.distributed's all-reduce (RCCL) instead of taking the job down
                if dist.get_rank() == 0:
                    print(f"[hgs.dp] direct peer-pointer all-reduce unavailable, using torch.distributed "
                          f"({dist.get_backend()}) instead: {e}", file=sys.stderr, flush=True)
        self.flat = self.direct.flat if self.direct is not None else torch.zeros(total, dtype=torch.float32, device=device)
        self._comm = None
        self._calls = 0
        self._verify_left = int(os.environ.get("HGS_P2P_VERIFY", "0") or 0) if self.direct is not None else 0
        self._check_every = max(1, int(os.environ.get("HGS_P2P_CHECK_EVERY", "16") or 16))
        self.views: Dict[str, torch.Tensor] = {}
        off = 0
        for n, sz in zip(self.names, sizes):
            self.views[n] = self.flat[off:off + sz].view(*shapes[n])
            off += pad(sz)

    def fill(self, grads: Dict[str, torch.Tensor]):
        for n in self.names:
            self.views[n].copy_(grads[n])

    def all_reduce(self, average: bool = False):
        """SUM over ranks (the gradient of the sum of the per-view losses)."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            if self.direct is not None:
                ref = self._verify_begin(self.flat)
                self.direct.all_reduce()
                self._verify_end(ref, self.flat)
                self.check_direct()
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if average:
                self.flat.div_(dist.get_world_size())
        return self.views

    # ---- safety net of the direct route ---------------------------------------------------------------------------
    def check_direct(self, force: bool = False):
        """A barrier timeout of the direct route is fatal (csrc/p2p.hip poisons the bucket with NaN on the device);
        this turns it into an exception on the host.  It costs a host sync, so it runs every HGS_P2P_CHECK_EVERY-th
        exchange (default 16) unless ``force``."""
        if self.direct is None:
            return
        self._calls += 1
        if force or self._calls % self._check_every == 0:
            self.direct.check()

    def _verify_begin(self, span):
        """HGS_P2P_VERIFY=N: the first N exchanges of the direct route are recomputed by torch.distributed's
        all-reduce from a copy of the inputs and compared -- the self-check for the first run on real xGMI."""
        if self._verify_left <= 0:
            return None
        ref = span.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        return ref

    def _verify_end(self, ref, span):
        if ref is None:
            return
        self._verify_left -= 1
        torch.cuda.synchronize(span.device)
        self.direct.check()
        scale = float(ref.abs().max())
        err = float((span - ref).abs().max()) if bool(torch.isfinite(span).all()) else float("inf")
        if not err <= 1e-5 * max(scale, 1e-30):      # summation orders differ (rank order here, ring order in RCCL)
            raise RuntimeError(f"direct all-reduce disagrees with torch.distributed: max |diff| {err:.3e} at scale "
                               f"{scale:.3e} (HGS_P2P_VERIFY); do not use HGS_DP_ALLREDUCE=direct on this system")

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
            if self.direct is not None:
                # the same contract on a communication stream of our own: ordered after the current stream's work,
                # wait() orders the current stream after the exchange
                sp = self.span(names)
                dev = self.flat.device
                if self._comm is None:
                    self._comm = torch.cuda.Stream(device=dev)
                ref = self._verify_begin(sp)
                self._comm.wait_stream(torch.cuda.current_stream(dev))
                self.direct.all_reduce(sp.storage_offset() - self.flat.storage_offset(), sp.numel(), stream=self._comm)
                self._verify_end(ref, sp)
                ev = torch.cuda.Event()
                ev.record(self._comm)
                return _EventHandle(ev, dev)
            return dist.all_reduce(self.span(names), op=dist.ReduceOp.SUM, async_op=True)
        return None


class _EventHandle:
    """What GradBucket.all_reduce_async returns on the direct route: wait() like a torch.distributed work handle."""

    def __init__(self, event, device):
        self.event, self.device = event, device

    def wait(self):
        torch.cuda.current_stream(self.device).wait_event(self.event)
        return True


class DirectRouteUnavailable(RuntimeError):
    """The peer-pointer route cannot be set up on this system (hipIpc refused, a peer could not be opened, ...).
    Raised by EVERY rank of the job together, so that all of them can fall back to torch.distributed."""


class _DeviceArray:
    """Raw device memory handed to torch through __cuda_array_interface__ (torch.as_tensor shares it, does not copy)."""

    def __init__(self, ptr: int, numel: int):
        self.__cuda_array_interface__ = {"shape": (numel,), "typestr": "<f4", "data": (ptr, False), "version": 2,
                                         "strides": None}


class DirectAllReduce:
    """SUM all-reduce of a flat float32 bucket over PEER POINTERS (include/hgs.h, hgs_p2p_*): every rank reduces one
    shard of all buckets by reading its peers' memory directly (xGMI is point to point: all seven links of a GPU carry
    data at once, where a ring all-reduce is bound by one link -- SURVEY.md section 5) and then copies the other
    shards.  Every rank ends with bit-identical sums (each shard is summed by one rank, in rank order).

    The bucket lives in memory this class allocates (``flat``: a torch tensor over it), because it has to be exported
    to the other processes; hand it to ``GradBucket(..., flat=...)``.  Control plane: torch.distributed (any backend)
    for the one-off exchange of the IPC handles.  Opt-in (``HGS_DP_ALLREDUCE=direct``); unmeasured on multi-GPU
    hardware -- the protocol is exercised by two ranks sharing one GPU (tests/test_dp_direct_gpu.py)."""

    def __init__(self, numel: int, device: torch.device):
        import ctypes as C
        from . import _lib
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            raise RuntimeError("DirectAllReduce needs an initialised process group with more than one rank")
        self._lib, self._C = _lib, C
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if self.world > _lib.P2P_MAX_WORLD:
            raise RuntimeError(f"DirectAllReduce supports up to {_lib.P2P_MAX_WORLD} ranks")
        self.device = torch.device(device)
        self.dev_index = self.device.index or 0
        self.numel = int(numel)
        padded = (self.numel + 3) // 4 * 4
        lib = _lib.lib()
        own_buf, own_flag = C.c_void_p(), C.c_void_p()
        fine = 2 if os.environ.get("HGS_P2P_FINEGRAINED", "") == "1" else 0
        self._own, self._opened = [], []          # every pointer is recorded right after its own allocation
        # Every step that can fail for ONE rank only (allocation, hipIpc export, opening a peer) is followed by an
        # exchange of the outcome, so that all ranks raise DirectRouteUnavailable together instead of one raising
        # while the others wait in a collective.
        mine, err = None, None
        try:
            if os.environ.get("HGS_P2P_INJECT_FAILURE", "") == str(self.rank):     # tests: one rank cannot export
                raise RuntimeError("injected failure (HGS_P2P_INJECT_FAILURE)")
            _lib.check(lib.hgs_p2p_alloc(padded * 4, fine, C.byref(own_buf), self.dev_index), "hgs_p2p_alloc")
            self._own.append(own_buf.value)       # (a failing flag-block allocation must not leak the bucket)
            _lib.check(lib.hgs_p2p_alloc(_lib.P2P_FLAG_BYTES, 1, C.byref(own_flag), self.dev_index), "hgs_p2p_alloc")
            self._own.append(own_flag.value)
            hb, hf = C.create_string_buffer(_lib.P2P_HANDLE_BYTES), C.create_string_buffer(_lib.P2P_HANDLE_BYTES)
            _lib.check(lib.hgs_p2p_export(own_buf, hb, self.dev_index), "hgs_p2p_export")
            _lib.check(lib.hgs_p2p_export(own_flag, hf, self.dev_index), "hgs_p2p_export")
            mine = (hb.raw, hf.raw)
        except RuntimeError as e:
            err = f"rank {self.rank}: {e}"
        handles = [None] * self.world
        dist.all_gather_object(handles, (mine, err))
        failed = [e for _, e in handles if e]
        bufs, flags = (C.c_void_p * self.world)(), (C.c_void_p * self.world)()
        if not failed:
            try:
                for k, ((kb, kf), _) in enumerate(handles):
                    if k == self.rank:
                        bufs[k], flags[k] = own_buf.value, own_flag.value
                        continue
                    pb, pf = C.c_void_p(), C.c_void_p()
                    _lib.check(lib.hgs_p2p_open(kb, C.byref(pb), self.dev_index), "hgs_p2p_open")
                    self._opened.append(pb.value)
                    _lib.check(lib.hgs_p2p_open(kf, C.byref(pf), self.dev_index), "hgs_p2p_open")
                    self._opened.append(pf.value)
                    bufs[k], flags[k] = pb.value, pf.value
            except RuntimeError as e:
                err = f"rank {self.rank}: {e}"
            outcomes = [None] * self.world
            dist.all_gather_object(outcomes, err)
            failed = [e for e in outcomes if e]
        if failed:
            self._release()
            raise DirectRouteUnavailable("; ".join(failed))
        self._bufs, self._flags = bufs, flags
        self.flat = torch.as_tensor(_DeviceArray(own_buf.value, padded), device=self.device)[:self.numel]
        self._flag_t = torch.as_tensor(_DeviceArray(own_flag.value, _lib.P2P_FLAG_BYTES // 4), device=self.device)
        self.epoch = 0
        dist.barrier()          # every rank has opened every handle before anybody starts

    def all_reduce(self, offset: int = 0, numel: Optional[int] = None, stream: Optional[torch.cuda.Stream] = None):
        """Enqueue the all-reduce of flat[offset : offset + numel] (multiples of 4 floats; the tail of the bucket is
        padded) on ``stream`` (default: the current one).  Every rank must make the same calls in the same order."""
        n = self.numel - offset if numel is None else int(numel)
        if offset % 4:
            raise ValueError("offset must be a multiple of 4 floats")
        n = (n + 3) // 4 * 4
        self.epoch += 1
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._lib.check(self._lib.lib().hgs_p2p_allreduce_sum(self.rank, self.world, self._bufs, self._flags, int(offset),
                                                              n, self.epoch, self._C.c_void_p(st.cuda_stream),
                                                              self.dev_index), "hgs_p2p_allreduce_sum")

    def check(self):
        """Raise if a barrier timed out (host sync).  The error word is sticky and the device side has already
        replaced this rank's sums by NaN (csrc/p2p.hip): the step that timed out can not have been applied silently."""
        if int(self._flag_t.view(torch.int32)[3].item()) != 0:
            raise RuntimeError("direct all-reduce: a peer did not reach a barrier within HGS_P2P_TIMEOUT_S; the "
                               "gradient bucket of this step holds NaN -- the job must stop")

    def _release(self):
        lib = self._lib.lib()
        for ptr in self._opened:
            lib.hgs_p2p_close(self._C.c_void_p(ptr), self.dev_index)
        self._opened = []
        if self._own is not None:
            self.flat = self._flag_t = None
            for ptr in self._own:
                if ptr:
                    lib.hgs_p2p_free(self._C.c_void_p(ptr), self.dev_index)
            self._own = None

    def close(self):
        torch.cuda.synchronize(self.device)
        dist.barrier()          # nobody still reads this rank's memory
        self._release()


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
            if self._views == 0:
                # a rank that rendered no view this step (fewer views than ranks, or the uneven last round of
                # shard_views) still holds the previous step's REDUCED gradients: it must contribute zeros
                self.bucket.flat.zero_()
                self.means2D_grad.zero_()
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
        self.bucket.check_direct()      # direct route: a barrier timeout is fatal (periodic host check; NaN on device)
        return self._views


def shard_rows(P: int, rank: int, world: int):
    """Rows [r0, r1) of the Gaussian arrays owned by ``rank`` in the sharded step: equal blocks of ceil(P / world) rows
    (what reduce-scatter / all-gather move), the last block cut at P."""
    per = (P + world - 1) // world
    r0 = min(rank * per, P)
    return r0, min(r0 + per, P), per


class ShardedDataParallelStep(DataParallelStep):
    """The data-parallel step with the optimizer SHARDED over the ranks ("what comes next" of round
    3): instead of all-reduce + the same full Adam step on every rank,

        reduce-scatter of every gradient tensor      rank r receives the SUM of ITS rows only
        fused Adam on the rank's rows                state (exp_avg, exp_avg_sq) and HBM traffic of the optimizer / N
        all-gather of the updated rows               every rank ends with the same parameters

    The wire carries the same bytes as the all-reduce (a ring all-reduce IS reduce-scatter + all-gather), but the
    all-gather moves PARAMETERS, so it can run under whatever follows the step, and the optimizer -- 28 bytes of HBM
    traffic per updated element, 0.28 ms dense at 59 M elements -- shrinks with N.  Row r's update depends on row r's
    reduced gradient only (Adam is element-wise, ``relevant`` is taken per row from the reduced opacity gradient,
    train_single.py:170-174), so the result equals the all-reduce route's: bit for bit whenever the two collectives sum
    in the same order (always at 2 ranks).

    ``params`` as for DataParallelStep (full, replicated leaf tensors).  ``make_optimizer(shard_params)`` builds the
    optimizer over THIS RANK'S ROWS: ``shard_params[name]`` are detached views of rows [r0, r1) of ``params[name]``
    (the update lands in the full tensors), their ``.grad`` the reduced shard gradients.  Parameters whose row count
    is not a multiple of the world size take a padded staging buffer for the all-gather (one extra copy per step)."""

    def __init__(self, params: Dict[str, torch.Tensor], make_optimizer, backward_stream=None, make_context=None):
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        P = params["means3D"].shape[0]
        self.world, self.rank = world, rank
        self.r0, self.r1, self.per = shard_rows(P, rank, world)
        self.P, self.Pp = P, self.per * world
        dev = params["means3D"].device
        self.params = params
        self.optimizer = None
        # gradient bucket with the row count padded to a multiple of the world size (equal reduce-scatter blocks); the
        # op's backward writes the first P rows, the pad rows stay zero
        self.bucket = GradBucket({k: (self.Pp,) + tuple(v.shape[1:]) for k, v in params.items()}, dev, direct=False)
        self.full_views = {k: self.bucket.views[k][:P] for k in params}
        for k, v in params.items():
            v.grad = self.full_views[k]
        self.shard_grads = {k: torch.zeros((self.per,) + tuple(v.shape[1:]), dtype=torch.float32, device=dev)
                            for k, v in params.items()}
        n_own = self.r1 - self.r0
        self.shard_params = {}
        for k, v in params.items():
            sp = v.detach()[self.r0:self.r1]
            sp.grad = self.shard_grads[k][:n_own]
            self.shard_params[k] = sp
        self._stage = {}                    # padded all-gather buffers, only when P % world != 0
        if self.Pp != P and world > 1:
            self._stage = {k: torch.zeros((self.Pp,) + tuple(v.shape[1:]), dtype=torch.float32, device=dev)
                           for k, v in params.items()}
        self.optimizer = make_optimizer(self.shard_params)
        self.means2D_grad = torch.zeros(P, 3, dtype=torch.float32, device=dev)
        self.stats = DensifyStats(P, dev)
        self.backward_stream = backward_stream
        if make_context is None:
            import diff_gaussian_rasterization as dgr
            make_context = dgr.RasterContext
        self.context = make_context(grad_buffers=dict(self.full_views, means2D=self.means2D_grad),
                                    backward_stream=backward_stream)
        self._views = 0

    def _reduce_scatter(self, names):
        if self.world == 1:
            for k in names:
                self.shard_grads[k].copy_(self.bucket.views[k])
            return []
        return [dist.reduce_scatter_tensor(self.shard_grads[k], self.bucket.views[k], op=dist.ReduceOp.SUM, async_op=True)
                for k in names]

    def _all_gather(self, names):
        if self.world == 1:
            return []
        works = []
        for k in names:
            if self._stage:
                st = self._stage[k]
                st[self.r0:self.r1].copy_(self.shard_params[k])
                works.append(dist.all_gather_into_tensor(st, st[self.rank * self.per:(self.rank + 1) * self.per],
                                                         async_op=True))
            else:                           # in place: this rank's block of the full tensor is the input
                full = self.params[k].detach()
                works.append(dist.all_gather_into_tensor(full, full[self.r0:self.r1], async_op=True))
        return works

    def finish(self, sh_backward=None):
        with self._on_backward_stream():
            if self._views == 0:
                self.bucket.flat.zero_()
                self.means2D_grad.zero_()
            early = self._reduce_scatter(self.EARLY)             # ordered after the last view's backward
            if sh_backward is not None:
                sh_backward()
            late = self._reduce_scatter(self.LATE)
            self.stats.all_reduce()
            for w in early:
                w.wait()
            mask = self.shard_grads["opacities"][:self.r1 - self.r0]
            self.optimizer.step_masked(mask, params=[self.shard_params[k] for k in self.EARLY])
            ag = self._all_gather(self.EARLY)                    # on the wire while the late half is still reduced / stepped
            for w in late:
                w.wait()
            self.optimizer.step_masked(mask, params=[self.shard_params[k] for k in self.LATE])
            ag += self._all_gather(self.LATE)
            for w in ag:
                w.wait()
            if self._stage:
                with torch.no_grad():
                    for k, v in self.params.items():
                        v.detach().copy_(self._stage[k][:self.P])
        self.context.wait_backward_stream()
        return self._views
