"""VRAM-budgeted residency of a hierarchy's attribute rows ("VRAM-budgeted streaming LOD", BASELINE configs[4]; the
``--budget <MB>`` of the reference's hierarchy viewer, README.md:233-235, whose implementation is in the un-vendored
SIBR viewer).  Opt-in, beside the drop-in path: ``render_hierarchy.py`` itself loads the whole hierarchy onto the GPU
(scene/gaussian_model.py:329,376-399) and so does ``bench.py``'s configs[4] loop -- 15 GB of 288 GB.

The full attributes live in pinned host memory that the GPU can read directly, as ONE PACKED ROW of 64 floats per
Gaussian (SH, rotation, mean, scale, opacity: 4 (3 M + 11) useful bytes in 256 -- four 64-byte PCIe reads per row instead
of seven from five separate arrays); the GPU holds ``budget`` rows in slot arrays plus one int32 per Gaussian (its
slot, or "absent").  Per view::

    sel = bh.select(nodes, boxes, tau, viewpoint_gpu, viewpoint_cpu)   # cut, weights, residency; raises tau if needed
    rs  = GaussianRasterizationSettings(..., render_indices=sel.render_indices, parent_indices=sel.parent_indices,
                                        interpolation_weights=sel.weights, num_node_kids=sel.kids)
    GaussianRasterizer(rs)(means3D=bh.means3D, shs=bh.shs, opacities=bh.opacities, scales=bh.scales,
                           rotations=bh.rotations, means2D=...)

``select`` runs the reference's two LOD calls (``expand_to_size`` / ``get_interpolation_weights``), marks the rows the
cut needs (the node row of every entry, and its parent row unless the entry's weight is exactly 1 -- the in-op LOD gather
does not read that parent), fetches the missing ones over PCIe with ONE kernel that reads the host
arrays itself (no host-side gather, no staging buffer), recycles the slots that have gone unused for the longest when the
free list runs out, and returns the cut's indices translated to slots.  A view whose rows do not fit the budget is cut
again at a coarser granularity (tau x 1.2 per attempt), as the reference's viewer "auto-regulates and raises the
granularity until the scene can fit inside the defined VRAM budget".  The rasterizer's in-op LOD path runs on the slot
arrays unchanged: rows are rows."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib


def _host_array(shape, dtype=np.float32):
    """A numpy array over pinned, device-mapped host memory (hgs_host_alloc) and the owner that frees it."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = _lib.lib().hgs_host_alloc(n)
    if not p:
        raise RuntimeError(f"cannot allocate {n} bytes of pinned host memory")
    buf = (C.c_char * max(n, 1)).from_address(p)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    return arr, p


@dataclass
class Selection:
    """What one view renders.  The index / weight tensors are views of buffers the BudgetedHierarchy owns: valid until
    its next ``select`` / ``make_resident`` (enqueue the render first -- stream order does the rest)."""
    n: int                              # entries of the cut
    tau: float                          # the granularity that was rendered (>= the requested one)
    render_indices: torch.Tensor        # int32 [n]: SLOT of the node row
    parent_indices: torch.Tensor        # int32 [n]: SLOT of the parent row
    weights: torch.Tensor               # f32 [>= n]
    kids: torch.Tensor                  # int32 [>= n]
    misses: int                         # rows fetched for this view
    attempts: int                       # cuts tried (1 = the requested granularity fitted)


class BudgetedHierarchy:
    def __init__(self, means3D, shs, opacities, scales, rotations, device, budget_mb: Optional[float] = None,
                 budget_rows: Optional[int] = None, index_capacity: Optional[int] = None):
        """The five attribute arrays as CPU tensors ([G,3], [G,M,3], [G] or [G,1], [G,3], [G,4], float32, already in the
        form the rasterizer takes: activated).  They are COPIED into pinned host memory.  ``budget_mb``: megabytes of
        GPU memory for the attribute rows (the reference's ``--budget``); or ``budget_rows`` directly."""
        self.dev = torch.device(device)
        self.lib = _lib.lib()
        G = int(means3D.shape[0])
        M = int(shs.shape[1])
        self.G, self.M = G, M
        self.row_bytes = 4 * (3 * M + 11)
        if budget_rows is None:
            if budget_mb is None:
                raise ValueError("budget_mb or budget_rows")
            budget_rows = int(budget_mb * 1e6 // self.row_bytes)
        self.B = B = max(1, min(int(budget_rows), G))
        R = _lib.RESID_HOST_ROW_FLOATS
        assert 3 * M <= 48
        self._rows, self._rows_ptr = _host_array((G, R))           # the packed host rows (include/hgs.h)
        f = lambda t, shape: t.detach().to("cpu", torch.float32).reshape(shape).numpy()
        rows = self._rows
        rows[:, :3 * M] = f(shs, (G, 3 * M))
        rows[:, 3 * M:48] = 0.0
        rows[:, 48:52] = f(rotations, (G, 4))
        rows[:, 52:55] = f(means3D, (G, 3))
        rows[:, 55:58] = f(scales, (G, 3))
        rows[:, 58] = f(opacities, (G,))
        rows[:, 59:] = 0.0
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.means3D = torch.zeros(B, 3, **f32)
        self.shs = torch.zeros(B, M, 3, **f32)
        self.opacities = torch.zeros(B, 1, **f32)
        self.scales = torch.ones(B, 3, **f32)
        self.rotations = torch.zeros(B, 4, **f32)
        self.rotations[:, 0] = 1.0
        i32 = dict(dtype=torch.int32, device=self.dev)
        self.slot_of = torch.full((G,), -1, **i32)
        self.id_of_slot = torch.full((B,), -1, **i32)
        self.stamp = torch.zeros(B, **i32)
        self.free_list = torch.arange(B - 1, -1, -1, **i32)
        self.free_top = B
        self.counters = torch.zeros(_lib.RESID_COUNTER_WORDS, **i32)
        cap = int(index_capacity or G)
        self.ri = torch.zeros(cap, **i32); self.pi = torch.zeros(cap, **i32); self.ni = torch.zeros(cap, **i32)
        self.ro = torch.zeros(cap, **i32); self.po = torch.zeros(cap, **i32)
        self.miss_ids = torch.zeros(2 * cap, **i32)
        self.w = torch.zeros(cap, dtype=torch.float32, device=self.dev)
        self.ns = torch.zeros(cap, **i32)
        self.frame = 0
        self._regulated = None          # granularity the previous view was coarsened to (None: the request fitted)
        self._skip_batch = 0            # evictions left that skip the batch attempt (it failed recently)
        self._since_probe, self.probe_every = 0, 16
        self.stats = dict(views=0, rows_fetched=0, bytes_fetched=0, evictions=0, retries=0)
        self.profile_fetch = False      # True: (rows, start event, end event) of every fetch launch -> self.fetch_events
        self.fetch_events = []
        self._slot_rows = _lib.ResidRows(*[C.c_void_p(t.data_ptr()) for t in
                                           (self.means3D, self.shs, self.opacities, self.scales, self.rotations)])

    @classmethod
    def from_hier_file(cls, path: str, device, budget_mb: Optional[float] = None, budget_rows: Optional[int] = None):
        """A ``.hier`` file (gaussian_hierarchy._C.load_hierarchy, scene/gaussian_model.py:329) straight into the
        budgeted form, with the activations the reference applies to a loaded hierarchy: opacity = |alpha|
        (scene/gaussian_model.py:393), scales = exp(log-scales), rotations normalised (scene/gaussian_model.py:108-116).
        Returns (BudgetedHierarchy, nodes, boxes) with nodes / boxes on ``device`` (they stay resident: 60 B per node)."""
        from gaussian_hierarchy._C import load_hierarchy
        xyz, shs, alpha, log_scales, rots, nodes, boxes = load_hierarchy(path)
        bh = cls(xyz, shs, alpha.abs(), torch.exp(log_scales), torch.nn.functional.normalize(rots), device,
                 budget_mb=budget_mb, budget_rows=budget_rows)
        return bh, nodes.to(device), boxes.to(device)

    def __del__(self):
        try:
            p = getattr(self, "_rows_ptr", None)
            if p:
                self._rows = None
                self.lib.hgs_host_free(C.c_void_p(p))
                self._rows_ptr = None
        except Exception:
            pass

    # ---------------------------------------------------------------------------------------------------------------
    @property
    def budget_bytes(self):
        return self.B * self.row_bytes

    @property
    def resident_rows(self):
        return self.B - self.free_top

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def make_resident(self, render_indices: torch.Tensor, parent_indices: torch.Tensor,
                      weights: Optional[torch.Tensor] = None):
        """Rows of a cut (int32 GPU tensors of Gaussian rows, equal length) -> (slots of the node rows, slots of the
        parent rows, rows fetched).  ``weights`` (float32 GPU tensor, one interpolation weight per entry, optional): the
        parent row of an entry of weight exactly 1 is not read by the rasterizer's in-op LOD gather and is therefore
        neither fetched nor stamped -- its slot is reported as the node's own.  Raises _lib.HgsError with code
        HGS_ERR_CAPACITY when the rows do not fit the budget."""
        n = int(render_indices.numel())
        assert parent_indices.numel() >= n and n <= self.ro.numel()
        if weights is not None:
            assert weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous() and weights.numel() >= n
        p, dev_i, s = _lib.ptr, self.dev.index or 0, self._stream()
        self.frame += 1
        miss = C.c_uint32(0)

        def unqueue():
            # the rows queued by the mark pass (slot_of = -2) go back to "absent"
            k = int(miss.value)
            if k:
                ids = self.miss_ids[:k].long()
                self.slot_of[ids] = torch.where(self.slot_of[ids] == -2, torch.full_like(self.slot_of[ids], -1),
                                                self.slot_of[ids])

        try:
            _lib.check(self.lib.hgs_resid_mark(p(render_indices), p(parent_indices), p(weights), n, self.G, p(self.slot_of),
                                               p(self.stamp), self.frame, p(self.miss_ids), p(self.counters), p(self.ro),
                                               p(self.po), C.byref(miss), s, dev_i), "hgs_resid_mark")
        except _lib.HgsError:
            unqueue()                   # a bad index is reported after the valid rows of the cut were queued
            raise
        m = int(miss.value)
        if m:
            if m > 4096:
                # slots are handed out in miss-list order: sorted by row, a bulk fetch (cold start, a jump of the camera)
                # lays the rows out in the hierarchy's own order and K1's gathers stay as local as on the full arrays
                self.miss_ids[:m] = torch.sort(self.miss_ids[:m]).values
            try:
                if m > self.free_top:
                    # An eviction costs two passes over the slots and two host round trips: free a batch (1 / 32 of
                    # the budget) beyond what this view needs, so that a camera in motion evicts every few frames
                    # instead of on every frame; if that many old rows do not exist, exactly what is needed.
                    # A view whose working set nearly fills the budget has no such batch to give: after a failed
                    # batch attempt the next 16 evictions ask for exactly what they need (one pass, one round trip).
                    batch = max(m, min(self.B, m + self.B // 32))
                    if self._skip_batch > 0:
                        self._skip_batch -= 1
                        batch = m
                    for need in dict.fromkeys((batch, m)):
                        top = C.c_uint32(self.free_top)
                        rc = self.lib.hgs_resid_evict(p(self.stamp), p(self.id_of_slot), p(self.slot_of), self.B,
                                                      self.frame, need, p(self.free_list), p(self.counters),
                                                      C.byref(top), s, dev_i)
                        if rc == _lib.ERR_CAPACITY and need > m:
                            self._skip_batch = 16
                            continue
                        _lib.check(rc, "hgs_resid_evict")
                        break
                    self.stats["evictions"] += int(top.value) - self.free_top
                    self.free_top = int(top.value)
                if self.profile_fetch:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                _lib.check(self.lib.hgs_resid_fetch(p(self.miss_ids), m, p(self.free_list), self.free_top, p(self.slot_of),
                                                    p(self.id_of_slot), p(self.stamp), self.frame,
                                                    C.c_void_p(self._rows_ptr), C.byref(self._slot_rows), self.M, s, dev_i),
                           "hgs_resid_fetch")
                if self.profile_fetch:
                    e1.record()
                    self.fetch_events.append((m, e0, e1))
            except _lib.HgsError:
                unqueue()
                raise
            self.free_top -= m
            _lib.check(self.lib.hgs_resid_remap(p(render_indices), p(parent_indices), p(weights), n, p(self.slot_of),
                                                p(self.ro), p(self.po), s, dev_i), "hgs_resid_remap")
            self.stats["rows_fetched"] += m
            self.stats["bytes_fetched"] += m * self.row_bytes
        return self.ro[:n], self.po[:n], m

    def select(self, nodes, boxes, tau, viewpoint_gpu, viewpoint_cpu, max_attempts: int = 96, growth: float = 1.2) -> Selection:
        """expand_to_size + get_interpolation_weights at ``tau`` (train_post.py:91-113, render_hierarchy.py:58-80), the
        cut's rows made resident; a cut that does not fit the budget is repeated at ``growth`` x tau (from 1e-4 when the
        request was tau = 0: every leaf)."""
        from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
        zero3 = torch.zeros(3)
        t = float(tau)
        probing = False
        if self._regulated is not None and self._regulated > t:
            # the previous view had to be coarsened: start from what fitted then, and only every `probe_every`-th view
            # one step finer (a cut that does not fit costs a cut, its weights and a pass over its rows with every miss
            # queued and taken back: 10 ms at 25 M entries).  A probe that fails doubles the interval (up to 256 views), one
            # that fits resets it: a camera that stays in a region the budget cannot show finer stops paying for asking.
            self._since_probe += 1
            probing = self._since_probe >= self.probe_every
            if probing:
                self._since_probe = 0
            t = max(t, self._regulated / growth if probing else self._regulated)
        for attempt in range(1, max_attempts + 1):
            n = expand_to_size(nodes, boxes, t, viewpoint_gpu, zero3, self.ri, self.pi, self.ni)
            try:
                if n > self.B:          # more node rows than slots: no need to look at them
                    raise _lib.HgsError(f"a cut of {n} entries cannot fit a budget of {self.B} rows", _lib.ERR_CAPACITY)
                # the weights first: an entry of weight 1 does not need its parent row (make_resident)
                get_interpolation_weights(self.ni[:n], t, nodes, boxes, viewpoint_cpu, zero3, self.w, self.ns)
                ro, po, m = self.make_resident(self.ri[:n], self.pi[:n], self.w)
            except _lib.HgsError as e:
                if e.code != _lib.ERR_CAPACITY:
                    raise
                self.stats["retries"] += 1
                if probing and attempt == 1:
                    self.probe_every = min(2 * self.probe_every, 256)
                t = t * growth if t > 0 else 1e-4
                continue
            if probing and attempt == 1:
                self.probe_every = 16
            self.stats["views"] += 1
            self._regulated = t if t > float(tau) else None
            return Selection(n, t, ro, po, self.w, self.ns, m, attempt)
        raise RuntimeError(f"no granularity up to tau = {t:g} fits a budget of {self.B} rows")
