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
again at a coarser granularity (tau x 1.2 per attempt, x 1.05 once the cut is within a tenth of the budget), as the
reference's viewer "auto-regulates and raises the granularity until the scene can fit inside the defined VRAM budget".
The rasterizer's in-op LOD path runs on the slot arrays unchanged: rows are rows.

``prefetch(nodes, boxes, tau, next_viewpoint_gpu, next_viewpoint_cpu)``, called once the CURRENT view's render has been
enqueued, runs the NEXT view's cut, weights and residency on a second stream while the render occupies the first: the rows
a camera jump needs (1.4 M rows = 360 MB over PCIe in the 50 M-node fly-through) cross the bus under the previous frame
instead of in front of the next one, and ``select`` for that viewpoint reuses the prefetched cut."""
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


class _CutBuffers:
    """Index / weight buffers of one cut (Gaussian rows in, slots out)."""

    def __init__(self, cap, dev):
        i32 = dict(dtype=torch.int32, device=dev)
        self.ri = torch.zeros(cap, **i32); self.pi = torch.zeros(cap, **i32); self.ni = torch.zeros(cap, **i32)
        self.ro = torch.zeros(cap, **i32); self.po = torch.zeros(cap, **i32)
        self.w = torch.zeros(cap, dtype=torch.float32, device=dev)
        self.ns = torch.zeros(cap, **i32)


@dataclass
class Selection:
    """What one view renders.  The index / weight tensors are views of buffers the BudgetedHierarchy owns: valid until
    its next ``select`` / ``make_resident`` (enqueue the render first -- stream order does the rest; a ``prefetch`` in
    between writes the other set of buffers)."""
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
        # two sets of cut buffers: the view being rendered reads one (its Selection aliases it), ``prefetch`` fills the other
        self._sets = [_CutBuffers(cap, self.dev), _CutBuffers(cap, self.dev)]
        self._cur = 0
        self.miss_ids = torch.zeros(2 * cap, **i32)
        self.frame = 0
        self._side = None               # the prefetch stream (created on first use)
        self._resid_done = None         # event: the residency kernels of the last select are enqueued (before its render)
        self._prefetch_done = None      # event: the last prefetch has finished on the side stream
        self._prefetched = None         # (viewpoint key, tau, n) of the cut waiting in the other buffer set
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

    # the CURRENT set's buffers under their old names (tests and callers read them)
    ri = property(lambda s: s._sets[s._cur].ri)
    pi = property(lambda s: s._sets[s._cur].pi)
    ni = property(lambda s: s._sets[s._cur].ni)
    ro = property(lambda s: s._sets[s._cur].ro)
    po = property(lambda s: s._sets[s._cur].po)
    w = property(lambda s: s._sets[s._cur].w)
    ns = property(lambda s: s._sets[s._cur].ns)

    def make_resident(self, render_indices: torch.Tensor, parent_indices: torch.Tensor,
                      weights: Optional[torch.Tensor] = None, _bufs=None, _new_frame=True, _best_effort=False):
        """Rows of a cut (int32 GPU tensors of Gaussian rows, equal length) -> (slots of the node rows, slots of the
        parent rows, rows fetched).  ``weights`` (float32 GPU tensor, one interpolation weight per entry, optional): the
        parent row of an entry of weight exactly 1 is not read by the rasterizer's in-op LOD gather and is therefore
        neither fetched nor stamped -- its slot is reported as the node's own.  Raises _lib.HgsError with code
        HGS_ERR_CAPACITY when the rows do not fit the budget."""
        n = int(render_indices.numel())
        bufs = self._sets[self._cur] if _bufs is None else _bufs
        ro_buf, po_buf = bufs.ro, bufs.po
        assert parent_indices.numel() >= n and n <= ro_buf.numel()
        if weights is not None:
            assert weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous() and weights.numel() >= n
        p, dev_i, s = _lib.ptr, self.dev.index or 0, self._stream()
        if _new_frame:                  # (a prefetch stamps with the frame being rendered: its rows are protected with it)
            self.frame += 1
        miss = C.c_uint32(0)

        def unqueue(first=0):
            # the rows queued by the mark pass (slot_of = -2) go back to "absent" (from entry `first` of the miss list on)
            k = int(miss.value)
            if k > first:
                ids = self.miss_ids[first:k].long()
                self.slot_of[ids] = torch.where(self.slot_of[ids] == -2, torch.full_like(self.slot_of[ids], -1),
                                                self.slot_of[ids])

        try:
            _lib.check(self.lib.hgs_resid_mark(p(render_indices), p(parent_indices), p(weights), n, self.G, p(self.slot_of),
                                               p(self.stamp), self.frame, p(self.miss_ids), p(self.counters), p(ro_buf),
                                               p(po_buf), C.byref(miss), s, dev_i), "hgs_resid_mark")
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
                    # best effort (a prefetch: the rows of the view being rendered carry this frame's stamp and stay): what
                    # cannot be freed is left to the next select
                    tries = (batch, m) if not _best_effort else (m, max(1, m // 2), max(1, m // 4), max(1, m // 8))
                    for need in dict.fromkeys(tries):
                        top = C.c_uint32(self.free_top)
                        rc = self.lib.hgs_resid_evict(p(self.stamp), p(self.id_of_slot), p(self.slot_of), self.B,
                                                      self.frame, need, p(self.free_list), p(self.counters),
                                                      C.byref(top), s, dev_i)
                        if rc == _lib.ERR_CAPACITY and (need > m or _best_effort):
                            if not _best_effort:
                                self._skip_batch = 16
                            continue
                        _lib.check(rc, "hgs_resid_evict")
                        break
                    self.stats["evictions"] += int(top.value) - self.free_top
                    self.free_top = int(top.value)
                if _best_effort and m > self.free_top:
                    unqueue(self.free_top)
                    m = self.free_top
                    if m == 0:
                        return None, None, 0
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
            if _best_effort and m < int(miss.value):
                self.stats["rows_fetched"] += m
                self.stats["bytes_fetched"] += m * self.row_bytes
                return None, None, m            # (part of the view is still missing: no slot indices)
            _lib.check(self.lib.hgs_resid_remap(p(render_indices), p(parent_indices), p(weights), n, p(self.slot_of),
                                                p(ro_buf), p(po_buf), s, dev_i), "hgs_resid_remap")
            self.stats["rows_fetched"] += m
            self.stats["bytes_fetched"] += m * self.row_bytes
        return ro_buf[:n], po_buf[:n], m

    @staticmethod
    def _vp_key(viewpoint_cpu):
        return tuple(float(x) for x in viewpoint_cpu.reshape(-1)[:3])

    def _join_prefetch(self):
        """The current stream waits for a prefetch in flight (it owns slot_of / stamp / the free list until it is done)."""
        if self._prefetch_done is not None:
            torch.cuda.current_stream(self.dev).wait_event(self._prefetch_done)
            self._prefetch_done = None

    def _start_tau(self, tau, fine_growth):
        """Where the regulator starts for a request of ``tau``: (granularity, probing a finer step?)."""
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
            t = max(t, self._regulated / fine_growth if probing else self._regulated)
        return t, probing

    def _fit(self, nodes, boxes, tau, t, probing, viewpoint_gpu, viewpoint_cpu, bufs, new_frame, max_attempts, growth,
             fine_growth, reuse_n=None):
        """Cut + weights + residency into ``bufs``, coarsening until the rows fit: (n, t, ro, po, rows fetched, attempts).
        ``reuse_n``: the cut and its weights at ``t`` are already in ``bufs`` (a prefetch left them)."""
        from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
        zero3 = torch.zeros(3)
        for attempt in range(1, max_attempts + 1):
            reuse = attempt == 1 and reuse_n is not None
            n = reuse_n if reuse else expand_to_size(nodes, boxes, t, viewpoint_gpu, zero3, bufs.ri, bufs.pi, bufs.ni)
            near = n <= 1.1 * self.B
            try:
                if n > self.B:          # more node rows than slots: no need to look at them
                    raise _lib.HgsError(f"a cut of {n} entries cannot fit a budget of {self.B} rows", _lib.ERR_CAPACITY)
                # the weights first: an entry of weight 1 does not need its parent row (make_resident)
                if not reuse:
                    get_interpolation_weights(bufs.ni[:n], t, nodes, boxes, viewpoint_cpu, zero3, bufs.w, bufs.ns)
                ro, po, m = self.make_resident(bufs.ri[:n], bufs.pi[:n], bufs.w, _bufs=bufs, _new_frame=new_frame)
            except _lib.HgsError as e:
                if e.code != _lib.ERR_CAPACITY:
                    raise
                self.stats["retries"] += 1
                if probing and attempt == 1:
                    self.probe_every = min(2 * self.probe_every, 256)
                t = t * (fine_growth if near else growth) if t > 0 else 1e-4
                continue
            if probing and attempt == 1:
                self.probe_every = 16
            self._regulated = t if t > float(tau) else None
            return n, t, ro, po, m, attempt
        raise RuntimeError(f"no granularity up to tau = {t:g} fits a budget of {self.B} rows")

    def prefetch(self, nodes, boxes, tau, viewpoint_gpu, viewpoint_cpu) -> int:
        """The NEXT view's cut, weights and residency on a second stream, to be called right after the current view's
        render was enqueued (its pose known or predicted: a viewer extrapolates its camera).  BEST EFFORT: rows the
        current view uses are never evicted (they carry the current frame's stamp; so do the rows fetched here), nothing
        the render reads is written -- free slots and slots of older frames are filled, the other set of cut buffers
        receives the indices -- and no granularity is changed.  When everything the next view needs became resident,
        ``select`` for the same viewpoint and request starts from this cut (its mark pass only stamps the rows); otherwise
        it finds that many fewer rows missing.  Returns the rows fetched."""
        from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        side, zero3 = self._side, torch.zeros(3)
        self._join_prefetch()
        if self._resid_done is not None:
            side.wait_event(self._resid_done)       # the last select's residency kernels -- NOT the render behind them
        else:
            side.wait_stream(torch.cuda.current_stream(self.dev))
        other = self._sets[1 - self._cur]
        self._prefetched = None
        t = max(float(tau), self._regulated or 0.0)
        m = 0
        with torch.cuda.stream(side):
            try:
                n = expand_to_size(nodes, boxes, t, viewpoint_gpu, zero3, other.ri, other.pi, other.ni)
                if n <= self.B:
                    get_interpolation_weights(other.ni[:n], t, nodes, boxes, viewpoint_cpu, zero3, other.w, other.ns)
                    ro, _, m = self.make_resident(other.ri[:n], other.pi[:n], other.w, _bufs=other, _new_frame=False,
                                                  _best_effort=True)
                    if ro is not None:
                        self._prefetched = (self._vp_key(viewpoint_cpu), float(tau), t, n)
                    self.stats["prefetched_rows"] = self.stats.get("prefetched_rows", 0) + m
            finally:
                self._prefetch_done = torch.cuda.Event()
                self._prefetch_done.record(side)
        return m

    def select(self, nodes, boxes, tau, viewpoint_gpu, viewpoint_cpu, max_attempts: int = 96, growth: float = 1.2,
               fine_growth: float = 1.05) -> Selection:
        """expand_to_size + get_interpolation_weights at ``tau`` (train_post.py:91-113, render_hierarchy.py:58-80), the
        cut's rows made resident; a cut that does not fit the budget is repeated at ``growth`` x tau (from 1e-4 when the
        request was tau = 0: every leaf) -- at ``fine_growth`` x tau once the cut is within a tenth of the budget, so that
        the regulator settles on the last few per cent of it."""
        self._join_prefetch()
        pre, self._prefetched = self._prefetched, None
        (t, probing), reuse_n = self._start_tau(tau, fine_growth), None
        if not probing and pre is not None and pre[0] == self._vp_key(viewpoint_cpu) and pre[1] == float(tau) and pre[2] == t:
            self._cur = 1 - self._cur   # the cut and its weights are waiting in the other buffer set
            reuse_n = pre[3]
        bufs = self._sets[self._cur]
        n, t, ro, po, m, attempt = self._fit(nodes, boxes, tau, t, probing, viewpoint_gpu, viewpoint_cpu, bufs, True,
                                             max_attempts, growth, fine_growth, reuse_n)
        self.stats["views"] += 1
        self._resid_done = torch.cuda.Event()
        self._resid_done.record(torch.cuda.current_stream(self.dev))
        return Selection(n, t, ro, po, bufs.w, bufs.ns, m, attempt)
