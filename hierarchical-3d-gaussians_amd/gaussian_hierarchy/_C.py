"""``gaussian_hierarchy._C`` over libhgs.so: the four functions the reference imports
(train_post.py:26, render_hierarchy.py:27, scene/gaussian_model.py:24)."""
from __future__ import annotations

import ctypes as C
import time
import weakref

import numpy as np
import torch

from hgs import _lib


# Opt-in (``set_viewpoint_cache(True)``): remember the host copy of a GPU viewpoint ON the tensor object.  A viewer loop
# that keeps its Camera objects then pays no device-to-host read per frame.  Off by default because the key --
# (data_ptr, version counter) -- cannot see writes that bypass the version counter (``t.data.copy_()``, ``set_()``,
# writes from external kernels or through DLPack): a caller that opts in promises not to make those.
_viewpoint_cache = False


def set_viewpoint_cache(on: bool) -> bool:
    """Enable / disable the per-tensor host copy of GPU viewpoints (see above); returns the previous setting."""
    global _viewpoint_cache
    prev, _viewpoint_cache = _viewpoint_cache, bool(on)
    return prev


def _vec3(t) -> "C.Array":
    """The C-ABI takes viewpoints as host floats; the reference passes ``camera_center`` as a GPU tensor
    (train_post.py:96, render_hierarchy.py:63).  Reading it back waits for everything enqueued before it -- the previous
    frame's render in a viewer loop -- so the wait polls the stream instead of sleeping in the blocking copy (the
    wake-up of a sleeping wait was measured at milliseconds on virtualised hosts); with ``set_viewpoint_cache(True)``
    the host copy is also remembered on the tensor object (an in-place edit through torch re-reads it)."""
    on_gpu = torch.is_tensor(t) and t.is_cuda
    if on_gpu:
        if _viewpoint_cache:
            cached = getattr(t, "_hgs_vec3", None)
            if cached is not None and cached[0] == (t.data_ptr(), t._version):
                return cached[1]
        stream = torch.cuda.current_stream(t.device)
        deadline = time.perf_counter() + 0.2
        while not stream.query() and time.perf_counter() < deadline:
            pass
    v = t.detach().to("cpu", torch.float32).reshape(-1) if torch.is_tensor(t) else torch.tensor(t, dtype=torch.float32)
    if v.numel() != 3:
        raise RuntimeError("expected a 3-vector")
    out = (C.c_float * 3)(*[float(x) for x in v])
    if on_gpu and _viewpoint_cache:
        try:
            t._hgs_vec3 = ((t.data_ptr(), t._version), out)
        except Exception:       # (a tensor subclass without a __dict__)
            pass
    return out


def _check_hier(nodes, boxes):
    if not nodes.is_cuda or not boxes.is_cuda:
        raise RuntimeError("nodes / boxes must be GPU tensors")
    if nodes.dtype != torch.int32 or nodes.dim() != 2 or nodes.shape[1] != 7 or not nodes.is_contiguous():
        raise RuntimeError("nodes must be a contiguous int32 [N,7] tensor")
    if boxes.dtype != torch.float32 or boxes.numel() != nodes.shape[0] * 8 or not boxes.is_contiguous():
        raise RuntimeError("boxes must be a contiguous float32 [N,2,4] tensor")


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


# (nodes ptr, boxes ptr, N, nodes version, boxes version) -> do the boxes nest?  (view independent: checked once per
# hierarchy; an in-place edit of either tensor bumps its version and triggers a re-check).  An entry lives only as
# long as BOTH tensor objects it was computed for: the caching allocator hands freed addresses to new tensors, whose
# version counters start at 0 again, so a key must never outlive its tensors (a finalizer on each evicts it).
_nested_cache = {}


def _boxes_nested(nodes, boxes) -> bool:
    key = (nodes.data_ptr(), boxes.data_ptr(), nodes.shape[0], nodes._version, boxes._version)
    hit = _nested_cache.get(key)
    if hit is None:
        lib = _lib.lib()
        dev = nodes.device
        tmp = torch.empty(4, dtype=torch.uint8, device=dev)
        out = C.c_int32(0)
        _lib.check(lib.hgs_hier_boxes_nested(_lib.ptr(nodes), _lib.ptr(boxes), nodes.shape[0], _lib.ptr(tmp),
                                             C.byref(out), _stream(dev), dev.index or 0), "hgs_hier_boxes_nested")
        if len(_nested_cache) > 16:
            _nested_cache.clear()
        hit = _nested_cache[key] = bool(out.value)
        for t in (nodes, boxes):
            weakref.finalize(t, _nested_cache.pop, key, None)
    return hit


def expand_to_size(nodes, boxes, size, viewpoint, viewdir, render_indices, parent_indices,
                   nodes_for_render_indices) -> int:
    """LOD cut for one view (train_post.py:91-99, render_hierarchy.py:58-66).  Fills the three
    preallocated int32 GPU arrays and returns how many entries are valid.  Hierarchies whose boxes nest (every one
    built by bounding-box union) take the single-pass kernel, anything else the level-by-level expansion."""
    _check_hier(nodes, boxes)
    for t in (render_indices, parent_indices, nodes_for_render_indices):
        if not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError("output index arrays must be contiguous int32 GPU tensors")
    lib = _lib.lib()
    N = nodes.shape[0]
    dev = nodes.device
    cap = min(render_indices.numel(), parent_indices.numel(), nodes_for_render_indices.numel())
    tmp = torch.empty(lib.hgs_expand_tmp_bytes(N), dtype=torch.uint8, device=dev)
    count = C.c_int32(0)
    p = _lib.ptr
    fn = lib.hgs_expand_to_size_nested if (N > 0 and _boxes_nested(nodes, boxes)) else lib.hgs_expand_to_size
    _lib.check(fn(p(nodes), p(boxes), N, float(size), _vec3(viewpoint), _vec3(viewdir),
                  p(render_indices), p(parent_indices), p(nodes_for_render_indices), cap,
                  p(tmp), C.byref(count), _stream(dev), dev.index or 0), "hgs_expand_to_size")
    return int(count.value)


def get_interpolation_weights(node_indices, size, nodes, boxes, viewpoint, viewdir, interpolation_weights,
                              num_siblings) -> None:
    """Blend weight and sibling count of every node of the cut (train_post.py:104-113)."""
    _check_hier(nodes, boxes)
    n = int(node_indices.numel())
    if n == 0:
        return
    if not node_indices.is_cuda or node_indices.dtype != torch.int32:
        raise RuntimeError("node_indices must be an int32 GPU tensor")
    if not interpolation_weights.is_cuda or interpolation_weights.dtype != torch.float32 or \
            not num_siblings.is_cuda or num_siblings.dtype != torch.int32:
        raise RuntimeError("interpolation_weights (float32) / num_siblings (int32) must be GPU tensors")
    if interpolation_weights.numel() < n or num_siblings.numel() < n:
        raise RuntimeError("output arrays are shorter than node_indices")
    node_indices = node_indices.contiguous()
    lib = _lib.lib()
    dev = nodes.device
    p = _lib.ptr
    _lib.check(lib.hgs_interp_weights(p(node_indices), n, float(size), p(nodes), p(boxes), nodes.shape[0],
                                      _vec3(viewpoint), _vec3(viewdir), p(interpolation_weights), p(num_siblings),
                                      _stream(dev), dev.index or 0), "hgs_interp_weights")


def load_hierarchy(path: str):
    """-> (xyz[P,3], shs[P,16,3], alpha[P,1], log_scales[P,3], rots[P,4], nodes[N,7] int32, boxes[N,2,4]) CPU
    tensors (scene/gaussian_model.py:329)."""
    lib = _lib.lib()
    h = _lib.HierHost()
    _lib.check(lib.hgs_hier_load(str(path).encode(), C.byref(h)), "hgs_hier_load")
    try:
        def arr(ptr, n, ctype, dtype):
            if n == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).astype(dtype, copy=True)
        P, N, M = h.P, h.N, h.M
        xyz = torch.from_numpy(arr(h.xyz, P * 3, C.c_float, np.float32)).view(P, 3)
        shs = torch.from_numpy(arr(h.shs, P * M * 3, C.c_float, np.float32)).view(P, M, 3)
        alpha = torch.from_numpy(arr(h.alpha, P, C.c_float, np.float32)).view(P, 1)
        scales = torch.from_numpy(arr(h.log_scales, P * 3, C.c_float, np.float32)).view(P, 3)
        rots = torch.from_numpy(arr(h.rots, P * 4, C.c_float, np.float32)).view(P, 4)
        nodes = torch.from_numpy(arr(h.nodes, N * 7, C.c_int32, np.int32)).view(N, 7)
        boxes = torch.from_numpy(arr(h.boxes, N * 8, C.c_float, np.float32)).view(N, 2, 4)
    finally:
        lib.hgs_hier_free(C.byref(h))
    return xyz, shs, alpha, scales, rots, nodes, boxes


def write_hierarchy(path: str, xyz, shs, alpha, log_scales, rots, nodes, boxes) -> None:
    """scene/gaussian_model.py:420-427 (tensors may live on the GPU)."""
    lib = _lib.lib()
    f = lambda t: np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
    a_xyz, a_shs, a_alpha, a_sc, a_rot = f(xyz), f(shs), f(alpha), f(log_scales), f(rots)
    a_nodes = np.ascontiguousarray(nodes.detach().to("cpu", torch.int32).numpy())
    a_boxes = f(boxes)
    P = a_xyz.shape[0]
    M = a_shs.shape[1] if a_shs.ndim == 3 else 0
    if a_shs.size != P * M * 3 or a_alpha.size != P or a_sc.size != P * 3 or a_rot.size != P * 4:
        raise RuntimeError("write_hierarchy: attribute arrays disagree on the number of Gaussians")
    N = a_nodes.shape[0]
    if a_nodes.size != N * 7 or a_boxes.size != N * 8:
        raise RuntimeError("write_hierarchy: nodes must be [N,7] and boxes [N,2,4]")
    h = _lib.HierHost()
    # the upstream tools' layout whenever it can express the data (16 SH coefficients), else the private one
    h.P, h.N, h.M, h.reserved = P, N, M, (_lib.HIER_UPSTREAM if M == 16 else _lib.HIER_PRIVATE)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    h.xyz, h.shs, h.alpha, h.log_scales, h.rots = vp(a_xyz), vp(a_shs), vp(a_alpha), vp(a_sc), vp(a_rot)
    h.nodes, h.boxes = vp(a_nodes), vp(a_boxes)
    _lib.check(lib.hgs_hier_write(str(path).encode(), C.byref(h)), "hgs_hier_write")
