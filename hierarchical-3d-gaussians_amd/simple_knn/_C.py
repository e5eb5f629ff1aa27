"""``simple_knn._C.distCUDA2`` (called at scene/gaussian_model.py:190) over libhgs.so."""
import ctypes as C

import torch

from hgs import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """Mean squared distance of every point to its 3 nearest neighbours. points: [P,3] float32 GPU."""
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 needs a GPU tensor; there is no CPU path")
    pts = points.detach().to(torch.float32).contiguous()
    P = pts.shape[0]
    out = torch.zeros(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    lib = _lib.lib()
    tmp = torch.empty(lib.hgs_knn_tmp_bytes(P), dtype=torch.uint8, device=pts.device)
    _lib.check(lib.hgs_dist2_knn3(_lib.ptr(pts), P, _lib.ptr(out), _lib.ptr(tmp),
                                  C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream),
                                  pts.device.index or 0), "hgs_dist2_knn3")
    return out
