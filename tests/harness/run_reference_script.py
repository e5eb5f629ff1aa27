"""Launch one of the reference's entry scripts (train_single.py, train_post.py, render_hierarchy.py ...) UNMODIFIED as
``__main__`` on a machine without a GPU:

    python tests/harness/run_reference_script.py [--backend cpu|hip] train_single.py -s <scene> --model_path <out> ...

``--backend hip`` (wherever a GPU AND a reference checkout exist; HGS_REFERENCE names the checkout): the scripts meet
the REAL packages -- libhgs.so's HIP kernels through diff_gaussian_rasterization / gaussian_hierarchy._C /
simple_knn._C, real ``torch.cuda`` -- and only ``sys.path`` is arranged.  ``--backend cpu`` (default, the build
container):

What this launcher adds around the script, and nothing else:
  * ``sys.path``: the reference checkout, this repository's drop-in packages (diff_gaussian_rasterization,
    gaussian_hierarchy, simple_knn) and the plyfile / cv2 / torchvision shims of tests/shims;
  * the oracle-backed extension layers of tests/harness/cpu_backends.py (TEST ONLY: no GPU here);
  * a torch-function mode that maps ``device="cuda"`` / ``.cuda()`` / ``.to("cuda")`` to the CPU, and no-op stand-ins for
    the handful of ``torch.cuda.*`` calls the scripts make (Event, max_memory_allocated, empty_cache, set_device).
The reference's files are executed from /root/reference as they are."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("HGS_REFERENCE", "/root/reference")

import torch
from torch.overrides import TorchFunctionMode


def _is_cuda(d):
    if isinstance(d, torch.device):
        return d.type == "cuda"
    return isinstance(d, str) and d.startswith("cuda")


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if _is_cuda(kwargs.get("device")):
            kwargs["device"] = "cpu"
        if func is torch.Tensor.cuda:
            return args[0]
        if func is torch.Tensor.to and any(_is_cuda(a) for a in args[1:]):
            args = (args[0],) + tuple("cpu" if _is_cuda(a) else a for a in args[1:])
        if func is torch.Tensor.pin_memory:
            return args[0]
        return func(*args, **kwargs)


class _Event:
    def __init__(self, *a, **k):
        pass

    def record(self, *a, **k):
        pass

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return 0.0


def main():
    backend = "cpu"
    if len(sys.argv) > 2 and sys.argv[1] == "--backend":
        backend = sys.argv[2]
        del sys.argv[1:3]
    if backend not in ("cpu", "hip"):
        raise SystemExit("--backend must be cpu or hip")
    script = sys.argv[1]
    sys.path[:0] = [REF, os.path.join(ROOT, "tests", "shims"), os.path.join(ROOT, "hierarchical-3d-gaussians_amd"), ROOT,
                    os.path.join(ROOT, "tests")]
    sys.argv = [os.path.join(REF, script)] + sys.argv[2:]
    if backend == "hip":
        # the acceptance sentence of BASELINE.json's north_star, literally: the unmodified script on the HIP op
        if not torch.cuda.is_available():
            raise SystemExit("--backend hip needs a GPU")
        from hgs import _lib
        _lib.lib()                                  # fail loudly if libhgs.so is absent: there is no fallback
        runpy.run_path(os.path.join(REF, script), run_name="__main__")
        return
    from harness import cpu_backends
    cpu_backends.install()
    torch.cuda.Event = _Event
    torch.cuda.max_memory_allocated = lambda *a, **k: 0
    torch.cuda.empty_cache = lambda: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    with CudaToCpu():
        runpy.run_path(os.path.join(REF, script), run_name="__main__")


if __name__ == "__main__":
    main()
