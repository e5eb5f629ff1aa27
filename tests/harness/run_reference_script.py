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


def _install_stats(path):
    """HGS_HARNESS_STATS=<file>: what the op did while the UNMODIFIED script drove it (scripts/run_config2_config3.py).
    Nothing of the script is touched: the forward entry of this repository's own extension module is wrapped to count
    calls and to look at the clock / the allocator; the numbers are written when the interpreter exits."""
    import atexit
    import json
    import time
    import diff_gaussian_rasterization as dgr
    from hgs import _lib
    C = dgr._C
    st = {"calls": 0, "rows": [], "t_first": None, "mem_at_500": None, "window": None}
    orig_fwd, orig_plan = C.rasterize_gaussians, C._plan
    plan = {"queries": 0, "misses": 0}
    W0, W1 = int(os.environ.get("HGS_STATS_WINDOW0", "1200")), int(os.environ.get("HGS_STATS_WINDOW1", "1700"))

    def counted_plan(lib, P, W, H, L_ws):
        plan["queries"] += 1
        if (P, W, H, L_ws) not in C._plan_cache:
            plan["misses"] += 1
        return orig_plan(lib, P, W, H, L_ws)

    def mem():
        m = torch.cuda.memory_stats()
        return {k: m.get(k, 0) for k in ("allocated_bytes.all.current", "allocated_bytes.all.peak", "reserved_bytes.all.current",
                                         "reserved_bytes.all.peak", "num_alloc_retries", "segment.all.current",
                                         "inactive_split_bytes.all.current")}

    def counted_fwd(*a, **k):
        n = st["calls"] = st["calls"] + 1
        if n == 1:
            st["t_first"] = time.perf_counter()
        if n % 100 == 0 or n == 1:
            st["rows"].append((n, int(a[1].shape[0]), round(time.perf_counter() - st["t_first"], 3)))
        if n == 500:
            st["mem_at_500"] = mem()
        if n == W0:                                   # a window of steps with the op's stage timers on
            torch.cuda.synchronize()
            _lib.timing_read(True); _lib.timing_enable(True)
            st["window"] = {"t0": time.perf_counter(), "first_call": n}
        if n == W1 and st["window"] and "t1" not in st["window"]:
            torch.cuda.synchronize()
            _lib.timing_enable(False)
            w = st["window"]
            w["t1"], w["last_call"] = time.perf_counter(), n
            w["stages_ms_per_call"] = {kk: ms / max(c, 1) for kk, (ms, c) in _lib.timing_read(True).items() if c}
        return orig_fwd(*a, **k)

    C.rasterize_gaussians, C._plan = counted_fwd, counted_plan

    def dump():
        try:
            torch.cuda.synchronize()
            out = {"script": os.path.basename(sys.argv[0]), "forward_calls": st["calls"],
                   "wall_s_first_to_last_call": None if st["t_first"] is None else time.perf_counter() - st["t_first"],
                   "op_stats": dict(C.stats), "plan_cache": dict(plan, entries=len(C._plan_cache)),
                   "rows_by_call": st["rows"], "memory_at_call_500": st["mem_at_500"], "memory_at_exit": mem()}
            w = st["window"]
            if w and "t1" in w:
                n = w["last_call"] - w["first_call"]
                out["window"] = {"calls": n, "wall_ms_per_call": (w["t1"] - w["t0"]) / n * 1e3,
                                 "op_gpu_ms_per_call": sum(w["stages_ms_per_call"].values()),
                                 "stages_ms_per_call": w["stages_ms_per_call"]}
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
        except Exception as e:                        # never turn a finished run into a failed one
            print("harness stats not written:", repr(e), file=sys.stderr)

    atexit.register(dump)


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
        if os.environ.get("HGS_HARNESS_STATS"):
            _install_stats(os.environ["HGS_HARNESS_STATS"])
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
