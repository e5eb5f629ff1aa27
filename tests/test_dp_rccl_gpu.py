"""RCCL itself on the GPU box: a ONE-rank communicator (two ranks cannot share a device under RCCL) carrying the calls
the data-parallel step makes at N > 1 -- `dist.all_reduce` of the gradient bucket's spans, blocking and `async_op=True`
on the stream the backwards run on (bench.py `_reduce_async`, hgs/dp.py GradBucket.all_reduce_async).  What it pins:
backend "nccl" initialises on this image / driver (HSA_ENABLE_IPC_MODE_LEGACY=0), the collectives run on a slice of the
flat bucket in place, and the stream ordering contract (work.wait() orders the CURRENT stream) holds.  The N > 1
arithmetic is covered by the gloo tests; the xGMI transport stays unmeasured until a multi-GPU node runs bench.py."""
import os
import sys

import pytest
import torch

from mp_util import run_world

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "hierarchical-3d-gaussians_amd"))
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from hgs import dp
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    P = 100_003
    shapes = dict(means3D=(P, 3), shs=(P, 16, 3), opacities=(P, 1), scales=(P, 3), rotations=(P, 4))
    bucket = dp.GradBucket(shapes, dev)
    g = torch.Generator(device=dev).manual_seed(0)
    bucket.flat.copy_(torch.randn(bucket.flat.numel(), device=dev, generator=g))
    expect = bucket.flat.clone()
    # blocking, whole bucket
    dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    ok_sync = bool(torch.equal(bucket.flat, expect))
    # the two groups of the data-parallel step, async on a second stream, ordered after work enqueued on that stream
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        bucket.views["opacities"].mul_(2.0)                       # producer on the side stream
        w1 = dist.all_reduce(bucket.span(dp.DataParallelStep.EARLY), op=dist.ReduceOp.SUM, async_op=True)
        bucket.views["shs"].mul_(3.0)
        w2 = dist.all_reduce(bucket.span(dp.DataParallelStep.LATE), op=dist.ReduceOp.SUM, async_op=True)
        w1.wait(); w2.wait()                                      # orders `side`
        snap = bucket.flat.clone()                                # consumer on the side stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref = dp.GradBucket(shapes, dev)
    ref.flat.copy_(expect)
    ref.views["opacities"].mul_(2.0); ref.views["shs"].mul_(3.0)
    ok_async = bool(torch.equal(snap, ref.flat))
    q.put((ok_sync, ok_async, dist.get_backend(), str(torch.cuda.nccl.version())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_one_rank_bucket_spans_sync_and_async(gpu):
    (ok_sync, ok_async, backend, version), = run_world(_worker, 1, timeout=400, join_timeout=120)
    print("backend", backend, "RCCL", version)
    assert backend == "nccl" and ok_sync and ok_async
