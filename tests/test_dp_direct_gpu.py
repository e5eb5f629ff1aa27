"""Direct (two-shot, peer-pointer) all-reduce -- include/hgs.h hgs_p2p_*, hgs.dp.DirectAllReduce.

Only one GPU is available to the test, so the ranks SHARE it: every process exports its bucket and flag block through
hipIpc, opens the others' and runs the real protocol (three system-scope flag barriers, reduce of the own shard from
all buckets, gather of the other shards).  What one GPU cannot show is the traffic going over xGMI.
Checked: whole-bucket and sub-range reductions over several epochs give, on every rank, bit-identical results equal
to the float32 sum in rank order; a GradBucket on the direct route gives what the torch.distributed route gives."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mp_util import run_world

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))



def _data(rank, n, epoch):
    g = torch.Generator().manual_seed(1000 * epoch + rank)
    return torch.randn(n, generator=g)


def _worker(rank, world, port, q, extra_env=None):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), **(extra_env or {}))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "hierarchical-3d-gaussians_amd"))
    sys.path.insert(0, os.path.dirname(HERE))
    from hgs import dp
    dp.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    out = {}
    n = 1_000_003                                   # not a multiple of 4, nor of the rank count
    ar = dp.DirectAllReduce(n, dev)
    try:
        for epoch in range(4):
            ar.flat.copy_(_data(rank, n, epoch).to(dev))
            if epoch < 2:
                ar.all_reduce()                     # whole bucket
            else:                                   # two disjoint sub-ranges, the middle is left alone
                ar.all_reduce(0, 400_000)
                ar.all_reduce(600_000, n - 600_000)
            torch.cuda.synchronize()
            ar.check()
            out[f"e{epoch}"] = ar.flat.cpu().numpy().copy()
        # GradBucket: direct route vs torch.distributed route on the same gradients
        shapes = dict(means3D=(1001, 3), shs=(1001, 16, 3), opacities=(1001, 1), scales=(1001, 3), rotations=(1001, 4))
        grads = {k: _data(rank, int(np.prod(s)), 7 + i).view(*s).to(dev) for i, (k, s) in enumerate(shapes.items())}
        for direct in (True, False):
            b = dp.GradBucket(shapes, dev, direct=direct)
            b.fill(grads)
            h1 = b.all_reduce_async(("opacities", "scales", "rotations"))
            h2 = b.all_reduce_async(("means3D", "shs"))
            h1.wait(); h2.wait()
            torch.cuda.synchronize()
            if b.direct is not None:
                b.direct.check()
            out["bucket_direct" if direct else "bucket_dist"] = {k: v.cpu().numpy().copy() for k, v in b.views.items()}
            if b.direct is not None:
                b.direct.close()
    finally:
        ar.close()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,env", [(2, None), (3, None), (8, {"HGS_P2P_VERIFY": "2"}), (2, {"HGS_P2P_FINEGRAINED": "1"})])
def test_direct_all_reduce_between_processes_sharing_the_gpu(gpu, world, env):
    """world 8: the shard arithmetic of a full node (HGS_P2P_MAX_WORLD), with the HGS_P2P_VERIFY self-check armed for
    the GradBucket exchanges; the last case allocates the buckets fine-grained."""
    res = dict(run_world(_worker, world, extra=(env,), timeout=500, join_timeout=120))
    n = 1_000_003
    for epoch in range(4):
        parts = [_data(r, n, epoch).numpy() for r in range(world)]
        want = parts[0].copy()
        for k in range(1, world):
            want = want + parts[k]                 # float32, rank order: what the reducing rank computes
        for r in range(world):
            got = res[r][f"e{epoch}"]
            if epoch < 2:
                assert np.array_equal(got, res[0][f"e{epoch}"])                # bit-identical on every rank
                assert np.array_equal(got, want)
            else:
                assert np.array_equal(got[:400_000], want[:400_000]) and np.array_equal(got[600_000:], want[600_000:])
                assert np.array_equal(got[400_000:600_000], parts[r][400_000:600_000])   # untouched
    for r in range(world):
        for k, v in res[r]["bucket_direct"].items():
            assert np.array_equal(v, res[0]["bucket_direct"][k])
            ref = res[r]["bucket_dist"][k]
            assert np.abs(v - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), k


def _fallback_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HGS_P2P_INJECT_FAILURE="1")
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "hierarchical-3d-gaussians_amd"))
    from hgs import dp
    dp.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    shapes = dict(means3D=(257, 3), opacities=(257, 1))
    b = dp.GradBucket(shapes, dev, direct=True)          # rank 1 cannot export: EVERY rank falls back, nobody hangs
    b.fill({k: torch.full(s, float(rank + 1), device=dev) for k, s in shapes.items()})
    b.all_reduce()
    torch.cuda.synchronize()
    vals = torch.cat([v.reshape(-1) for v in b.views.values()])      # (the padding between the tensors stays 0)
    q.put((rank, b.direct is None, float(vals.min()), float(vals.max())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_direct_route_falls_back_collectively_when_one_rank_cannot_export(gpu):
    world = 3
    res = run_world(_fallback_worker, world, timeout=250, join_timeout=60)
    for rank, fell_back, lo, hi in res:
        assert fell_back and lo == hi == 6.0, (rank, fell_back, lo, hi)
