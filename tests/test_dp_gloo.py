"""Per-view data parallelism on CPU: 2 processes, gloo, the same flat-bucket code path
bench.py uses with RCCL.  The per-view renderer here is the CPU oracle (tests may use it); the property
checked is the exchange: all-reduced bucket == sum of the per-view gradients."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mp_util import run_world



def _view_grads(rank, world):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import parity as pa
    from hgs import synth
    torch.set_num_threads(1)      # same float32 summation order in the workers and in the expectation
    W, H = 64, 48
    base = synth.make_camera(W, H)
    scene = synth.make_scene(200, base, seed=0)
    cam = synth.orbit_camera(W, H, rank, world)
    gc, gd = synth.upstream_grads(H, W)
    _, g = pa.run_oracle(scene, cam, torch.zeros(3), gc, gd, dtype=torch.float32)
    return scene, g


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from hgs import dp
    r, _, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    scene, g = _view_grads(rank, world)
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    bucket = dp.GradBucket({k: tuple(g[k].shape) for k in names}, "cpu")
    bucket.fill({k: g[k].float() for k in names})
    assert bucket.flat.numel() == 59 * scene.P
    views = bucket.all_reduce()
    q.put((rank, {k: v.clone() for k, v in views.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucket_allreduce_equals_sum_of_view_grads():
    world = 2
    got = dict(run_world(_worker, world, timeout=240, join_timeout=60))
    expect = None
    for r in range(world):
        _, g = _view_grads(r, world)
        expect = g if expect is None else {k: expect[k] + g[k] for k in g}
    for k in got[0]:
        assert torch.equal(got[0][k], got[1][k])                       # every rank holds the same reduced bucket
        assert torch.allclose(got[0][k], expect[k].float(), rtol=1e-4, atol=1e-5 * float(expect[k].abs().max())), k


def test_view_sharding():
    from hgs import dp
    assert dp.shard_views(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((dp.shard_views(10, r, 4) for r in range(4)), [])) == list(range(10))
