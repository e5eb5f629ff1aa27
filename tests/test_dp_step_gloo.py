"""The data-parallel STEP (hgs.dp.DataParallelStep + DensifyStats) on CPU: 2 processes, gloo.  The per-view renderer is
the CPU oracle and the optimizer a CPU stand-in (both test infrastructure: the HIP op and the fused Adam have no CPU
path -- tests/test_dp_step_gpu.py runs the same protocol through them); what is tested here is the protocol of
SURVEY.md section 8(e):
  * the split SUM all-reduce of the gradient bucket,
  * the three densification reductions (MAX of |means2D.grad|, SUM of the visibility count, MAX of the radii:
    /root/reference/scene/gaussian_model.py:687-689, train_single.py:147),
  * ``relevant`` taken from the REDUCED opacity gradient (train_single.py:170-174),
  * every rank ends the step with bit-identical parameters, equal to one process stepping through all views."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mp_util import run_world

HERE = os.path.dirname(os.path.abspath(__file__))



class _Ctx:           # what DataParallelStep needs of a RasterContext when the op itself is not in play
    def __init__(self, grad_buffers=None, backward_stream=None):
        self.grad_buffers, self.backward_stream, self.grad_accumulate = grad_buffers, backward_stream, False

    def wait_backward_stream(self):
        pass


def _run_steps(rank, world, n_steps, views_per_rank, n_views=None, sharded=False):
    sys.path.insert(0, HERE)
    import dp_common as dc
    import parity as pa
    from hgs import dp
    torch.set_num_threads(1)
    n_views = world * views_per_rank if n_views is None else n_views
    scene, cams, targets = dc.scene_and_cams(n_views)
    params = {k: getattr(scene, k).clone() for k in dc.NAMES}
    if sharded:       # reduce-scatter + Adam on this rank's rows + all-gather (hgs.dp.ShardedDataParallelStep)
        step = dp.ShardedDataParallelStep(
            params, lambda sp: dc.OracleAdam([dict(params=[sp[k]], lr=dc.LRS[k]) for k in dc.NAMES]), make_context=_Ctx)
    else:
        opt = dc.OracleAdam([dict(params=[params[k]], lr=dc.LRS[k]) for k in dc.NAMES])
        step = dp.DataParallelStep(params, opt, make_context=_Ctx)
    accum = dict(xyz_gradient_accum=torch.zeros(dc.P, 1), denom=torch.zeros(dc.P, 1), max_radii2D=torch.zeros(dc.P))
    for _ in range(n_steps):
        step.begin()
        for j in dp.shard_views(n_views, rank, world):
            cur = type(scene)(params["means3D"], params["scales"], params["rotations"], params["opacities"],
                              params["shs"], 3)
            oo, g = pa.run_oracle(cur, cams[j], torch.zeros(3), *_upstream(targets[j]), dtype=torch.float32)
            for k in dc.NAMES:                       # what the op's backward does with grad_buffers / grad_accumulate
                buf = step.context.grad_buffers[k]
                buf.copy_(g[k].float().view_as(buf) + (buf if step.context.grad_accumulate else 0))
            step.context.grad_buffers["means2D"].copy_(g["means2D"].float())
            step.view_done(oo.radii)
        step.finish()
        step.stats.apply(accum["xyz_gradient_accum"], accum["denom"], accum["max_radii2D"])
    return {k: v.detach().clone() for k, v in params.items()}, accum


def _upstream(target):
    """Fixed seeded dL/dcolor, dL/dinvdepth per view (a loss would need the forward first; for the protocol test a
    fixed upstream gradient does the same job)."""
    return target[0] - 0.5, target[1] - 0.15


def _worker(rank, world, port, q, n_views=None, sharded=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, HERE)
    from hgs import dp
    dp.init_from_env(backend="gloo")
    params, accum = _run_steps(rank, world, 2, 2, n_views=n_views, sharded=sharded)
    # numpy: pickled by value (a tensor would travel as a shared-memory handle that dies with this process)
    q.put((rank, {k: v.numpy() for k, v in params.items()}, {k: v.numpy() for k, v in accum.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_dp_step_ranks_agree_and_match_one_process():
    world = 2
    got = {}
    for r, params, accum in run_world(_worker, world, timeout=500, join_timeout=60):
        got[r] = ({k: torch.from_numpy(v) for k, v in params.items()}, {k: torch.from_numpy(v) for k, v in accum.items()})
    ref_params, ref_accum = _run_steps(0, 1, 2, 4)          # one process, the same 4 views per step
    sys.path.insert(0, HERE)
    import dp_common as dc
    for k in dc.NAMES:
        assert torch.equal(got[0][0][k], got[1][0][k]), f"{k}: ranks diverged"
        scale = float(ref_params[k].abs().max())
        assert float((got[0][0][k] - ref_params[k]).abs().max()) <= 1e-6 * scale, k
    # rows no view ever saw were left alone by every rank (the row selection uses the REDUCED opacity gradient)
    scene, _, _ = dc.scene_and_cams(4)
    hidden = slice(dc.P // 2, dc.P)
    assert torch.equal(got[0][0]["means3D"][hidden], scene.means3D[hidden])
    assert not torch.equal(got[0][0]["means3D"][:dc.P // 2], scene.means3D[:dc.P // 2])
    for k, v in ref_accum.items():
        assert torch.equal(got[0][1][k], got[1][1][k]) and torch.allclose(got[0][1][k], v, rtol=1e-6, atol=0), k
    assert float(ref_accum["denom"].max()) == 8.0 and float(ref_accum["denom"][hidden].max()) == 0.0


@pytest.mark.timeout(600)
def test_rank_without_a_view_contributes_zeros():
    """Fewer views than ranks (shard_views hands rank 2 nothing): the idle rank still holds the previous step's REDUCED
    gradients in its bucket -- it must contribute zeros, not those (DataParallelStep.finish).  2 views per step on 3
    ranks for 2 steps = one process rendering the same 2 views per step."""
    world, n_views = 3, 2
    got = {}
    for r, params, accum in run_world(_worker, world, extra=(n_views,), timeout=500, join_timeout=60):
        got[r] = ({k: torch.from_numpy(v) for k, v in params.items()}, {k: torch.from_numpy(v) for k, v in accum.items()})
    ref_params, ref_accum = _run_steps(0, 1, 2, n_views)
    sys.path.insert(0, HERE)
    import dp_common as dc
    for k in dc.NAMES:
        for r in (1, 2):
            assert torch.equal(got[0][0][k], got[r][0][k]), f"{k}: ranks diverged"
        scale = float(ref_params[k].abs().max())
        assert float((got[0][0][k] - ref_params[k]).abs().max()) <= 1e-6 * scale, k
    for k, v in ref_accum.items():
        assert torch.equal(got[0][1][k], got[2][1][k]) and torch.allclose(got[0][1][k], v, rtol=1e-6, atol=0), k


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_optimizer_step_equals_the_all_reduce_step(world):
    """reduce-scatter + rank-sharded Adam + all-gather (hgs.dp.ShardedDataParallelStep) against all-reduce + the full
    Adam step on every rank (DataParallelStep): the same parameters after two steps -- BIT FOR BIT at two ranks (a + b
    is the only summation order there is), and at three ranks (400 rows: blocks of 134, the last one padded -- the
    staged all-gather) up to the order in which the two collectives add three numbers.  The ranks of one run always agree
    bit for bit, and so do the densification statistics."""
    runs = {}
    for sharded in (False, True):
        got = {}
        for r, params, accum in run_world(_worker, world, extra=(None, sharded), timeout=800, join_timeout=60):
            got[r] = ({k: torch.from_numpy(v) for k, v in params.items()}, {k: torch.from_numpy(v) for k, v in accum.items()})
        runs[sharded] = got
    sys.path.insert(0, HERE)
    import dp_common as dc
    for k in dc.NAMES:
        for r in range(1, world):
            assert torch.equal(runs[True][0][0][k], runs[True][r][0][k]), f"{k}: sharded ranks diverged"
        a, b = runs[False][0][0][k], runs[True][0][0][k]
        if world == 2:
            assert torch.equal(a, b), f"{k}: sharded step differs from the all-reduce step"
        else:
            assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()), k
    for k in runs[False][0][1]:
        assert torch.equal(runs[False][0][1][k], runs[True][0][1][k]), k
