"""Acceptance harness of BASELINE.json's north_star: the reference's three entry scripts -- train_single.py,
train_post.py, render_hierarchy.py -- run UNMODIFIED as ``__main__`` from /root/reference on top of this repository's
drop-in packages (diff_gaussian_rasterization, gaussian_hierarchy._C, simple_knn._C), chained the way
scripts/full_train.py:172-210 chains them (single -> hierarchy file into the chunk's model directory -> post ->
render).

No GPU exists in the build container and /root/reference does not exist on the GPU box, so the scripts run here on the
CPU with the extension layers backed by the oracle (tests/harness/cpu_backends.py -- test infrastructure) and
``"cuda"`` mapped to the CPU (tests/harness/run_reference_script.py).  What this pins is every call signature that
crosses the boundary as the REAL callers make it: GaussianRasterizationSettings / GaussianRasterizer through
render() and render_post(), expand_to_size / get_interpolation_weights with their mixed-device arguments,
load_hierarchy / write_hierarchy (the real host code of libhgs.so, upstream .hier layout), distCUDA2 -- plus the
plyfile / cv2 / torchvision shims of tests/shims and the synthetic COLMAP scene of tests/harness/make_scene.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

REF = os.environ.get("HGS_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_reference = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train_single.py")),
                                     reason="reference checkout not present (GPU box)")


def run_chain(tmp_path, backend, psnr_floor=30.0):
    """train_single.py -> train_post.py -> render_hierarchy.py, unmodified; backend "cpu" (oracle-backed extension
    layers) or "hip" (the real packages on a GPU: tests/test_reference_on_gpu.py)."""
    from PIL import Image

    def _run(script, *args):
        cp = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "harness", "run_reference_script.py"),
                             "--backend", backend, script, *args], capture_output=True, text=True, timeout=900)
        assert cp.returncode == 0, f"{script} failed:\n{cp.stdout[-2000:]}\n{cp.stderr[-4000:]}"
        return cp.stdout

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from harness import make_scene
    scene = str(tmp_path / "scene")
    hier = make_scene.make(scene)
    out = str(tmp_path / "chunk")

    # --- train_single.py (BASELINE config 2's script): 5 iterations, one camera per step ---------------------------
    log = _run("train_single.py", "-s", scene, "--model_path", out, "--iterations", "5", "--disable_viewer", "-r", "1")
    assert "Training complete." in log
    ply = os.path.join(out, "point_cloud", "iteration_5", "point_cloud.ply")
    for f in (ply, os.path.join(out, "exposure.json"), os.path.join(out, "cfg_args"), os.path.join(out, "cameras.json")):
        assert os.path.getsize(f) > 0, f
    sys.path.insert(0, os.path.join(ROOT, "tests", "shims"))
    from plyfile import PlyData
    el = PlyData.read(ply).elements[0]
    assert len(el) == 240 and {"x", "opacity", "f_dc_0", "f_rest_44", "scale_2", "rot_3"} <= set(el.data.dtype.names)

    # --- train_post.py (config 3's script) on the merged 2-chunk hierarchy, placed where full_train.py puts it ------
    hier_in = os.path.join(out, "hierarchy.hier")
    os.replace(hier, hier_in)
    log = _run("train_post.py", "-s", scene, "--model_path", out, "--hierarchy", hier_in, "--iterations", "5",
               "--disable_viewer", "-r", "1")
    assert "Training complete." in log
    hier_opt = hier_in + "_opt"                                      # scene/gaussian_model.py:419-427
    assert os.path.getsize(hier_opt) == os.path.getsize(hier_in)
    sys.path.insert(0, os.path.join(ROOT, "hierarchical-3d-gaussians_amd"))
    from gaussian_hierarchy._C import load_hierarchy
    a, b = load_hierarchy(hier_in), load_hierarchy(hier_opt)
    assert np.array_equal(a[5].numpy(), b[5].numpy()) and np.array_equal(a[6].numpy(), b[6].numpy())   # topology kept
    assert not np.array_equal(a[0].numpy(), b[0].numpy()), "4 optimizer steps must have moved the Gaussians"

    # --- render_hierarchy.py (configs 3 / 5's script) at two granularities --------------------------------------------
    renders = str(tmp_path / "renders")
    _run("render_hierarchy.py", "-s", scene, "--model_path", out, "--hierarchy", hier_opt, "--out_dir", renders,
         "--taus", "0", "6", "-r", "1")
    psnrs = {}
    for tau in ("0.0", "6.0"):
        vals = []
        for k in range(6):
            img = np.asarray(Image.open(os.path.join(renders, f"render_{tau}", f"view_{k:02d}.png")), np.float64) / 255
            gt = np.asarray(Image.open(os.path.join(scene, "images", f"view_{k:02d}.png")), np.float64) / 255
            assert img.shape == gt.shape == (48, 64, 3)
            vals.append(10 * np.log10(1.0 / max(np.mean((img - gt) ** 2), 1e-12)))
        psnrs[tau] = float(np.mean(vals))
    print("PSNR vs the ground-truth renders:", psnrs)
    # tau = 0 draws the leaves = the Gaussians the ground truth was rendered from; a coarser cut can only be worse
    assert psnrs["0.0"] > psnr_floor and psnrs["0.0"] >= psnrs["6.0"] - 0.5
    return psnrs


TAUS = ("0.0", "3.0", "6.0", "15.0")                                    # render_hierarchy.py:129's default sweep


def _psnr_table(renders, scene, n_views):
    """{tau: mean PSNR over the views, vs the ground-truth images} from render_hierarchy.py's PNGs
    (utils/image_utils.py:17-19 on the 8-bit images)."""
    from PIL import Image
    table = {}
    for tau in TAUS:
        vals = []
        for k in range(n_views):
            img = np.asarray(Image.open(os.path.join(renders, f"render_{tau}", f"view_{k:02d}.png")), np.float64) / 255
            gt = np.asarray(Image.open(os.path.join(scene, "images", f"view_{k:02d}.png")), np.float64) / 255
            vals.append(10 * np.log10(1.0 / max(np.mean((img - gt) ** 2), 1e-12)))
        table[tau] = float(np.mean(vals))
    return table


def run_trained_chain(tmp_path, backend, iters_single, iters_post, n_points, W, H, n_views=8):
    """The PSNR half of the metric on something TRAINED: train_single.py (densification on) -> hierarchy built from the
    trained chunk (tests/harness/ply_to_hier.py stands where full_train.py:212-250 runs the C++ creator / merger) ->
    train_post.py -> render_hierarchy.py at tau in {0, 3, 6, 15}, all on ``backend``; then render_hierarchy.py AGAIN on
    the same saved model with the oracle-backed CPU stand-ins.  Returns (PSNR table of the backend's renders, PSNR table
    of the oracle's renders, worst 8-bit pixel difference between the two sets, lines worth logging)."""
    from PIL import Image

    def _run(be, script, *args):
        cp = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "harness", "run_reference_script.py"),
                             "--backend", be, script, *args], capture_output=True, text=True, timeout=3000)
        assert cp.returncode == 0, f"{script} ({be}) failed:\n{cp.stdout[-2000:]}\n{cp.stderr[-4000:]}"
        return cp.stdout

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from harness import make_scene, ply_to_hier
    scene = str(tmp_path / "scene")
    make_scene.make(scene, n_points=n_points, n_views=n_views, W=W, H=H, radius=1.5, look_at_depth=5.5, hier=False)
    out = str(tmp_path / "chunk")
    log = []
    stdout = _run(backend, "train_single.py", "-s", scene, "--model_path", out, "--iterations", str(iters_single),
                  "--disable_viewer", "-r", "1", "--skip_scale_big_gauss")
    assert "Training complete." in stdout
    ply = os.path.join(out, "point_cloud", f"iteration_{iters_single}", "point_cloud.ply")
    hier_in = os.path.join(out, "hierarchy.hier")
    P, N = ply_to_hier.hier_from_ply(ply, hier_in)
    log.append(f"train_single.py: {iters_single} iterations on '{backend}', {n_points} SfM points -> {P} Gaussians; "
               f"hierarchy of {N} nodes (2 chunks merged)")
    stdout = _run(backend, "train_post.py", "-s", scene, "--model_path", out, "--hierarchy", hier_in,
                  "--iterations", str(iters_post), "--disable_viewer", "-r", "1")
    assert "Training complete." in stdout
    hier_opt = hier_in + "_opt"
    tables, dirs = {}, {}
    for be in dict.fromkeys((backend, "cpu")):                        # backend "cpu": one render serves as both
        dirs[be] = str(tmp_path / f"renders_{be}")
        _run(be, "render_hierarchy.py", "-s", scene, "--model_path", out, "--hierarchy", hier_opt, "--out_dir", dirs[be],
             "--taus", *[t[:-2] for t in TAUS], "-r", "1")
        tables[be] = _psnr_table(dirs[be], scene, n_views)
    t_be, t_or, worst = tables[backend], tables["cpu"], 0
    for tau in TAUS:
        for k in range(n_views):
            a, b = (np.asarray(Image.open(os.path.join(dirs[be], f"render_{tau}", f"view_{k:02d}.png")), np.int32)
                    for be in (backend, "cpu"))
            worst = max(worst, int(np.abs(a - b).max()))
    return t_be, t_or, worst, log


@needs_reference
def test_train_single_train_post_render_hierarchy_run_unmodified(tmp_path):
    run_chain(tmp_path, "cpu")


@needs_reference
def test_trained_chain_plumbing(tmp_path):
    """The long chain of tests/test_reference_on_gpu.py::test_trained_psnr_hip_vs_oracle at toy length on the
    oracle-backed stand-ins: only that every stage finds the previous one's files."""
    t_be, t_or, worst, log = run_trained_chain(tmp_path, "cpu", 6, 6, n_points=120, W=64, H=48, n_views=4)
    assert t_be == t_or and worst == 0 and set(t_be) == set(TAUS)
