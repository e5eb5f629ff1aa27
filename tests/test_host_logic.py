"""Host-side behaviour that needs no GPU: the op surface the reference's glue expects
(gaussian_renderer/__init__.py:44-64,105-113,247-277), loud failure without a GPU, .hier I/O."""
import inspect
import os

import pytest
import torch

import diff_gaussian_rasterization as dgr
from hgs import synth


def test_settings_fields_match_reference_call_sites():
    # keyword set used at gaussian_renderer/__init__.py:44-62 (render) and :247-265 (render_post)
    expected = {"image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                "projmatrix", "sh_degree", "campos", "prefiltered", "debug", "do_depth", "render_indices",
                "parent_indices", "interpolation_weights", "num_node_kids"}
    assert set(dgr.GaussianRasterizationSettings._fields) == expected
    sig = inspect.signature(dgr.GaussianRasterizer.forward)
    assert list(sig.parameters)[1:] == ["means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                        "rotations", "cov3D_precomp"]
    assert hasattr(dgr, "_C")                                   # imported at gaussian_renderer/__init__.py:17


def test_op_fails_loudly_without_gpu_and_never_falls_back():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cam = synth.make_camera(32, 32)
    e_i, e_f = torch.empty(0, dtype=torch.int32), torch.empty(0)
    rs = dgr.GaussianRasterizationSettings(
        image_height=32, image_width=32, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3),
        scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3,
        campos=cam.camera_center, prefiltered=False, debug=False, do_depth=True, render_indices=e_i,
        parent_indices=e_i, interpolation_weights=e_f, num_node_kids=e_i)
    sc = synth.make_scene(8, cam)
    with pytest.raises(RuntimeError, match="no CPU path"):
        dgr.GaussianRasterizer(rs)(means3D=sc.means3D, means2D=torch.zeros(8, 3), shs=sc.shs, opacities=sc.opacities,
                                   scales=sc.scales, rotations=sc.rotations)
    import simple_knn._C as knn
    with pytest.raises(RuntimeError, match="no CPU path"):
        knn.distCUDA2(torch.zeros(4, 3))


def test_product_code_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "hierarchical-3d-gaussians_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(dirpath, f)


def test_hier_roundtrip(tmp_path):
    from gaussian_hierarchy._C import load_hierarchy, write_hierarchy
    from hgs import hierarchy
    cam = synth.make_camera(64, 64)
    sc = synth.make_scene(37, cam, seed=4)
    h = hierarchy.build_hierarchy(sc)
    path = str(tmp_path / "toy.hier")
    write_hierarchy(path, h.xyz, h.shs, h.alpha, h.log_scales, h.rots, h.nodes, h.boxes)
    xyz, shs, alpha, ls, rots, nodes, boxes = load_hierarchy(path)
    assert shs.shape == (h.xyz.shape[0], 16, 3) and alpha.shape == (h.xyz.shape[0], 1)
    assert nodes.dtype == torch.int32 and nodes.shape[1] == 7 and boxes.shape[1:] == (2, 4)
    for a, b in ((xyz, h.xyz), (shs, h.shs), (alpha, h.alpha), (ls, h.log_scales), (rots, h.rots),
                 (nodes, h.nodes), (boxes, h.boxes)):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        load_hierarchy(str(tmp_path / "missing.hier"))
    bad = tmp_path / "bad.hier"
    bad.write_bytes(b"NOTAHIER" + b"\0" * 64)
    with pytest.raises(RuntimeError, match="neither an upstream .hier file"):
        load_hierarchy(str(bad))
    trunc = tmp_path / "trunc.hier"
    trunc.write_bytes(open(path, "rb").read()[:200])
    with pytest.raises(RuntimeError, match="neither an upstream .hier file"):
        load_hierarchy(str(trunc))


def test_hier_upstream_layout_bytes(tmp_path):
    """The default on-disk layout is the header-less one of the upstream gaussian-hierarchy tools (int P, pos, rot,
    log-scale, alpha, sh[16][3], int N, nodes[7], boxes[2][4]) -- checked byte for byte against a file assembled here
    with numpy, in both directions; the half-precision variant (P < 0) is read; other SH counts use the private
    layout."""
    import numpy as np
    from gaussian_hierarchy._C import load_hierarchy, write_hierarchy
    rng = np.random.default_rng(0)
    P, N = 5, 3
    pos, rot = rng.normal(size=(P, 3)).astype("<f4"), rng.normal(size=(P, 4)).astype("<f4")
    ls, al = rng.normal(size=(P, 3)).astype("<f4"), rng.random((P, 1)).astype("<f4")
    sh = rng.normal(size=(P, 16, 3)).astype("<f4")
    nodes = rng.integers(-1, 9, size=(N, 7)).astype("<i4")
    boxes = rng.normal(size=(N, 2, 4)).astype("<f4")
    blob = b"".join([np.int32(P).tobytes(), pos.tobytes(), rot.tobytes(), ls.tobytes(), al.tobytes(), sh.tobytes(),
                     np.int32(N).tobytes(), nodes.tobytes(), boxes.tobytes()])
    up = tmp_path / "upstream.hier"
    up.write_bytes(blob)
    got = load_hierarchy(str(up))
    for a, b in zip(got, (pos, sh, al, ls, rot, nodes, boxes)):
        assert np.array_equal(a.numpy(), b)
    mine = tmp_path / "mine.hier"
    write_hierarchy(str(mine), *(torch.from_numpy(np.ascontiguousarray(t)) for t in (pos, sh, al, ls, rot, nodes, boxes)))
    assert mine.read_bytes() == blob
    # half-precision variant: P stored negative, everything but the positions as IEEE half
    h16 = lambda t: t.astype("<f2").tobytes()
    half = tmp_path / "half.hier"
    half.write_bytes(b"".join([np.int32(-P).tobytes(), pos.tobytes(), h16(rot), h16(ls), h16(al), h16(sh),
                               np.int32(N).tobytes(), nodes.tobytes(), boxes.tobytes()]))
    got = load_hierarchy(str(half))
    assert np.array_equal(got[0].numpy(), pos) and np.array_equal(got[5].numpy(), nodes)
    for a, b in zip((got[1], got[2], got[3], got[4]), (sh, al, ls, rot)):
        assert np.array_equal(a.numpy(), b.astype("<f2").astype("<f4"))
    # a size that does not add up is rejected (there is no magic to go by)
    (tmp_path / "odd.hier").write_bytes(blob + b"\0")
    with pytest.raises(RuntimeError, match="declared sizes do not match"):
        load_hierarchy(str(tmp_path / "odd.hier"))
    # 4 SH coefficients (a degree-1 model): private layout, still round-trips
    sh4 = torch.from_numpy(np.ascontiguousarray(sh[:, :4]))
    p4 = tmp_path / "m4.hier"
    write_hierarchy(str(p4), torch.from_numpy(pos), sh4, torch.from_numpy(al), torch.from_numpy(ls),
                    torch.from_numpy(rot), torch.from_numpy(nodes), torch.from_numpy(boxes))
    assert p4.read_bytes()[:8] == b"HGSHIER1"
    assert torch.equal(load_hierarchy(str(p4))[1], sh4)


def test_opt_in_fast_paths_fail_loudly_on_cpu_tensors():
    """The opt-in entry points (raw-parameter path, batched SH colours, fused Adam) have no CPU path either."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cam = synth.make_camera(32, 32)
    sc = synth.make_scene(8, cam)
    with pytest.raises(RuntimeError, match="no CPU path"):
        dgr.sh_colors_batched(sc.means3D, sc.shs, 3, [cam.camera_center])
    from hgs.optim import Adam
    p = torch.nn.Parameter(torch.zeros(4, 3))
    p.grad = torch.ones(4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Adam([p], lr=1e-3).step(None)
    with pytest.raises(NotImplementedError):
        Adam([p], lr=1e-3, amsgrad=True)
    e_i, e_f = torch.empty(0, dtype=torch.int32), torch.empty(0)
    rs = dgr.GaussianRasterizationSettings(
        image_height=32, image_width=32, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3),
        scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3,
        campos=cam.camera_center, prefiltered=False, debug=False, do_depth=True, render_indices=e_i,
        parent_indices=e_i, interpolation_weights=e_f, num_node_kids=e_i)
    with pytest.raises(RuntimeError, match="no CPU path"):
        dgr.GaussianRasterizer(rs).forward_raw(sc.means3D, torch.zeros(8, 3), sc.shs[:, :1].contiguous(),
                                               sc.shs[:, 1:].contiguous(), sc.opacities, sc.scales, sc.rotations)
    with pytest.raises(RuntimeError, match="unknown opacity_activation"):
        dgr.GaussianRasterizer(rs).forward_raw(sc.means3D, torch.zeros(8, 3), sc.shs[:, :1].contiguous(),
                                               sc.shs[:, 1:].contiguous(), sc.opacities, sc.scales, sc.rotations,
                                               opacity_activation="tanh")


def test_training_loop_extras_live_on_a_context_object():
    """grad_buffers / backward_stream / deferred SH backward are per RasterContext, never process-global state on the
    autograd class (two models or a viewer render in one process must not share them)."""
    cls = dgr._RasterizeGaussians
    for name in ("grad_buffers", "grad_accumulate", "defer_sh_backward", "pending_sh", "backward_stream", "variant"):
        assert not hasattr(cls, name), name
    a, b = dgr.RasterContext(), dgr.RasterContext(defer_sh_backward=True)
    assert a.grad_buffers is None and a.backward_stream is None and a.defer_sh_backward is False
    assert a.grad_accumulate is False and a.pending_sh == [] and a.pending_sh is not b.pending_sh
    a.wait_backward_stream()              # no-op without a backward stream
    a.finish_deferred_sh_backward()       # no-op without pending views
    rs = dgr.GaussianRasterizationSettings(*([None] * 16))
    assert dgr.GaussianRasterizer(rs).context is None and dgr.GaussianRasterizer(rs, context=b).context is b


def test_bench_byte_model_is_consistent():
    """bench.py's algorithmic byte model: every stage positive, the batched-SH variants move strictly fewer bytes
    per view than the per-view model, and the headline figure is reproduced (1.25 GB-scale per frame at 1 M / 1080p)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    P, L, N, T, M = 1_000_000, 2_665_270, 1920 * 1080, 8160, 16
    base = bench.algorithmic_bytes(P, P, L, N, T, M)
    assert all(v >= 0 for v in base.values()) and 1.5e9 < sum(base.values()) < 2.0e9
    k = 8
    per_step = ("sh_bwd_batched", "sh_colors_batched")
    frame = lambda ab: sum(v for n, v in ab.items() if n not in per_step) + sum(ab.get(n, 0) for n in per_step) / k
    deferred = bench.algorithmic_bytes(P, P, L, N, T, M, k=k, deferred_sh=True)
    both = bench.algorithmic_bytes(P, P, L, N, T, M, k=k, deferred_sh=True, sh_forward=True)
    assert frame(both) < frame(deferred) < frame(base)
    assert both["preprocess_fwd"] < base["preprocess_fwd"] and "sh_colors_batched" in both
    sb = bench.survey_bytes(P, P, L, N, T, M)
    assert sb["preprocess_fwd"] == 44 * P + 12 * M * P + 4 * P + 40 * P


def test_source_stamp_ignores_comments_but_not_code():
    """bench.kernel_source_sha(): the stamp that ties committed PMC summaries to a build hashes the sources without
    comments and blank lines -- rewording a comment is not another build, touching code is."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        bench = importlib.import_module("bench")
    finally:
        sys.argv = argv
    a = "int f(int x) {\n  // doubles x\n  return 2 * x;   /* really */\n}\n\n#define URL \"http://a//b\"\n"
    b = "int f(int x) {\n  return 2 * x;\n}\n#define URL \"http://a//b\"\n"
    c = "int f(int x) {\n  return 3 * x;\n}\n#define URL \"http://a//b\"\n"
    assert bench._code_only(a) == bench._code_only(b) != bench._code_only(c)
    assert "http://a//b" in bench._code_only(a)                  # '//' behind a ':' is not a comment
    assert len(bench.kernel_source_sha()) == 16


def test_forward_arena_layout_against_the_real_size_functions(monkeypatch):
    """The Python glue carves ONE arena per forward (geometry, per-pixel state, binning workspace, backward scratch) and
    hands raw addresses to the C ABI.  No GPU here: the compute entry points are replaced by fakes that check every
    address range the glue passes against the REAL hgs_raster_ws_sizes (ranges disjoint, each at least as large as the
    library asks for), on the two-stage path, the speculative path, a capacity miss and an empty scene;
    hgs_raster_views_get -- pure pointer arithmetic -- runs for real and must hand back correctly shaped views."""
    import ctypes as C
    import torch
    import diff_gaussian_rasterization as dgr
    from hgs import _lib
    Cm = dgr._C
    real = _lib.lib()

    class Fake:
        def __init__(self):
            self.calls, self.L_next = [], 5000

        def __getattr__(self, name):
            return getattr(real, name)

        def _check(self, a, L_ws, **ptrs):
            g, b, i, w = (C.c_size_t() for _ in range(4))
            assert real.hgs_raster_ws_sizes(a.P, a.width, a.height, L_ws, C.byref(g), C.byref(b), C.byref(i), C.byref(w)) == 0
            need = dict(geom=g.value, bin=b.value, img=i.value, bwd=w.value)
            spans = sorted((p, p + need[k], k) for k, p in ptrs.items())
            for (a0, a1, ka), (b0, b1, kb) in zip(spans, spans[1:]):
                assert a1 <= b0, f"{ka} [{a0}, {a1}) overlaps {kb} [{b0}, {b1})"

        def hgs_raster_fwd(self, a, geom, binb, img, L_cap, radii, color, invd, Lref, stream, dev):
            self._check(a._obj, L_cap, geom=geom, bin=binb, img=img)
            self.calls.append(("fwd", L_cap))
            C.cast(Lref, C.POINTER(C.c_uint32))[0] = self.L_next
            return _lib.ERR_CAPACITY if self.L_next > L_cap else 0

        def hgs_raster_fwd_stage1(self, a, geom, radii, Lref, stream, dev):
            C.cast(Lref, C.POINTER(C.c_uint32))[0] = self.L_next
            self.calls.append(("stage1",))
            return 0

        def hgs_raster_fwd_stage2(self, a, geom, binb, img, L, color, invd, stream, dev):
            self._check(a._obj, L, geom=geom, bin=binb, img=img)
            self.calls.append(("stage2", L))
            return 0

        def hgs_raster_bwd(self, a, geom, binb, img, bwd, L, *rest):
            self._check(a._obj, L, geom=geom, bin=binb, img=img, bwd=bwd)
            self.calls.append(("bwd", L))
            return 0

    fake = Fake()
    monkeypatch.setattr(_lib, "lib", lambda: fake)
    monkeypatch.setattr(Cm, "_require_gpu", lambda t, n: t.contiguous())
    monkeypatch.setattr(Cm, "_small", lambda t, n, k: t.to(torch.float32).contiguous())
    monkeypatch.setattr(Cm, "_stream", lambda d: None)
    monkeypatch.setattr(Cm, "_last_L", {})
    W, H = 208, 144
    z = lambda *s: torch.zeros(*s)

    def fwd(P, prepare):
        return Cm.rasterize_gaussians(z(3), z(P, 3), None, z(P, 1), z(P, 3), z(P, 4), 1.0, None, z(4, 4), z(4, 4), 1.0,
                                      1.0, H, W, z(P, 16, 3), 3, z(3), False, False, None, None, None, None, True,
                                      prepare_backward=prepare)[-1]

    def bwd(call):
        Cm.rasterize_gaussians_backward(call, z(3, H, W), z(1, H, W), z(3, H, W), z(1, H, W))

    P = 3000
    call = fwd(P, True)                                      # first view of a shape: two stages, exact sizes
    assert [c[0] for c in fake.calls] == ["stage1", "stage2"] and call.L_ws == call.L == 5000
    v = Cm.raster_views(call)
    assert v["final_T"].shape == (H, W) and v["records"].shape == (P, 16) and v["point_list"].shape == (5000,)
    assert call.scratch is not None and len(call.bufs) == 2
    fake.calls.clear()
    call = fwd(P, True)                                      # speculative: one arena, the backward's scratch inside it
    assert [c[0] for c in fake.calls] == ["fwd"] and call.L_ws >= int(5000 * Cm.SPEC_GROWTH) + Cm.SPEC_SLACK
    assert len(call.bufs) == 1 and call.p_bwd != 0 and Cm.raster_views(call)["n_contrib"].shape == (H, W)
    bwd(call)
    assert fake.calls[-1] == ("bwd", call.L_ws)
    first_plan = call.L_ws
    call = fwd(P, True)
    assert call.L_ws == first_plan                           # the quantised capacity repeats: a plan-cache hit
    fake.calls.clear()
    call = fwd(P + 700, True)                                # another row count at this resolution (train_post.py: a new
    assert [c[0] for c in fake.calls] == ["fwd"]             # cut every iteration): capacity scaled from the last frame
    assert call.L_ws >= int(5000 * (P + 700) / P * Cm.SPEC_GROWTH)
    bwd(call)
    fake.calls.clear()
    fake.L_next = 10 ** 6                                    # the scene grew: capacity miss, exact second stage
    call = fwd(P, False)
    assert [c[0] for c in fake.calls] == ["fwd", "stage2"] and call.L_ws == 10 ** 6 and call.p_bwd == 0
    bwd(call)                                                # a backward nobody announced allocates its scratch itself
    assert call.p_bwd != 0 and fake.calls[-1] == ("bwd", 10 ** 6)
    # the remembered count of a shape is a DECAYING MAXIMUM (train_post.py: a cut that shrank must not make the next, larger
    # one overflow): after the 10^6 frame, a small one, then a large one again -- no capacity miss
    fake.calls.clear()
    fake.L_next = 4000
    fwd(P, False)
    fake.L_next = int(0.9e6)
    call = fwd(P, False)
    assert [c[0] for c in fake.calls] == ["fwd", "fwd"] and call.L_ws >= int(0.9e6)
    # workspace arenas come in size classes: 2^(k/4) above 1 MB, never smaller than asked, at most 19 % larger
    for n in (1, 4097, 1 << 20, (1 << 20) + 1, 123_456_789, 700_000_001, 1 << 30, (1 << 30) + 1):
        c = Cm.size_class(n)
        assert c >= n and (c == n if n <= 1 << 20 else c <= 1.19 * n + 512) and Cm.size_class(c) == c, (n, c)
    assert len({Cm.size_class(int(7e8 * 1.01 ** i)) for i in range(30)}) <= 3     # a drifting request asks for few sizes
    assert all(t.numel() >= 1 for t in call.bufs) and call.bufs[0].numel() == Cm.size_class(call.bufs[0].numel())
    fake.calls.clear()
    fake.L_next = 0
    monkeypatch.setattr(Cm, "_last_L", {})
    call = fwd(0, False)                                     # empty scene
    assert [c[0] for c in fake.calls] == ["stage1", "stage2"] and call.L == 0


def test_pmc_counters_to_bytes_per_launch():
    """scripts/make_pmc_json.derive: FETCH_SIZE / WRITE_SIZE (KiB per launch) -> HBM bytes per launch by the rules of
    MI355X_MICROARCH.md (2 x FETCH for streaming kernels on gfx950; FETCH + half of the streamed reads for the two
    gather kernels), keyed by bench.py's stage names.  The kernel names are the MANGLED symbols rocprofv3 reports,
    shortened by pmc_summary.short_name -- a summary whose names start with '_' once lost every kernel to the filter
    meant for a summary file's metadata entries."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    import make_pmc_json
    import pmc_summary
    mangled = {
        "_ZN3hgs12_GLOBAL__N_122render_bwd_quad_kernelILb1EEEv15hgs_raster_argsNS_6GeomWsE": dict(FETCH_SIZE=1000.0, WRITE_SIZE=100.0, SQ_INSTS_VALU=5e6, SQ_WAVES=100.0, SQ_ACTIVE_INST_VALU=4e5, SQ_WAVE_CYCLES=8e5),
        "_ZN3hgs12_GLOBAL__N_121preprocess_fwd_kernelILb1ELb0ELb1EEEv15hgs_raster_argsNS_6GeomWsEPi": dict(FETCH_SIZE=2000.0, WRITE_SIZE=500.0),
        "_ZN3hgs12_GLOBAL__N_115tb_count_kernelEPKjjS2_iijiPjS3_i": dict(FETCH_SIZE=10.0, WRITE_SIZE=1.0),
        "_ZN3hgs12_GLOBAL__N_117tb_scatter_kernelEPKjS2_jS2_iijiiiS2_S2_S2_PjS3_S3_S3_": dict(FETCH_SIZE=20.0, WRITE_SIZE=30.0),
        "_ZN2at6native29vectorized_elementwise_kernelILi4E": dict(FETCH_SIZE=7.0, WRITE_SIZE=7.0),
    }
    d = {pmc_summary.short_name(k): v for k, v in mangled.items()}
    assert all(not k.startswith("_ZN3hgs") for k in d if "hgs" in k)
    d["_src_sha"] = "0123456789abcdef"                       # metadata of a summary file: ignored
    N, L = 1000, 200
    traffic, valu = make_pmc_json.derive(d, N, L, {"_run": "t"})
    assert traffic["_run"] == "t" and valu["_run"] == "t"
    assert traffic["preprocess_fwd"] == (2 * 2000.0 + 500.0) * 1024                    # streaming: 2 x FETCH + WRITE
    assert traffic["tile_sort"] == (2 * 10.0 + 1.0 + 2 * 20.0 + 30.0) * 1024           # the stage's kernels added up
    assert traffic["render_bwd"] == (1000.0 + 100.0) * 1024 + (4 * L + 24 * N) / 2     # gather: FETCH exact + streamed / 2
    assert traffic["render_bwd_upper"] == (2 * 1000.0 + 100.0) * 1024
    assert "render_fwd" not in traffic and "preprocess_bwd" not in traffic             # no launch seen: no entry
    assert valu["render_bwd"]["valu_insts_per_launch"] == 5e6 and valu["render_bwd"]["valu_active_quadcycles_per_wave"] == 4e3
    # the committed summaries were made by the same function from the committed per-kernel file
    import json
    src = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    per_kernel = json.load(open(os.path.join(root, src["_source"])))
    again, _ = make_pmc_json.derive(per_kernel, 1920 * 1080, float(per_kernel.get("_L", 2_660_211)))
    for k, v in again.items():
        assert src[k] == v, k


def test_live_pmc_collection_declines_under_a_profiler(monkeypatch):
    """bench.py's own counter passes (rocprofv3 child processes) are not started from a process that already runs under
    rocprofv3 -- the driver's profile of the test suite, scripts/r05_final.sh: the reason goes into the line instead."""
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    t, v, why = bench.measure_pmc_live(types.SimpleNamespace(gaussians=1000, width=64, height=64), 64 * 64, 100)
    assert t is None and v is None and "under a profiler" in why


def test_k3_share_partition_covers_every_excess_slot_exactly_once():
    """The arithmetic of K3's rare path (csrc/binning.hip, duplicate_tiles_banded_kernel), restated: heavy blocks = raw
    workgroup sums above thr = max(4096, 4 x the mean), in block order, at most `share_max` of them; a listed block's own
    workgroup emits slots [0, thr), the listed blocks' excesses laid end to end are dealt out in contiguous shares of
    q = max(1024, ceil(E / nblk)) by blockIdx.  Every slot of every block must be emitted exactly once whatever the
    sums and whatever share_max."""
    import numpy as np
    rng = np.random.default_rng(3)
    for nblk, share_max in ((47, 256), (47, 1), (1467, 256), (300, 7), (5, 256)):
        sums = rng.integers(0, 1500, nblk).astype(np.int64)
        heavy = rng.choice(nblk, size=min(nblk, 12), replace=False)
        sums[heavy] = rng.integers(20_000, 300_000, len(heavy))
        thr = max(4096, 4 * -(-int(sums.sum()) // nblk))
        listed = [b for b in range(nblk) if sums[b] > thr][:share_max]
        pe = np.concatenate([[0], np.cumsum([sums[b] - thr for b in listed])]).astype(np.int64)
        E = int(pe[-1])
        emitted = [np.zeros(int(n), dtype=np.int32) for n in sums]
        q = max(1024, -(-E // nblk))
        for wg in range(nblk):
            emitted[wg][:(thr if wg in listed else sums[wg])] += 1                     # its own block
            lo, hi = wg * q, min(wg * q + q, E)
            for i, b in enumerate(listed):
                if lo >= hi or pe[i + 1] <= lo or pe[i] >= hi:
                    continue
                s0, s1 = thr + max(lo, pe[i]) - pe[i], thr + min(hi, pe[i + 1]) - pe[i]
                emitted[b][s0:s1] += 1
        assert all((e == 1).all() for e in emitted), (nblk, share_max)



def test_k7_slot_assignment_and_batch_cut():
    """The arithmetic of K7's staging lanes (csrc/render.hip, render_bwd_quad_kernel), restated: a lane's first slot = the
    (instance, quadrant) pairs of the lanes below it = the sum of the four ballots' ranks; a batch with more pairs than
    slots keeps the longest run of its BACK-most lanes whose pairs fit and the next batch starts at the first lane left
    out.  Invariants over random hit patterns: kept lanes are a suffix that starts at lane <= 64 - slots / 4, their pairs
    take slots 0 .. n - 1 exactly once in (lane, quadrant) order, a lane's pairs are consecutive, no pair of a kept lane is
    dropped, and walking a whole list in such batches processes every instance exactly once."""
    import numpy as np
    rng = np.random.default_rng(11)
    for slots in (128, 124):
        for trial in range(300):
            p = rng.choice([0.1, 0.35, 0.7, 0.95, 1.0])
            hit = rng.random((64, 4)) < p                                   # lane i, quadrant q
            if trial % 7 == 0:
                hit[:] = True
            masks = [sum(int(hit[i, q]) << i for i in range(64)) for q in range(4)]
            counts = [bin(m).count("1") for m in masks]
            below = lambda m, i: bin(m & ((1 << i) - 1)).count("1")
            k = np.array([[below(masks[q], i) for q in range(4)] for i in range(64)])
            first = 0
            if sum(counts) > slots:
                from_here = sum(counts) - k.sum(1)                          # pairs of lanes >= this one
                keep = from_here <= slots
                first = int(np.argmax(keep))
                assert keep[first:].all() and not keep[:first].any()        # a suffix
                assert first <= 64 - slots // 4
                low = (1 << first) - 1
                d = [bin(m & low).count("1") for m in masks]
                k = k - np.array(d)[None, :]
                assert first == 0 or hit[first - 1:].sum() > slots                # ... the LONGEST one that fits
                hit = hit & (np.arange(64) >= first)[:, None]
            slot0 = k.sum(1)
            taken = []
            for i in range(first, 64):
                s_ = int(slot0[i])
                for q in range(4):
                    if hit[i, q]:
                        taken.append(s_)
                        s_ += 1
            assert taken == list(range(len(taken))) and len(taken) <= slots
            assert len(taken) == int(hit[first:].sum())
    # a whole list: batches back to front from the last contributor, every instance in exactly one batch
    for trial in range(50):
        maxnc = int(rng.integers(1, 700))
        pairs = rng.integers(0, 5, size=maxnc)                               # pairs per instance
        seen = np.zeros(maxnc, int)
        hi = maxnc
        while hi > 0:
            bstart = hi - 64
            lanes = np.arange(64)
            staged = bstart + lanes >= 0
            n = np.where(staged, pairs[np.clip(bstart + lanes, 0, maxnc - 1)], 0)
            first = 0
            if n.sum() > 128:
                suffix = n[::-1].cumsum()[::-1]
                first = int(np.argmax(suffix <= 128))
            seen[bstart + lanes[(lanes >= first) & staged]] += 1
            hi = bstart + first
        assert (seen == 1).all()


def test_bench_gpus_flag_means_ranks():
    """`python bench.py --gpus N` with N > 1 and no torchrun environment launches itself under torch.distributed.run --
    and refuses, with a non-zero exit and no JSON line, a node that shows fewer GPUs than ranks (here: none)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HGS_DP_BACKEND")}
    if __import__("torch").cuda.device_count() >= 2:
        pytest.skip("a multi-GPU node: the refusal cannot be provoked")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "GPU" in r.stderr and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, env=dict(env, WORLD_SIZE="4", RANK="0"), timeout=300)
    assert r.returncode != 0 and "must agree" in r.stderr
