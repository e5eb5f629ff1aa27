"""Build-time guard of the frame's kernels (no GPU): registers, LDS and scratch as the compiler reports them for gfx950
(scripts/kernel_resources.py reads the device assembly of the Makefile's compiler invocation).  The occupancy of the
compositing and per-Gaussian kernels was chosen by measurement (profiles/r04_kernel_resources.md); a
source change that silently costs a wave per SIMD or starts spilling shows up here before it reaches a GPU box."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hierarchical-3d-gaussians_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "scripts"))

pytestmark = pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("c++filt")),
                                reason="needs hipcc (cross-compiles without a GPU) and c++filt")

# kernel -> (most registers, fewest waves per SIMD the registers AND the LDS must allow)
FRAME = {
    "render_fwd_quad_kernel<true, false>": (80, 6),
    "render_bwd_quad_kernel<true, false>": (128, 4),
    # K1 at M = 16: the SH block straight into LDS by DMA (whole rows, 48 KB per workgroup), round 5; was 144 registers
    "preprocess_fwd_h48_kernel<true>": (96, 3),
    "preprocess_fwd_h48_kernel<false>": (96, 3),
    "preprocess_bwd_kernel<false, false>": (160, 3),
    "sh_bwd_kernel<false, true, false>": (72, 3),
    "duplicate_tiles_banded_kernel": (56, 8),      # + the call into the rare path (heavy blocks shared out)
    "tile_depth_sort_wave_kernel<false>": (72, 7),      # sample sort of the waves' own lists (round 6)
    "tb_count_kernel": (72, 7),             # count + column scan in one launch, 32 table words in flight per lane (round 5)
    "tb_scatter_kernel<true>": (72, 3),     # 16 instances per lane in flight, the chunk staged in LDS by tile (round 5)
}


@pytest.fixture(scope="module")
def rows():
    import kernel_resources
    return {r["kernel"]: r for r in kernel_resources.collect(
        [os.path.join(CSRC, f) for f in ("render.hip", "preprocess.hip", "binning.hip", "tile_bin.hip")])}


def test_frame_kernels_keep_their_occupancy_and_do_not_spill(rows):
    for name, (max_regs, min_waves) in FRAME.items():
        assert name in rows, (name, sorted(rows))
        r = rows[name]
        assert r["scratch"] == 0, f"{name} uses {r['scratch']} bytes of scratch per lane (register spills)"
        assert r["vgpr"] + r["agpr"] <= max_regs, f"{name}: {r['vgpr'] + r['agpr']} registers, at most {max_regs} expected"
        assert min(r["waves_regs"], r["waves_lds"]) >= min_waves, (name, r["waves_regs"], r["waves_lds"])


def test_no_kernel_of_the_drop_in_path_has_a_private_segment(rows):
    """The only kernels of the library that spill are two instantiations of the batched colour route's SH backward
    (profiles/r04_kernel_resources.md); nothing on the drop-in path does."""
    spilling = sorted(k for k, r in rows.items() if r["scratch"])
    # (+ the long-list depth sort since round 6: ~30 values of its 16-keys-per-lane sample sort spilled once per tile at
    # four waves per SIMD -- measured against the 168-register build at three waves, profiles/r06_depth_sort.md)
    assert all(k.startswith(("sh_bwd_batched_color_kernel", "tile_depth_sort_quad_kernel")) for k in spilling), spilling


def test_double_precision_stays_where_conditioning_needs_it(rows):
    """K6 / K7 / K8b and the binning kernels are float32 / integer only; K1 and K8a carry the double chain (recomputed
    forward and chain rule: tests/tools/k8a_float_chain_study.py says why it stays)."""
    for name in ("render_fwd_quad_kernel<true, false>", "render_bwd_quad_kernel<true, false>", "sh_bwd_kernel<false, true, false>",
                 "duplicate_tiles_banded_kernel", "tb_scatter_kernel<true>", "tile_depth_sort_wave_kernel<false>"):
        assert rows[name]["mix"]["valu_f64"] == 0, name
    assert rows["preprocess_bwd_kernel<false, false>"]["mix"]["valu_f64"] > 0
    assert rows["preprocess_fwd_h48_kernel<true>"]["mix"]["valu_f64"] > 0
