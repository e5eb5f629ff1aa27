#!/usr/bin/env python
"""Derive profiles/pmc_traffic.json (HBM bytes per launch) and profiles/pmc_valu.json (vector-ALU view) from the
per-kernel PMC summary written by scripts/pmc_summary.py, and stamp them with the run they come from.

    python scripts/pmc_summary.py SQ=gpurun_out/pmc_SQ/pmc_results.db ... > profiles/r02_runNN_pmc.json
    python scripts/make_pmc_json.py profiles/r02_runNN_pmc.json r02_runNN [profiles/r02_microbench_valu_issue.txt]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB and gfx950's
FETCH_SIZE counts 128-byte requests as 64 (MI355X_MICROARCH.md, HBM / rocprofv3 section).  That correction holds for
wide streaming reads.  The two compositing kernels read 64-byte records at random places: there FETCH_SIZE is exact
(profiles/r03_microbench_gather_fetch.txt: 255 MiB counted for 256 MiB gathered), so for them
HBM bytes = FETCH_SIZE * 1024 + S / 2 + WRITE_SIZE * 1024 with S = the kernel's STREAMED reads by the byte model (pixel
planes + the tiles' id lists, counted at half by the counter); the uncorrected 2 * FETCH figure is kept as `*_upper`.
The VALU roof (``_peak_ginst_s``) is taken from the issue-rate microbenchmark (scripts/microbench/valu_issue.hip): the
line of K7's instruction mix at 5 waves per SIMD, the occupancy K7 runs at."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGES = {           # bench.py stage -> substrings of the kernels it launches
    "render_bwd": ["render_bwd_quad"], "render_fwd": ["render_fwd_quad"],
    "preprocess_fwd": ["preprocess_fwd"],
    "preprocess_bwd": ["preprocess_bwd", "sh_bwd"], "duplicate_keys": ["duplicate_tiles"],
    "tile_ranges": ["tile_ranges"],
    "tile_depth_sort": ["tile_depth_sort_wave", "tile_depth_sort_medium", "tile_depth_sort_big"],
    "tile_sort": ["tb_count", "tb_colscan", "tb_base", "tb_scatter"],      # (colscan, base: kernels of rounds 2-4)
}
PER_FRAME = {}   # launches per frame of one kernel symbol when it is not 1


def derive(d, N, L, head=None):
    """(traffic, valu) dictionaries from a per-kernel counter summary {kernel name: {counter: mean per launch}}.
    N pixels, L tile instances of the profiled frames (the streamed reads of the two gather kernels)."""
    traffic, valu = dict(head or {}), dict(head or {})
    d = {k: v for k, v in d.items() if isinstance(v, dict)}        # (metadata entries of a summary file are scalars)
    # streamed reads of the gather kernels: K6 reads the id list (4 L); K7 the id list + dL/dcolor, dL/dinvdepth,
    # final_T, n_contrib (24 N)
    streamed = {"render_fwd": 4 * L, "render_bwd": 4 * L + 24 * N}
    for stage, pats in STAGES.items():
        tot = upper = 0.0
        for p in pats:      # template variants of one kernel (accumulate on / off ...) are averaged, not added
            sel = [cs for k, cs in d.items() if isinstance(cs, dict) and p in k and "FETCH_SIZE" in cs]
            if not sel:
                continue
            up = sum((2 * cs["FETCH_SIZE"] + cs.get("WRITE_SIZE", 0.0)) * 1024 for cs in sel) / len(sel)
            if stage in streamed:
                val = sum((cs["FETCH_SIZE"] + cs.get("WRITE_SIZE", 0.0)) * 1024 for cs in sel) / len(sel) + streamed[stage] / 2
            else:
                val = up
            tot += val * PER_FRAME.get(p, 1)
            upper += up * PER_FRAME.get(p, 1)
        if tot:
            traffic[stage] = tot
            if stage in streamed:
                traffic[stage + "_upper"] = upper
    for stage in ("render_bwd", "render_fwd"):
        for k, cs in d.items():
            if isinstance(cs, dict) and STAGES[stage][0] in k and "SQ_INSTS_VALU" in cs:
                w = cs.get("SQ_WAVES", 0.0) or 1.0
                valu[stage] = {"valu_insts_per_launch": cs["SQ_INSTS_VALU"], "salu_insts_per_launch": cs.get("SQ_INSTS_SALU"),
                               "waves": cs.get("SQ_WAVES"),
                               "valu_active_quadcycles_per_wave": cs.get("SQ_ACTIVE_INST_VALU", 0.0) / w,
                               "wave_quadcycles_per_wave": cs.get("SQ_WAVE_CYCLES", 0.0) / w}
    return traffic, valu


def main():
    d = json.load(open(sys.argv[1]))
    run = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(sys.argv[1])
    sys.path.insert(0, ROOT)
    import bench                      # kernel_source_sha(): bench.py only uses a summary collected on the build it runs
    sha = d.get("_src_sha") or bench.kernel_source_sha()
    head = {"_run": run, "_source": os.path.relpath(sys.argv[1], ROOT), "_src_sha": sha}
    # the configuration the PMC passes run (bench.py defaults: 1920x1080, L from the run)
    traffic, valu = derive(d, 1920 * 1080, float(d.get("_L", 2_660_211)), head)
    if len(sys.argv) > 3:
        txt = open(sys.argv[3]).read()
        m = re.search(r"K7 mix.*waves/SIMD=5\s+[\d.]+ us\s+([\d.]+) G wave-inst/s", txt)
        if m:
            valu["_peak_ginst_s"] = float(m.group(1))
            valu["_peak_source"] = (f"{os.path.relpath(sys.argv[3], ROOT)}: measured issue rate of K7's instruction mix "
                                    "(8 pk_fma : 2 exp : 2 rcp : 4 cndmask : 16 fma) at 5 waves per SIMD on this chip")
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    json.dump(valu, open(os.path.join(ROOT, "profiles", "pmc_valu.json"), "w"), indent=1)
    print(json.dumps({"traffic": traffic, "valu": valu}, indent=1))


if __name__ == "__main__":
    main()
