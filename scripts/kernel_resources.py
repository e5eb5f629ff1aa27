#!/usr/bin/env python
"""Static resource table of every kernel in libhgs.so (no GPU): registers, LDS, scratch, the occupancy they allow on
gfx950, and the instruction mix of the kernel body -- read from the device assembly hipcc emits for each source
(`--cuda-device-only -S` with the Makefile's flags), i.e. from the same compiler run that builds the library.

    python scripts/kernel_resources.py [--md] [source.hip ...]        default: every .hip under csrc/

Occupancy rules used (MI355X_MICROARCH.md / cdna_hip_programming.md): 512 VGPRs per SIMD lane shared by the resident
waves of a SIMD, allocated in blocks of 8 (waves per SIMD = floor(512 / ceil8(vgprs)), at most 8); 160 KB of LDS per
compute unit shared by its resident workgroups; a workgroup's waves spread over the four SIMDs.  The instruction counts
are STATIC (lines of the kernel's body in the assembly), not executed counts: useful to see what a kernel is made of
(double-precision share, transcendental share, LDS and global accesses) and to compare two versions of one kernel."""
import argparse
import collections
import glob
import math
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hierarchical-3d-gaussians_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-Wno-unused-command-line-argument",
         "--cuda-device-only", "-x", "hip", "-S"]
EXTRA = {"render.hip": ["-fno-slp-vectorize"]}          # (per-file flags of csrc/Makefile)
LDS_PER_CU = 160 * 1024


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True,
                             check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def short(dem):
    """hgs::(anonymous namespace)::kernel<...>(args) -> kernel<...>"""
    s = dem.replace("hgs::(anonymous namespace)::", "").replace("hgs::", "")
    s = re.sub(r"^void ", "", s)
    depth, out = 0, []
    for ch in s:                       # cut at the opening parenthesis of the argument list (outside template brackets)
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)


CLASSES = [            # first match wins
    ("valu_f64", re.compile(r"^v_\w*_f64|^v_cvt_f64|^v_cvt_f32_f64")),
    ("valu_trans", re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_")),
    ("valu_pk", re.compile(r"^v_pk_")),
    ("valu_dpp", re.compile(r"^v_\w+_dpp|dpp")),
    ("valu", re.compile(r"^v_")),
    ("lds", re.compile(r"^ds_")),
    ("global_load", re.compile(r"^(global|buffer|flat)_load")),
    ("global_store", re.compile(r"^(global|buffer|flat)_(store|atomic)")),
    ("scratch", re.compile(r"^scratch_")),
    ("salu", re.compile(r"^s_(?!waitcnt|nop|barrier|endpgm|branch|cbranch|sleep|setprio)")),
    ("wait", re.compile(r"^s_(waitcnt|nop|barrier|sleep)")),
    ("branch", re.compile(r"^s_(branch|cbranch)")),
]


def classify(op, line):
    if "dpp" in line and op.startswith("v_"):
        return "valu_dpp"
    for name, rx in CLASSES:
        if rx.search(op):
            return name
    return "other"


def parse(asm):
    """-> {mangled kernel name: {"meta": {...}, "mix": Counter}}"""
    kernels = {}
    # instruction mix: the lines between "<name>:" and its ".Lfunc_end" label
    cur = None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur = m.group(1)
            kernels.setdefault(cur, {"meta": {}, "mix": collections.Counter()})
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is None:
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        kernels[cur]["mix"][classify(op, t)] += 1
    # metadata: the amdhsa.kernels YAML at the end of the file
    meta = asm[asm.find("amdhsa.kernels:"):]
    for block in re.split(r"\n  - ", meta)[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        d = {}
        for key in ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
                    "max_flat_workgroup_size", "vgpr_spill_count", "sgpr_spill_count", "uses_dynamic_stack"):
            mm = re.search(r"\.%s:\s+(\S+)" % key, block)
            if mm:
                d[key] = mm.group(1)
        kernels.setdefault(name.group(1), {"meta": {}, "mix": collections.Counter()})["meta"] = d
    return {k: v for k, v in kernels.items() if v["meta"]}


def occupancy(vgprs, agprs, lds_static, wg_threads, lds_dynamic=0):
    regs = int(math.ceil((vgprs + agprs) / 8.0) * 8) or 8
    by_regs = min(8, 512 // regs)
    waves_per_wg = max(1, (wg_threads + 63) // 64)
    lds = lds_static + lds_dynamic
    wgs_by_lds = LDS_PER_CU // lds if lds else 10 ** 6
    # waves per SIMD the LDS allows: workgroups per CU x waves per workgroup over four SIMDs
    by_lds = wgs_by_lds * waves_per_wg / 4.0
    return by_regs, min(8.0, by_lds)


# dynamic LDS of the kernels that take it at launch (bytes per workgroup at the metric configuration: M = 16, 256 lanes)
DYNAMIC_LDS = {"preprocess_fwd_kernel": 256 * (16 * 3 + 4) * 4, "preprocess_fwd_h48_kernel": 256 * 192,
               "sh_bwd_kernel": 256 * (16 * 3 + 4) * 4,
               # 1080p: two arrays over a band's 1 020 tiles + 8 bytes per instance of a 4 096-instance chunk (staged scatter)
               "tb_scatter_kernel": 2 * 1020 * 4 + 4096 * 8, "tb_scatter_kernel<false>": 1020 * 4, "tb_count_kernel": 1020 * 4}


def collect(sources):
    """One dictionary per kernel of the given sources (file, kernel, vgpr, agpr, sgpr, lds, lds_dyn, scratch, wg,
    waves_regs, waves_lds, insts, mix)."""
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in sources:
            out = os.path.join(tmp, os.path.basename(src) + ".s")
            cmd = [HIPCC] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-I", os.path.join(ROOT, "include"), src, "-o", out]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"{src}: {r.stderr[-2000:]}")
            ks = parse(open(out).read())
            names = demangle(list(ks))
            for mangled, k in sorted(ks.items(), key=lambda kv: names[kv[0]]):
                m = k["meta"]
                name = short(names[mangled])
                base = name.split("<")[0]
                v, a = int(m.get("vgpr_count", 0)), int(m.get("agpr_count", 0))
                lds, wg = int(m.get("group_segment_fixed_size", 0)), int(m.get("max_flat_workgroup_size", 256))
                dyn = DYNAMIC_LDS.get(name, DYNAMIC_LDS.get(base, 0))
                by_regs, by_lds = occupancy(v, a, lds, wg, dyn)
                mix = k["mix"]
                total = sum(mix.values()) or 1
                rows.append(dict(file=os.path.basename(src), kernel=name, vgpr=v, agpr=a, sgpr=int(m.get("sgpr_count", 0)),
                                 lds=lds, lds_dyn=dyn, scratch=int(m.get("private_segment_fixed_size", 0)),
                                 spills=int(m.get("vgpr_spill_count", 0)), wg=wg, waves_regs=by_regs, waves_lds=by_lds,
                                 insts=total, mix=mix))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sources", nargs="*")
    ap.add_argument("--md", action="store_true", help="markdown table")
    args = ap.parse_args()
    rows = collect(args.sources or sorted(glob.glob(os.path.join(CSRC, "*.hip"))))
    hdr = ["file", "kernel", "wg", "VGPR", "SGPR", "LDS B (+dyn)", "scratch B", "waves/SIMD regs", "waves/SIMD LDS",
           "static insts", "f64 %", "trans %", "pk %", "dpp %", "lds %", "ld/st"]
    table = []
    for r in rows:
        mix, t = r["mix"], r["insts"]
        pct = lambda k: f"{100.0 * mix[k] / t:.0f}"
        table.append([r["file"], r["kernel"][:70], r["wg"], r["vgpr"] + r["agpr"], r["sgpr"],
                      f"{r['lds']}" + (f" (+{r['lds_dyn']})" if r["lds_dyn"] else ""), r["scratch"],
                      r["waves_regs"], f"{r['waves_lds']:.1f}" if r["waves_lds"] < 8 else "8", t, pct("valu_f64"),
                      pct("valu_trans"), pct("valu_pk"), pct("valu_dpp"), pct("lds"),
                      f"{mix['global_load']}/{mix['global_store']}"])
    if args.md:
        print("| " + " | ".join(hdr) + " |")
        print("|" + "---|" * len(hdr))
        for row in table:
            print("| " + " | ".join(str(c) for c in row) + " |")
    else:
        w = [max(len(str(x)) for x in col) for col in zip(hdr, *table)]
        for row in [hdr] + table:
            print("  ".join(str(c).ljust(n) for c, n in zip(row, w)))


if __name__ == "__main__":
    main()
