#!/usr/bin/env python
"""Fill the <<PLACEHOLDER>> figures of DESIGN.md / profiles/r06_verdict_response.md from the closing profile set
(profiles/r06_final_*: bench line, kernel stats, the scripts-at-scale log, scale parity).  Idempotent only on a file that
still has its placeholders; run once, after scripts/r06_final.sh's outputs were copied into profiles/.
    python scripts/fill_design.py [--check]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(ROOT, "profiles", n)


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


b = last_json(P("r06_final_bench.json"))
st = b["stages_ms"]
ex = b.get("extra", {})
N, Lm = 1920 * 1080, b["config"].get("instances_per_frame") or b["config"].get("L") or 2_660_211
V = b["config"].get("visible") or 997_473
vals = {}
vals["FPS"] = f"{b['value']:,.0f}".replace(",", " ")
vals["MS"] = f"{b['ms_per_step']:.3f}"
vals["BATCHED"] = f"{(b.get('batched') or {}).get('value', 0):,.0f}".replace(",", " ")
vals["FRAME_FRAC"] = f"{1.25e9 / (b['ms_per_step'] * 1e-3) / 8e12 * 100:.1f}"
k1, k3, k4a, k4b = st["preprocess_fwd"], st["duplicate_keys"], st["tile_sort"], st["tile_depth_sort"]
k6, k7, k8 = st["render_fwd"], st["render_bwd"], st["preprocess_bwd"]
vals.update(K1=f"{k1:.3f}", K3=f"{k3:.3f}", K4A=f"{k4a:.3f}", K4B=f"{k4b:.3f}", K6=f"{k6:.3f}", K7=f"{k7:.3f}", K8=f"{k8:.3f}")
gbs = lambda mb, ms: f"{mb * 1e6 / (ms * 1e-3) / 1e9:,.0f}".replace(",", " ")
vals.update(K1_GBS=gbs(279, k1), K6_GBS=gbs(167, k6), K7_GBS=gbs(207, k7), K8_GBS=gbs(524, k8))
vals["K7_FRAC"] = f"{207e6 / (k7 * 1e-3) / 8e12 * 100:.1f}"
tot = sum(st.values())
vals.update(P_K7=f"{100 * k7 / tot:.0f}", P_K6=f"{100 * k6 / tot:.0f}", P_K8=f"{100 * k8 / tot:.0f}",
            P_BIN=f"{100 * (k3 + k4a + k4b) / tot:.0f}", P_K1=f"{100 * k1 / tot:.0f}")
fx = lambda k, d=0: f"{ex.get(k, {}).get('value', 0):,.{d}f}".replace(",", " ")
vals.update(C2=fx("config2_300k"), HEAVY=fx("heavy_1m"), TL=fx("trained_like_10m"), TC=fx("trained_cut_10m"),
            C3=fx("config3_train_post"), C5=fx("config5_50m_4k_render"), C5B=fx("config5_budgeted_6gb"))
vals["TL_MS"] = f"{ex.get('trained_like_10m', {}).get('ms_per_step', 0):.2f}"
vals["C5_MS"] = f"{ex.get('config5_50m_4k_render', {}).get('ms_per_step', 0):.1f}"
fm = ex.get("config5_budgeted_6gb", {}).get("frame_ms", {})
vals["C5_P50"], vals["C5_P99"] = f"{fm.get('p50', 0):.1f}", f"{fm.get('p99', 0):.1f}"

vb = b.get("valu") or b["roofline"].get("valu") or {}
vals["VALU_GUIDE"] = f"{100 * vb.get('frac_of_guide_peak', 0):.0f}"
vals["VALU_MIX"] = f"{100 * vb.get('frac_of_mix_peak', 0):.0f}"
vals["ALU_BUSY"] = f"{100 * vb.get('alu_busy_frac', 0):.0f}"

# ---- the scripts at scale
log = open(P("r06_final_config2_config3_scripts.log")).read()
ops = [float(x) for x in re.findall(r"of which the op's kernels ([0-9.]+) ms", log)]
stages = re.findall(r"op stages, ms per call: (.*)", log)
misses = re.findall(r"'capacity_misses': (\d+)", log)
allocs = re.findall(r"allocator at exit:\s+allocated \d+ MiB \(peak (\d+)\), reserved (\d+) MiB", log)


def stage_triplet(s):
    d = dict((k.strip(), float(v)) for k, v in (kv.rsplit(" ", 1) for kv in s.split(", ")))
    return " / ".join(f"{d.get(k, 0):.2f}" if k in d else "–" for k in ("tile_depth_sort", "preprocess_bwd", "duplicate_keys"))


for i, tag in enumerate(("TS", "TP", "RH")):
    vals[tag + "_OP"] = f"{ops[i]:.2f}" if i < len(ops) else "?"
    vals[tag + "_ST"] = stage_triplet(stages[i]) if i < len(stages) else "?"
vals["TP_MISS"] = misses[1] if len(misses) > 1 else "?"
for i, tag in enumerate(("TS", "TP")):
    vals[tag + "_AL"] = f"{int(allocs[i][1]) / 1024:.1f} / {int(allocs[i][0]) / 1024:.1f} GB" if i < len(allocs) else "?"
g = re.findall(r"gradients of the trained rows vs the float64 oracle: worst max-rel ([0-9.e+-]+), worst rel-L2 ([0-9.e+-]+), worst "
               r"element-wise figure ([0-9.]+) x the bound \(float32 oracle: ([0-9.]+)\)", log)
fails = len(re.findall(r"GRADIENT PARITY FAILED", log))
if g:
    vals["GRADS"] = (f"all six tensors compared on {len(g) + fails} cases: worst rel-L2 {max(float(x[1]) for x in g):.1e}, worst max-rel "
                     f"{max(float(x[0]) for x in g):.1e}, element-wise up to {max(float(x[2]) for x in g):.0f} x the bound where the "
                     f"float32 oracle is at {max(float(x[3]) for x in g):.0f}" + (f"; {fails} case(s) outside the rule" if fails else ""))
else:
    vals["GRADS"] = "not collected in the closing lease"

# ---- GPU suite
suite = ""
for n in ("r06_final_pytest_gpu_one_process.log", "r06_final_pytest_gpu.log"):
    if os.path.exists(P(n)):
        m = re.findall(r"(\d+) passed(?:, (\d+) skipped)?", open(P(n)).read())
        if m and n.endswith("one_process.log"):
            suite = (f"The whole GPU suite in one process, the driver's command, no reference checkout visible: **{m[-1][0]} passed, "
                     f"{m[-1][1] or 0} skipped** (`profiles/{n}`).")
vals["SUITE"] = suite or "(GPU suite log not collected)"
vals["NGPU"] = (re.search(r"\*\*(\d+ passed, \d+ skipped)\*\*", suite) or [None, "?"])[1]

# ---- scale parity table
rows = ["| case | P | L | longest list | sampled instances | worst max-rel | worst rel-L2 | worst element-wise (x bound) | float32 oracle element-wise |",
        "|---|---|---|---|---|---|---|---|---|"]
if os.path.exists(P("r06_final_scale_parity.jsonl")):
    seen = {}
    for line in open(P("r06_final_scale_parity.jsonl")):
        try:
            d = json.loads(line)
        except ValueError:
            continue
        if isinstance(d, dict) and "stats" in d and "case" in d:
            seen[d["case"]] = d
    for c, d in seen.items():
        ts = [v for v in d["stats"].values() if isinstance(v, dict)]
        f32 = [v.get("mixed", 0) for v in (d.get("float32_oracle_vs_float64") or {}).values()]
        rows.append(f"| {c[:44]} | {d['P']:,} | {d['L']:,} | {d['longest_list']:,} | {d['tile_instances_sampled']:,} | "
                    f"{max(v['maxrel'] for v in ts):.1e} | {max(v['l2'] for v in ts):.1e} | {max(v['mixed'] for v in ts):.2f} | "
                    f"{(max(f32) if f32 else 0):.2f} |".replace(",", " "))
vals["SCALE_TABLE"] = "\n".join(rows)

check = "--check" in sys.argv
for path in (os.path.join(ROOT, "DESIGN.md"), P("r06_verdict_response.md"), os.path.join(ROOT, "README.md")):
    s = open(path).read()
    need = set(re.findall(r"<<([A-Z0-9_]+)>>", s))
    missing = sorted(n for n in need if n not in vals)
    if missing:
        print(f"{path}: no value for {missing}")
    for k, v in vals.items():
        s = s.replace(f"<<{k}>>", v)
    s = s.replace("<<KB>>", f"{len(s.encode()) // 1024}")
    if not check:
        open(path, "w").write(s)
    print(path, "placeholders:", len(need), "bytes:", len(s.encode()))
if check:
    for k in sorted(vals):
        print(f"  {k:12s} {vals[k][:110]}")
