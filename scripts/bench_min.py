#!/usr/bin/env python
"""Tuning aid (GPU box): run the quick bench N times and print, per stage, the MINIMUM of the per-run averages (boxes and
runs differ by a few per cent; the minimum is the stable number to compare builds by).
    python scripts/bench_min.py [N] [extra bench.py flags...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
best, fps, bat = {}, [], []
for _ in range(n):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras"] + sys.argv[2:],
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    fps.append(d["value"]); bat.append((d.get("batched") or {}).get("value", 0))
    for k, v in (d.get("stages_ms") or {}).items():
        best[k] = min(best.get(k, 1e9), v)
print("dropin frames/s", [round(x, 1) for x in fps], "batched", [round(x, 1) for x in bat])
print("min stage ms", {k: round(v, 4) for k, v in best.items()}, "sum", round(sum(best.values()), 4))
