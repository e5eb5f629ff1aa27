#!/usr/bin/env python
"""Kernel-time summary (the `rocprofv3 --kernel-trace --stats` view) from a rocprofv3 rocpd .db file.
usage: rocprof_summary.py <results.db> [out.txt]
HGS_SKIP_CALLS=n leaves the first n launches of every kernel out (a fresh process runs its first ~20 frames at ramping
clocks: K7 300 - 327 us against 262 us afterwards, scripts/diag/run_k7_series.sh); the header line says so."""
import os
import collections
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = c.execute(f"select k.kernel_name, d.start, d.end, k.arch_vgpr_count, k.sgpr_count, d.group_segment_size, "
                     f"d.workgroup_size_x, d.grid_size_x from {disp} d join {sym} k on d.kernel_id = k.id").fetchall()
    skip = int(os.environ.get("HGS_SKIP_CALLS", "0"))
    rows.sort(key=lambda r: r[1])
    agg = collections.OrderedDict()
    for n, s, e, vg, sg, lds, wg, grid in rows:
        a = agg.setdefault(n, dict(t=[], vgpr=vg, sgpr=sg, lds=lds, wg=wg, grid=grid, seen=0))
        a["seen"] += 1
        if a["seen"] > skip:
            a["t"].append(e - s)
    agg = collections.OrderedDict((k, v) for k, v in agg.items() if v["t"])
    tot = sum(sum(a["t"]) for a in agg.values()) or 1
    lines = ["%-72s %6s %12s %10s %10s %10s %6s %5s %5s %6s %5s %9s" %
             ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%", "vgpr", "sgpr", "lds", "wg", "grid")]
    for n, a in sorted(agg.items(), key=lambda kv: -sum(kv[1]["t"])):
        t = a["t"]
        short = n.replace("_ZN3hgs12_GLOBAL__N_1", "hgs::").split("E15hgs")[0][:72]
        lines.append("%-72s %6d %12.1f %10.1f %10.1f %10.1f %6.2f %5s %5s %6s %5s %9s" %
                     (short, len(t), sum(t) / 1e3, sum(t) / len(t) / 1e3, min(t) / 1e3, max(t) / 1e3,
                      100.0 * sum(t) / tot, a["vgpr"], a["sgpr"], a["lds"], a["wg"], a["grid"]))
    if skip:
        lines.insert(0, f"# the first {skip} launches of every kernel left out (clock ramp of a fresh process)")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
