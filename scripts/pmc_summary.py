#!/usr/bin/env python
"""Per-kernel PMC totals from rocprofv3 --pmc rocpd .db files.
usage: pmc_summary.py <name=db> [<name=db> ...]  -> prints {kernel: {counter: mean per dispatch}}"""
import collections
import json
import sqlite3
import sys


def load(db_path):
    db = sqlite3.connect(db_path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))
    pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = (f"select k.kernel_name, i.name, e.value, d.id from {pe} e join {ip} i on e.pmc_id = i.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} k on d.kernel_id = k.id")
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for kname, cname, val, did in db.execute(q):
        per[(kname, did)][cname] += val
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for (kname, did), cs in per.items():
        for c, v in cs.items():
            out[kname][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in out.items()}


def short_name(k):
    """The mangled symbol without the namespace prefix and the argument list (what the committed summaries are keyed by)."""
    return k.replace("_ZN3hgs12_GLOBAL__N_1", "hgs::").split("E15hgs")[0][:60]


def main():
    res = collections.defaultdict(dict)
    for arg in sys.argv[1:]:
        name, path = arg.split("=", 1)
        for k, cs in load(path).items():
            res[short_name(k)].update(cs)
    try:        # the build these counters were collected on (bench.py refuses summaries from another build)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        res["_src_sha"] = bench.kernel_source_sha()
    except Exception:
        pass
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
