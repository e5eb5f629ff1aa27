#!/usr/bin/env python
"""Idle time between consecutive kernels of a step, from a rocprofv3 --kernel-trace rocpd .db file: the launch sequence of
one step (the period of the kernel-name sequence), every kernel's duration and the gap in front of it.
usage: rocprof_gaps.py <results.db> [first-kernel substring, default preprocess_fwd]"""
import sqlite3
import sys

import numpy as np

db = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "preprocess_fwd"
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
rows = c.execute(f"select k.kernel_name, d.start, d.end from {disp} d join {sym} k on d.kernel_id = k.id order by d.start").fetchall()
names = [r[0].replace("_ZN3hgs12_GLOBAL__N_1", "hgs::")[:48] for r in rows]
st = np.array([r[1] for r in rows], dtype=np.int64)
en = np.array([r[2] for r in rows], dtype=np.int64)
idx = [i for i, n in enumerate(names) if anchor in n]
steps = [(a, b) for a, b in zip(idx[:-1], idx[1:])]
steps = steps[len(steps) // 2:]                      # the later half: steady state
L = max(set(b - a for a, b in steps), key=[b - a for a, b in steps].count)
steps = [(a, b) for a, b in steps if b - a == L]
dur = np.array([[en[a + k] - st[a + k] for k in range(L)] for a, b in steps]) / 1e3
gap = np.array([[st[a + k] - en[a + k - 1] for k in range(L)] for a, b in steps]) / 1e3
per = np.array([st[b] - st[a] for a, b in steps]) / 1e3
print(f"{len(steps)} steps of {L} launches; step period mean {per.mean():.1f} us; kernels {dur.sum(1).mean():.1f} us; gaps {gap.sum(1).mean():.1f} us")
print("%-50s %9s %9s" % ("kernel", "dur us", "gap before"))
for k in range(L):
    print("%-50s %9.1f %9.1f" % (names[steps[0][0] + k], dur[:, k].mean(), gap[:, k].mean()))
