#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d /tmp/gprof -o r -- python $R/bench.py --steps 80 --warmup 20 --no-cpu-baseline --no-extras --no-stage-timing --no-secondary --no-live-pmc --schedule dropin > /tmp/gaps.log 2>&1; echo "rocprof exit $?"
tail -1 /tmp/gaps.log | cut -c1-200
cd $R
python - <<'PY'
import sqlite3, glob, numpy as np
db = glob.glob('/tmp/gprof/*.db')[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
rows = c.execute(f"select k.kernel_name, d.start, d.end from {disp} d join {sym} k on d.kernel_id = k.id order by d.start").fetchall()
for key in ("render_bwd", "render_fwd", "preprocess_fwd"):
    d = np.array([(e - s) / 1e3 for n, s, e in rows if key in n])
    print(key, len(d), "first 10:", np.round(d[:10], 1).tolist(), " mean of calls 20..: %.1f  min %.1f  max %.1f" % (d[20:].mean(), d[20:].min(), d[20:].max()))
    print("   by camera (call % 8):", [round(float(d[20:][i::8].mean()), 1) for i in range(8)])
PY
