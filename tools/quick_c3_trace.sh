#!/bin/bash
# Kernel-time table of a BASELINE config-3 batch (fir_qp_cvx as written, extended-precision KKT solves): bash tools/quick_c3_trace.sh [count] [streams]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r03/c3_trace -o c3 -- python3 tools/gpu_config3_batch.py ${1:-8} ${2:-8} > gpurun_out/r03/c3_trace.log 2>&1 || exit 1
grep "config 3" gpurun_out/r03/c3_trace.log
python3 - <<PY
import sqlite3, glob, re
from collections import defaultdict
db = glob.glob("gpurun_out/r03/c3_trace/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
agg = defaultdict(lambda: [0, 0.0])
t0, t1 = None, None
for name, s, e in cur.execute("select name, start, end from kernels"):
    k = re.sub(r"\(.*", "", name).replace("mbfir::", "")[:48]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
    t0 = s if t0 is None else min(t0, s); t1 = e if t1 is None else max(t1, e)
tot = sum(v[1] for v in agg.values())
print("kernel time %.1f ms over a span of %.1f ms" % (tot / 1e3, (t1 - t0) / 1e6))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-48s calls %6d total_ms %9.1f avg_us %9.1f %5.1f%%" % (k, c, t / 1e3, t / c, 100 * t / tot))
PY
rm -rf gpurun_out/r03/c3_trace
