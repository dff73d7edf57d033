#!/bin/bash
# The launch sequence of ONE interior-point iteration of a lock-step unit of 16 headline designs (kernel, duration, gap to the
# previous kernel's end), from a rocprofv3 kernel trace:  bash tools/quick_unit_sequence.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/r03/seq_trace -o seq -- python3 tools/gpu_lanes_one.py 512 16384 16 16 1 1 > gpurun_out/r03/seq_trace.log 2>&1 || exit 1
python3 - <<PY
import sqlite3, glob, re
db = glob.glob("gpurun_out/r03/seq_trace/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = sorted((s, e, re.sub(r"\(.*", "", n).replace("mbfir::", "").replace("void ", "")) for n, s, e in cur.execute("select name, start, end from kernels"))
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_chol_dag")]
a, b = idx[40], idx[41]                       # from one factorisation to the next: one iteration
print("one iteration: %d launches, %.1f us from launch to launch of k_chol_dag" % (b - a, (rows[b][0] - rows[a][0]) / 1e3))
prev = rows[a - 1][1]
busy = 0
for s, e, n in rows[a:b]:
    print("%-34s %8.1f us   gap %6.1f us" % (n[:34], (e - s) / 1e3, (s - prev) / 1e3))
    busy += e - s; prev = e
print("kernel time %.1f us" % (busy / 1e3))
PY
rm -rf gpurun_out/r03/seq_trace
