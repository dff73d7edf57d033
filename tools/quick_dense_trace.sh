#!/bin/bash
# kernel statistics of ONE dense design at the headline size (gpurun from the repo root): top kernels by time
#   bash tools/quick_dense_trace.sh TAG  ->  gpurun_out/r6/dense_stats_TAG.csv
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6/dense_trace_$1
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o dense -- python3 tools/gpu_dense_one.py 512 16384 > $OUT.log 2>&1 || exit 1
python3 - "$OUT" "gpurun_out/r6/dense_stats_$1.csv" <<'PY'
import glob, os, sqlite3, sys
from collections import defaultdict
db = glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0]
agg = defaultdict(lambda: [0, 0.0])
for name, s, e in sqlite3.connect(db).cursor().execute("select name, start, end from kernels"):
    k = name.split("(")[0].replace("void mbfir::", "")
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
with open(sys.argv[2], "w") as fh:
    fh.write("kernel,calls,total_us,avg_us,percent   (total %.1f us)\n" % tot)
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        fh.write('"%s",%d,%.1f,%.3f,%.2f\n' % (k, c, t, t / c, 100 * t / tot))
PY
rm -rf $OUT
