#!/bin/bash
# kernel statistics of ONE design of BASELINE config 3 (fir_qp_cvx, H-1 dual band, n = 512, m = 16384) alone on the GPU
#   bash tools/quick_c3_trace.sh TAG  ->  gpurun_out/r6/c3_stats_TAG.csv
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6/c3_trace_$1
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o c3 -- python3 tools/gpu_c3_trace.py 512 16384 > $OUT.log 2>&1 || exit 1
python3 - "$OUT" "gpurun_out/r6/c3_stats_$1.csv" <<'PY'
import glob, os, sqlite3, sys
from collections import defaultdict
db = glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0]
agg = defaultdict(lambda: [0, 0.0])
rows = list(sqlite3.connect(db).cursor().execute("select name, start, end from kernels order by start"))
for name, s, e in rows:
    k = name.split("(")[0].replace("void mbfir::", "")
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
span = (rows[-1][2] - rows[0][1]) / 1e3
with open(sys.argv[2], "w") as fh:
    fh.write("kernel,calls,total_us,avg_us,percent   (kernel time %.1f us, first start to last end %.1f us, %d launches)\n" % (tot, span, len(rows)))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        fh.write('"%s",%d,%.1f,%.3f,%.2f\n' % (k, c, t, t / c, 100 * t / tot))
PY
rm -rf $OUT
