"""How do the units of the bench's batch share the GPU?  From a rocprofv3 --kernel-trace database of
    rocprofv3 --kernel-trace -d <dir> -o t -- python3 tools/gpu_overlap.py run 512 16384 64 16 4
`python3 tools/gpu_overlap.py show <dir>` takes the LAST batch in the trace and reports
  * the time with 0, 1, 2, ... k_chol_dag launches in flight and with 0, 1, 2, ... kernels of any kind in flight;
  * per kernel type: launches, mean duration while a k_chol_dag of ANOTHER stream runs over more than half of its span and while
    none does, and the mean gap between the end of the stream's previous kernel and its start (dispatch latency of a dependent launch);
  * per stream: time in k_chol_dag, in other kernels, in gaps."""
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "run":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import warnings
    warnings.filterwarnings("ignore")
    import time
    import mbfir
    from gpu_lanes import jobs_for
    n, m, count, lanes, streams = [int(v) for v in sys.argv[2:7]]
    jobs = jobs_for(n, count)
    o = mbfir.make_opts(grid_m=m, lanes=lanes)
    for _ in range(2):
        t = time.time()
        mbfir.solve_batch(jobs, streams=streams, opts=o)
        print("%.1f ms" % (1e3 * (time.time() - t)), flush=True)
    sys.exit(0)

import sqlite3
hits = glob.glob(os.path.join(sys.argv[2], "**", "*_results.db"), recursive=True)
cur = sqlite3.connect(hits[0]).cursor()
rows = [(s, e, n.replace("(anonymous namespace)::", "").replace("mbfir::", "").split("(")[0].replace("void ", ""), st)
        for n, s, e, st in cur.execute("select name, start, end, stream_id from kernels order by start")]
# the last batch: from the last k_zero_lanes burst (arena zeroing at the start of solve_lanes) on
zl = [i for i, r in enumerate(rows) if r[2] == "k_zero_lanes"]
first = zl[-1]
while first > 0 and rows[first][0] - rows[first - 1][0] < 30e6 and (first - 1 in zl or rows[first - 1][0] > rows[zl[-1]][0] - 30e6):
    first -= 1
rows = rows[first:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
print("last batch: %d kernels on %d streams, %.1f ms" % (len(rows), len({r[3] for r in rows}), (t1 - t0) / 1e6))
chol = [r for r in rows if r[2] == "k_chol_dag"]


def coverage(intervals):
    """time with k = 0, 1, 2, ... of the intervals open"""
    ev = sorted([(s, 1) for s, e in intervals] + [(e, -1) for s, e in intervals])
    out, k, last = defaultdict(float), 0, t0
    for t, d in ev:
        out[k] += t - last
        last, k = t, k + d
    out[0] += t1 - last
    return out


def show_cov(title, cov):
    tot = sum(cov.values())
    print(title + ": " + "  ".join("%d: %.1f %%" % (k, 100 * v / tot) for k, v in sorted(cov.items()) if v / tot > 0.002))


show_cov("k_chol_dag launches in flight", coverage([(r[0], r[1]) for r in chol]))
show_cov("kernels of any kind in flight", coverage([(r[0], r[1]) for r in rows]))
show_cov("kernels other than k_chol_dag in flight", coverage([(r[0], r[1]) for r in rows if r[2] != "k_chol_dag"]))

# overlap of every kernel with the factorisations of other streams
import bisect
cs = sorted((c[0], c[1], c[3]) for c in chol)
starts = [c[0] for c in cs]
stat = defaultdict(lambda: [0, 0.0, 0, 0.0, 0.0, 0])       # n_with, t_with, n_without, t_without, gap, n_gap
prev_end = {}
per_stream = defaultdict(lambda: [0.0, 0.0, 0.0])
for s, e, name, st in rows:
    ov = 0
    i = bisect.bisect_left(starts, e)
    for c in cs[max(0, i - 8):i]:
        if c[2] != st:
            ov += max(0, min(e, c[1]) - max(s, c[0]))
    rec = stat[name]
    if ov > 0.5 * (e - s):
        rec[0] += 1; rec[1] += e - s
    else:
        rec[2] += 1; rec[3] += e - s
    if st in prev_end:
        rec[4] += max(0, s - prev_end[st]); rec[5] += 1
        per_stream[st][2] += max(0, s - prev_end[st])
    prev_end[st] = max(e, prev_end.get(st, 0))
    per_stream[st][0 if name == "k_chol_dag" else 1] += e - s
print("%-28s %6s | %6s %9s | %6s %9s | %9s" % ("kernel", "calls", "n", "us (with)", "n", "us (w/o)", "gap us"))
tot_with = tot_without = 0.0
for name, r in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:32]:
    print("%-28s %6d | %6d %9.1f | %6d %9.1f | %9.1f" % (name[:28], r[0] + r[2], r[0], r[1] / max(r[0], 1) / 1e3, r[2], r[3] / max(r[2], 1) / 1e3,
                                                     r[4] / max(r[5], 1) / 1e3))
for st, (a, b, g) in sorted(per_stream.items()):
    print("stream %s: k_chol_dag %.1f ms, other kernels %.1f ms, gaps %.1f ms" % (st, a / 1e6, b / 1e6, g / 1e6))
