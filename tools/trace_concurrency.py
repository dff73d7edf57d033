"""How many kernels run at once in a rocprofv3 kernel trace (SQLite *_results.db): busy fraction of the span, time-weighted
mean number of kernels in flight, and the same per queue.  Usage: python3 tools/trace_concurrency.py <dir with the db>"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

path = glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0]
cur = sqlite3.connect(path).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute("select start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")))
t0, t1 = rows[0][0], max(r[1] for r in rows)
# keep the last 60 % of the span (the timed steps; the start is warm-up, allocation, seed tables)
lo = t0 + 0.4 * (t1 - t0)
ev = []
perq = defaultdict(float)
for r in rows:
    s, e = max(r[0], lo), r[1]
    if e <= lo:
        continue
    ev.append((s, 1)); ev.append((e, -1))
    if qcol:
        perq[r[2]] += e - s
ev.sort()
busy = 0.0; area = 0.0; act = 0; last = lo
hist = defaultdict(float)
for t, d in ev:
    if act > 0:
        busy += t - last
    area += act * (t - last)
    hist[min(act, 8)] += t - last
    act += d; last = t
span = t1 - lo
print("span %.1f ms, some kernel running %.1f %%, mean kernels in flight %.2f" % (span / 1e6, 100 * busy / span, area / span))
print("time share by number of kernels in flight:", {k: round(100 * v / span, 1) for k, v in sorted(hist.items())})
if qcol:
    print("per %s: share of the span with a kernel of that queue running:" % qcol, {k: round(100 * v / span, 1) for k, v in perq.items()})
