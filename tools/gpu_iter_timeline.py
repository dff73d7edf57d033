"""One iteration of a solve as the GPU saw it: kernel by kernel, start offset, duration and the gap to the kernel before
(reads the rocprofv3 --kernel-trace database of `rocprofv3 --kernel-trace -d <dir> -o t -- python3 tools/gpu_iter_timeline.py run <n> <m> <lanes>`).
  python3 tools/gpu_iter_timeline.py run 512 16384 1        (under rocprofv3: the workload)
  python3 tools/gpu_iter_timeline.py show <dir> [iteration]   (afterwards: the timeline of one iteration, default the 40th k_scaling)"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "run":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import warnings
    warnings.filterwarnings("ignore")
    import mbfir
    from conftest import c13
    n, m, lanes = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    f, a, d = c13(n, "duration")
    o = mbfir.make_opts(grid_m=m, lanes=lanes)
    if lanes == 1:
        for _ in range(2):
            h, s, info = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=o, info=True)
        print(s, info["iters"], info["ms_total"] if "ms_total" in info else "")
    else:
        jobs = [("fir_ap_cvx", (n, f, a, d, 0.1, 1e-3 * (1 + 0.01 * q))) for q in range(lanes)]
        for _ in range(2):
            res = mbfir.solve_batch(jobs, opts=o, info=True)
        print([r[1] for r in res][:4], res[0][2]["iters"])
    sys.exit(0)

import sqlite3
hits = glob.glob(os.path.join(sys.argv[2], "**", "*_results.db"), recursive=True)
cur = sqlite3.connect(hits[0]).cursor()
rows = [(s, e, n.replace("(anonymous namespace)::", "").replace("mbfir::", "").split("(")[0].replace("void ", "")) for n, s, e in
        cur.execute("select name, start, end from kernels order by start")]
marks = [i for i, r in enumerate(rows) if r[2] == "k_scaling"]
which = int(sys.argv[3]) if len(sys.argv) > 3 else min(40, len(marks) - 2)
lo, hi = marks[which], marks[which + 1]
t0 = rows[lo][0]
busy = 0.0
print("iteration %d: %d kernels, %.1f us from the first start to the next iteration's first start" % (which, hi - lo, (rows[hi][0] - t0) / 1e3))
prev_end = t0
agg = {}
for s, e, n in rows[lo:hi]:
    print("  %8.1f  %-28s %7.1f us   gap %5.1f" % ((s - t0) / 1e3, n[:28], (e - s) / 1e3, (s - prev_end) / 1e3))
    busy += (e - s) / 1e3
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    prev_end = e
print("kernel time %.1f us, gaps %.1f us" % (busy, (rows[hi][0] - t0) / 1e3 - busy))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-30s x%-3d %8.1f us" % (n[:30], c, t))
