"""Throughput with P single-stream PROCESSES on one GPU (vs threads in one process): separates host-side
launch contention from device-side limits."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mbfir
    from conftest import c13
    n, m, reps = 512, 16384, int(sys.argv[2])
    f, a, d = c13(n, "duration")
    opts = mbfir.make_opts(grid_m=m)
    mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=opts)
    print("ready", flush=True)
    sys.stdin.readline()
    t0 = time.time()
    for _ in range(reps): mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=opts)
    print("done %.4f" % (time.time() - t0), flush=True)
    sys.exit(0)
for P in (1, 4, 6):
    reps = 4
    procs = [subprocess.Popen([sys.executable, __file__, "worker", str(reps)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(P)]
    for p in procs: assert p.stdout.readline().strip() == "ready"
    t0 = time.time()
    for p in procs: p.stdin.write("go\n"); p.stdin.flush()
    times = [float(p.stdout.readline().split()[1]) for p in procs]
    el = time.time() - t0
    for p in procs: p.wait()
    print("P=%d processes: %d designs in %.3f s -> %.2f designs/s (per-process %.1f ms/design)" % (P, P * reps, el, P * reps / el, max(times) / reps * 1e3), flush=True)
