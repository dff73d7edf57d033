"""The heterogeneous batch of bench.py's `heterogeneous_64` leg on its own (for kernel traces): 64 S-RAND specs (mbfir.spec.spec_rand,
seeds in order, those that solve) at n=512, 16384 grid points, lock-step units of 16 on 4 streams.  Prints designs/s."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import mbfir
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 16384)
ctxs = [mbfir.Context(0) for _ in range(4)]
cand = []
for seed in range(160):
    try:
        f, a, d = mbfir.spec.spec_rand(n, seed)
    except ValueError:
        continue
    cand.append(("fir_ap_cvx", (n, list(f), list(a), list(d), 0.1, 1e-3)))
o = mbfir.make_opts(grid_m=m, lanes=16)
keep, tried = [], 0
while len(keep) < 64 and tried < len(cand):
    chunk = cand[tried:tried + 64]; tried += len(chunk)
    keep += [j for j, r in zip(chunk, mbfir.solve_batch(chunk, ctxs=ctxs, opts=o)) if r[1] == "Solved"]
keep = keep[:64]
for _ in range(2):
    t = time.time()
    res = mbfir.solve_batch(keep, ctxs=ctxs, opts=o, info=True)
    dt = time.time() - t
    print("heterogeneous batch, n %d m %d: %d designs (%d shapes, bands %s), units of %s on 4 streams: %.3f s = %.1f designs/s, %.1f iterations per design" % (
        n, m, len(keep), len({(r[2]["n_rows"], r[2]["n_freq"]) for r in res}), sorted({len(j[1][3]) for j in keep}), sorted({r[2]["lanes"] for r in res}),
        dt, len(keep) / dt, sum(r[2]["iters"] for r in res) / len(keep)), flush=True)
