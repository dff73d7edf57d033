"""Random lock-step batches: every seed of tests/test_fuzz_gpu.py's generator gives a base spec and variants that differ in
ripples / spike bound / weights only (mostly the same shape: they share a lock-step unit; where the shape follows from the
ripples the batch front end separates them).  All variants of SEVERAL seeds go through one mbfir_solve_batch call (mixed
shapes: units are formed speculatively and regrouped); every job must equal its single-design solve bit for bit.
    python tools/gpu_fuzz_lockstep.py lo hi [seeds per call] [lanes (0 automatic)] [edges | orders | same] [dd]
With `edges` the variants of a seed also differ in their band edges (scaled towards DC): designs of one order whose grids, row
counts and chunk lists differ share HETEROGENEOUS units (round 4) and must still equal their single solves bit for bit.
With `orders` (round 5) the variants differ in their ORDER as well (n, n + 2, n + 4, ... : the same parity, as the probes of the
reference's min-order searches): unknowns, cone counts and lattice extent per lane, the unit sized to its largest design.
With `dd` (round 5) the extended-precision KKT solve is ON for every designer (opts.ddkkt = 1; fir_qp_cvx's default): units whose
lanes switch to the capacitance form one by one, each with its own strong set."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
import test_fuzz_gpu as F

lo, hi = int(sys.argv[1]), int(sys.argv[2])
per_call = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 0
edges = len(sys.argv) > 5 and sys.argv[5] in ("edges", "orders")       # round 4: the variants also differ in their BAND EDGES (heterogeneous units)
orders = len(sys.argv) > 5 and sys.argv[5] == "orders"     # round 5: ... and in their ORDER
dd = len(sys.argv) > 6 and sys.argv[6] == "dd"             # round 5: the extended-precision solve on (lock-step units take it too)
scales = (1.0, 1.25, 0.8, 1.6, 0.9, 1.1)
fscale = (1.0, 0.97, 0.99, 0.93, 0.985, 0.95)              # edges scaled towards DC: grids, row counts and chunk lists all move
def variants(seed):
    which, args = F.make_case(seed)
    out = []
    for q, s in enumerate(scales):
        a = list(args)
        a[3] = np.asarray(args[3]) * s                      # ripples
        if which == "fir_ap_cvx": a[5] = args[5] * (2.0 - s)      # the spike bound too
        if edges: a[1] = np.asarray(args[1]) * fscale[q]
        if orders: a[0] = args[0] + 2 * q
        out.append((which, tuple(a)))
    return out
bad, t0, njobs, lanes_seen = [], time.time(), 0, {}
ctxs = [mbfir.Context(0) for _ in range(3)]
for s0 in range(lo, hi, per_call):
    jobs, tags = [], []
    for seed in range(s0, min(hi, s0 + per_call)):
        for v, job in enumerate(variants(seed)):
            jobs.append(job); tags.append((seed, v))
    # qp: the extended-precision solve runs one design at a time; without it the units form
    opts = mbfir.make_opts(ddkkt=1 if dd else -1, lanes=lanes)
    try:
        res = mbfir.solve_batch(jobs, ctxs=ctxs, info=True, opts=opts)
    except Exception as e:
        bad.append((s0, "batch raised %r" % (e,))); continue
    for (seed, v), job, (h, status, info) in zip(tags, jobs, res):
        njobs += 1
        lanes_seen[info["lanes"]] = lanes_seen.get(info["lanes"], 0) + 1
        h1, s1, i1 = getattr(mbfir, job[0])(*job[1], ctx=ctxs[0], info=True, opts=opts)
        if s1 != status or i1["iters"] != info["iters"] or i1["dd_iters"] != info["dd_iters"] or (status == "Solved" and (not np.array_equal(h, h1) or info["pcost"] != i1["pcost"])):
            bad.append((seed, v, job[0], "batch %s %d it %.17g / single %s %d it %.17g, lanes %d" % (status, info["iters"], info["pcost"], s1, i1["iters"], i1["pcost"], info["lanes"])))
    print("seeds %d..%d: %d jobs so far, %d mismatches, %.0f s" % (lo, min(hi, s0 + per_call) - 1, njobs, len(bad), time.time() - t0), flush=True)
print("lock-step fuzz, seeds %d..%d: %d jobs, %d mismatches; unit sizes seen %s" % (lo, hi - 1, njobs, len(bad), sorted(lanes_seen.items())))
for b in bad:
    print("  FAIL", b)
