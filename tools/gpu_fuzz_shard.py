"""Random specs through the row-sharded solve in loop-back (tests/test_shard_gpu.py's harness: 2 or 3 contexts on one GPU, the
all-reduce hook sums the ranks' buffers in a fixed order) against the unsharded solve: same verdict, objective to 1e-8, taps to
1e-6, the ranks' taps bit-identical, iteration counts within 4 (20 %).    python tools/gpu_fuzz_shard.py lo hi"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
import test_fuzz_gpu as F
import test_shard_gpu as S
from conftest import relinf

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, t0, counts = [], time.time(), {}
for seed in range(lo, hi):
    which, args = F.make_case(seed)
    size = 2 + seed % 2
    dense = 1 if seed % 7 == 3 else 0
    o0 = mbfir.make_opts(dense_trig=dense, ddkkt=-1)          # (the extended-precision solve does not run row-sharded)
    h0, s0, i0 = getattr(mbfir, which)(*args, info=True, opts=o0)
    res = S._run_sharded(which, args, size, dense=dense)
    counts[(which, s0, size)] = counts.get((which, s0, size), 0) + 1
    errs = [r for r in res if isinstance(r, Exception) or r is None]
    if errs:
        bad.append((seed, which, size, "rank raised %r" % (errs[0],))); continue
    for rank, (h, s, info) in enumerate(res):
        if s != s0:
            bad.append((seed, which, size, "rank %d verdict %s, unsharded %s (rc %d / %d)" % (rank, s, s0, info["rc"], i0["rc"]))); break
        if s0 == "Solved":
            if abs(info["pcost"] - i0["pcost"]) > 1e-8 * max(1.0, abs(i0["pcost"])):
                bad.append((seed, which, size, "objective %.14g vs %.14g" % (info["pcost"], i0["pcost"]))); break
            if not np.array_equal(h, res[0][0]):
                bad.append((seed, which, size, "ranks' taps differ")); break
            clean = info["relgap"] <= 1e-6 and i0["relgap"] <= 1e-6 and info["iters"] == i0["iters"]
            if clean and np.max(np.abs(h - h0)) > 1e-6 * max(np.max(np.abs(h0)), 1e-3):
                bad.append((seed, which, size, "taps differ by %.3g from the unsharded solve (iters %d / %d)" % (np.max(np.abs(h - h0)), info["iters"], i0["iters"]))); break
        # (an unsharded solve that ends `numerical` / reduced-accuracy is repeated with the extended-precision KKT solve and
        # reports the iterations of both attempts; a row-sharded one is not: compare the counts only without that retry)
        # (round 6: the corrector's take-or-leave decisions amplify the rounding of the sharded sums: the counts may differ by a few
        #  iterations, the end points are compared above -- tests/test_switches_gpu.py ITER_SLACK)
        if i0["dd_iters"] == 0 and abs(info["iters"] - i0["iters"]) > max(4, 0.2 * i0["iters"]):
            bad.append((seed, which, size, "iterations %d vs %d" % (info["iters"], i0["iters"]))); break
    if (seed - lo) % 20 == 19:
        print("seeds %d..%d done, %d failures so far, %.0f s" % (lo, seed, len(bad), time.time() - t0), flush=True)
print("row-sharded loop-back, seeds %d..%d: %d failures; cases %s" % (lo, hi - 1, len(bad), sorted(counts.items())))
for b in bad:
    print("  FAIL", b)
