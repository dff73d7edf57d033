"""BASELINE config 5 (n=2048, 131072 grid points) row-sharded over 2 (or argv[1]) contexts of ONE GPU with the loop-back all-reduce of
tests/test_shard_gpu.py: iterations, collectives per iteration, the ranks' row / frequency counts, wall-clock (loop-back on one GPU:
the ranks share the chip, so this is not a scaling figure) beside the unsharded solve."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from conftest import c13
from test_shard_gpu import _run_sharded
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, m = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2048, 131072)
f, a, d = c13(n, "duration")
args = (n, f, a, d, 0.1, 1e-3)
mbfir.fir_ap_cvx(*args, opts=mbfir.make_opts(grid_m=m))
t = time.time(); h0, s0, i0 = mbfir.fir_ap_cvx(*args, opts=mbfir.make_opts(grid_m=m), info=True); t0 = time.time() - t
print("unsharded: %s, %d iterations, %.3f s (chol %.1f ms, normal matrix %.1f ms), rows %d, frequencies %d" % (s0, i0["iters"], t0, i0["ms_chol"], i0["ms_gram"], i0["n_rows"], i0["n_freq"]))
_run_sharded("fir_ap_cvx", args, size, grid_m=m)
t = time.time(); res = _run_sharded("fir_ap_cvx", args, size, grid_m=m); t1 = time.time() - t
for r, (h, s, info) in enumerate(res):
    print("rank %d of %d: %s, %d iterations, %d collectives = %.2f per iteration, rows %d, frequencies %d, taps vs unsharded %.2e, vs rank 0 %s" % (
        r, size, s, info["iters"], info["collectives"], info["collectives"] / max(1, info["iters"]), info["n_rows"], info["n_freq"],
        np.abs(h - h0).max() / np.abs(h0).max(), "identical" if np.array_equal(h, res[0][0]) else "DIFFERENT"))
print("sharded x%d in loop-back on one GPU: %.3f s wall (host-side loop-back all-reduce, both ranks on one chip)" % (size, t1))
