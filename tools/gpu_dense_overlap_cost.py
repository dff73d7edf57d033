"""What the chunked Gram product of the dense row-sharded build costs on ONE GPU (no collective can overlap anything here): one dense
design with the one-launch product against MBFIR_AR_OVERLAP=2 (the chunked form without shards) at 1 / 2 / 4 / 8 chunks, then the
design row-sharded over two contexts of the GPU (loop-back all-reduce) in both forms: iterations, collectives, bytes per build.
    python tools/gpu_dense_overlap_cost.py 512 16384      |      ... 2048 131072 (BASELINE config 5)"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); warnings.filterwarnings("ignore")
import numpy as np
import mbfir
from conftest import c13                          # noqa: E402
from test_shard_gpu import _run_sharded           # noqa: E402
n, m = int(sys.argv[1]), int(sys.argv[2])
f, a, d = c13(n, "duration")
args = (n, f, a, d, 0.1, 1e-3)
o = mbfir.make_opts(grid_m=m, dense_trig=1)
mbfir.fir_ap_cvx(*args, opts=o)
def one(tag):
    t = time.time(); h, s, i = mbfir.fir_ap_cvx(*args, opts=o, info=True); dt = time.time() - t
    print("%-34s %s %3d iterations %8.1f ms  product+folds %7.3f ms per build  factorisation %6.3f ms per build" % (
        tag, s, i["iters"], 1e3 * dt, i["ms_gram"] / max(1, i["builds"]), i["ms_chol"] / max(1, i["builds"])), flush=True)
    return h, i
h0, i0 = one("one launch")
for ch in (1, 2, 4, 8):
    os.environ["MBFIR_AR_OVERLAP"] = "2"; os.environ["MBFIR_AR_CHUNKS"] = str(ch)
    h, i = one("chunked x%d (no shards)" % ch)
    print("      objective %.3e apart, taps %.2e" % (abs(i["pcost"] - i0["pcost"]), np.abs(h - h0).max() / np.abs(h0).max()))
os.environ.pop("MBFIR_AR_CHUNKS")
for mode, tag in (("1", "tiles, chunk by chunk on stream 2"), ("0", "assembled H after the build")):
    os.environ["MBFIR_AR_OVERLAP"] = mode
    t = time.time(); res = _run_sharded("fir_ap_cvx", args, 2, dense=1, grid_m=m); dt = time.time() - t
    for r, (h, s, i) in enumerate(res):
        print("2 ranks, %s: rank %d %s %d iterations, %d collectives, %.1f MB per build, taps vs rank 0 %s, objective vs unsharded %.2e; %.2f s wall" % (
            tag, r, s, i["iters"], i["collectives"], i["collective_bytes"] / max(1, i["builds"]) / 1e6,
            "identical" if np.array_equal(h, res[0][0]) else "DIFFERENT", abs(i["pcost"] - i0["pcost"]), dt), flush=True)
