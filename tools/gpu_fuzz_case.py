"""Re-run one case of tools/gpu_fuzz.py (same RNG stream) with the IPM trace for both device paths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util, io, contextlib
seed, target = int(sys.argv[1]), int(sys.argv[2])
src = open(os.path.join(ROOT, "tools", "gpu_fuzz.py")).read()
# stop at the target case and expose its arguments
src = src.replace("    res = []\n    for dense in (0, 1):", "    if case == TARGET:\n        import pickle; pickle.dump((which, args), open('/tmp/fuzz_case.pkl', 'wb')); break\n    continue\n    res = []\n    for dense in (0, 1):")
src = src.replace("rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 12345)", "rng = np.random.default_rng(%d)" % seed)
src = src.replace("ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60", "ncase = %d\nTARGET = %d" % (target + 1, target))
exec(compile(src, "fuzz", "exec"), {"__name__": "fuzz", "__file__": os.path.join(ROOT, "tools", "gpu_fuzz.py")})
import pickle, mbfir
which, args = pickle.load(open('/tmp/fuzz_case.pkl', 'rb'))
print(which, [a if not hasattr(a, "shape") else a.round(5).tolist() for a in args])
for dense in (0, 1):
    try:
        h, s, i = getattr(mbfir, which)(*args, info=True, opts=mbfir.make_opts(dense_trig=dense, verbose=1))
        print("dense=%d:" % dense, s, i["iters"], i["pcost"], i["pres"], i["dres"], i["relgap"], flush=True)
    except Exception as e:
        print("dense=%d EXC %r" % (dense, e), flush=True)
