import os, sys, time, json, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
warnings.filterwarnings("ignore")
import numpy as np, conftest, mbfir
from conftest import CASES
g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'golden.json')))
which, args = CASES["lin_real64"]
mbfir.fir_linprog(*args)
t0 = time.time(); h, s, i = mbfir.fir_linprog(*args, info=True); t = time.time() - t0
hg = np.array(g["lin_real64"]["h"]["re"]) + 1j * np.array(g["lin_real64"]["h"]["im"])
print(s, i["iters"], "%.2f ms" % (t * 1e3), "relinf %.1e" % (np.abs(h - hg).max() / np.abs(hg).max()), "rows", i["n_rows"], "freq", i["n_freq"])
