"""Run one named case from tests/conftest.py (or 'c13:<n>') verbosely on the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import mbfir
from conftest import CASES, c13
name = sys.argv[1]
if name.startswith("c13:"):
    n = int(name[4:]); f, a, d = c13(64); fn, args = "fir_ap_cvx", (n, f, a, d, 0.1, 1e-3)
else:
    fn, args = CASES[name]
h, s, i = getattr(mbfir, fn)(*args, dbg=1, info=True)
print(s, i)
