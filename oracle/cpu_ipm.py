"""ctypes binding of oracle/cpu_ipm.cpp, the C++ / OpenMP twin of conic_ipm.solve (test infrastructure, see
oracle/__init__.py): bench.py's cpu_baseline leg times it at 1 thread and at all cores, and
tests/test_cpu_ipm_cpu.py holds it to the NumPy oracle."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libcpu_ipm.so")
_lib = None
_dp = ctypes.POINTER(ctypes.c_double)


def _bind(path):
    L = ctypes.CDLL(path)
    L.cpu_ipm_solve.restype = ctypes.c_int
    L.cpu_ipm_solve.argtypes = ([ctypes.c_int, ctypes.c_int, _dp, _dp, _dp] + [ctypes.c_int] * 4 + [ctypes.c_double] * 3 +
                                [ctypes.c_int, ctypes.c_int, _dp, _dp])
    L.cpu_ipm_isa.restype = ctypes.c_char_p
    return L


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.run(["make", "-C", _HERE, "_build/libcpu_ipm.so"], check=True, capture_output=True)
        _lib = _bind(_PATH)
    return _lib


def use_native_build():
    """Rebuild the solver with -march=native ON THIS MACHINE (the one the timing runs on) and switch to it; returns the
    instruction set the build selected, or None (and keeps the portable x86-64-v3 build) when the compiler is missing
    or the rebuilt library does not load."""
    global _lib
    path = os.path.join(_HERE, "_build", "libcpu_ipm_native.so")
    try:
        subprocess.run(["make", "-B", "-C", _HERE, "_build/libcpu_ipm_native.so"], check=True, capture_output=True)
        _lib = _bind(path)
        return _lib.cpu_ipm_isa().decode()
    except Exception:                                       # noqa: BLE001
        return None


def isa():
    return lib().cpu_ipm_isa().decode()


def solve(c, G, h, l, nq3=0, big=0, max_iter=200, feastol=1e-8, abstol=1e-10, reltol=1e-8, refine=2, threads=0):
    """Same arguments and result keys as conic_ipm.solve (x, status, iters, pcost, dcost, gap, relgap, pres, dres,
    chol_fixes) plus seconds_factor, seconds_total, threads.  threads = 0: OpenMP's default (all cores)."""
    G = np.ascontiguousarray(G, dtype=np.float64)
    h = np.ascontiguousarray(h, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    R, N = G.shape
    x = np.zeros(N)
    info = np.zeros(16)
    p = lambda a: a.ctypes.data_as(_dp)
    rc = lib().cpu_ipm_solve(R, N, p(G), p(h), p(c), int(l), int(nq3), int(big), int(max_iter), feastol, abstol, reltol, int(refine),
                             int(threads), p(x), p(info))
    if rc < 0:
        raise ValueError("cone dimensions do not add up to the rows of G")
    keys = ("pcost", "dcost", "gap", "relgap", "pres", "dres")
    out = dict(status=int(info[0]), iters=int(info[1]), x=x, chol_fixes=int(info[8]), seconds_factor=float(info[9]),
               seconds_total=float(info[10]), threads=int(info[11]))
    out.update({k: float(info[2 + i]) for i, k in enumerate(keys)})
    return out
