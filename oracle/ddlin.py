"""ctypes binding of oracle/ddlin.c (double-double dense kernels; test infrastructure, see oracle/__init__.py).

Built by `make -C oracle all` into oracle/_build/libddlin.so (git-ignored, travels to the GPU box)."""
import ctypes
import os

import numpy as np

_LIB = None
_D = ctypes.POINTER(ctypes.c_double)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libddlin.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/_build/libddlin.so missing: run `make -C oracle all`")
        _LIB = ctypes.CDLL(path)
        _LIB.dd_chol.restype = ctypes.c_int
    return _LIB


def _p(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_D)


def rank_k(U, x, Hh, Hl):
    """H (dd, lower triangle) += sum_r x[r] U[r] U[r]'."""
    k, N = U.shape
    lib().dd_rank_k(N, k, _p(U), _p(x), _p(Hh), _p(Hl))


def chol(Hh, Hl, pivtol, d0):
    """In-place lower Cholesky in dd; returns the number of replaced pivots."""
    return lib().dd_chol(Hh.shape[0], _p(Hh), _p(Hl), ctypes.c_double(pivtol), _p(d0))


def cho_solve(Lh, Ll, Bh, Bl):
    """B (N x nrhs, dd) <- (L L')^-1 B, in place."""
    N, nrhs = Bh.shape
    lib().dd_cho_solve(N, nrhs, _p(Lh), _p(Ll), _p(Bh), _p(Bl))


def rows_times(U, Xh, Xl):
    k, N = U.shape
    nrhs = Xh.shape[1]
    Yh = np.zeros((k, nrhs))
    Yl = np.zeros((k, nrhs))
    lib().dd_rows_times(k, N, nrhs, _p(U), _p(Xh), _p(Xl), _p(Yh), _p(Yl))
    return Yh, Yl


def cols_times_acc(U, Yh, Yl, Xh, Xl):
    k, N = U.shape
    lib().dd_cols_times_acc(k, N, Yh.shape[1], _p(U), _p(Yh), _p(Yl), _p(Xh), _p(Xl))


def vec_mul_d(ah, al, b):
    oh = np.empty_like(ah)
    ol = np.empty_like(ah)
    lib().dd_vec_mul_d(ah.size, _p(ah), _p(al), _p(np.ascontiguousarray(b)), _p(oh), _p(ol))
    return oh, ol


def vec_sub(ah, al, bh, bl):
    oh = np.empty_like(ah)
    ol = np.empty_like(ah)
    lib().dd_vec_sub(ah.size, _p(ah), _p(al), _p(bh), _p(bl), _p(oh), _p(ol))
    return oh, ol
