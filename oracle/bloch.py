"""Forward SLR / Bloch verification: RF pulse -> Cayley-Klein parameters over off-resonance.
TEST INFRASTRUCTURE.  NumPy restatement of rf_tools/abrm.m:24-62 (the .m twin of the MEX abrx) and of abr.m:19-34;
the magnetisation identities are those of abr.m:11-14.  Parity unpinned against reference outputs (the reference
holds none for this step; blochC.c / abrx.c are MEX files and need mex.h) -- what pins it instead is the SLR
identity: the beta of a pulse made by the (reference-C-pinned) inverse SLR of b is the DFT of b.
"""
import numpy as np


def abrm(rf, g=None, x=None):
    """[a, b] = abrm(rf, g, x): rf in radians per sample (sum(rf) = flip angle), g per-sample gradient weights
    (default 2 pi / n each, so that x counts cycles over the pulse), x positions / off-resonances."""
    rf = np.asarray(rf, dtype=np.complex128).ravel()
    if x is None:
        x, g = g, None
    if g is None:
        g = np.ones(len(rf)) * 2 * np.pi / len(rf)
    g = np.asarray(g, dtype=np.float64).ravel()
    x = np.asarray(x, dtype=np.float64).ravel()
    a = np.ones(len(x), dtype=np.complex128)
    b = np.zeros(len(x), dtype=np.complex128)
    for m in range(len(rf)):
        om = x * g[m]
        phi = np.sqrt(abs(rf[m]) ** 2 + om ** 2)
        safe = np.where(phi > 0, phi, 1.0)
        n1, n2, n3 = rf[m].real / safe, rf[m].imag / safe, om / safe
        av = np.cos(phi / 2) - 1j * n3 * np.sin(phi / 2)
        bv = -1j * (n1 + 1j * n2) * np.sin(phi / 2)
        a, b = av * a - np.conj(bv) * b, bv * a + np.conj(av) * b
    return a, b


def abr(rf, g=None, x=None):
    """abr.m: Le Roux's convention on beta, b = -conj(b) of the simulation."""
    a, b = abrm(rf, g, x)
    return a, -np.conj(b)


def mxy_excitation(a, b):
    return 2 * np.conj(a) * b                      # abr.m:11


def mz_inversion(b):
    return 1 - 2 * (b * np.conj(b)).real           # abr.m:12


def hard_pulse_ab(rf, x):
    """Forward SLR transform in the hard-pulse model the inverse transform (ab2rf.m) inverts exactly: per sample
    free precession z^-1 = exp(-2 pi i x / n) on beta, then the hard pulse (C, S) -- Pauly et al. 1991, eq. 9-10.
    For rf = ab2rf(a, b):  b(x) = z^-(n-1) ... up to a linear phase |b(x)| = |sum_k b_k exp(+i w k)|, w = -2 pi x / n."""
    rf = np.asarray(rf, dtype=np.complex128).ravel()
    x = np.asarray(x, dtype=np.float64).ravel()
    n = len(rf)
    a = np.ones(len(x), dtype=np.complex128)
    b = np.zeros(len(x), dtype=np.complex128)
    zi = np.exp(-2j * np.pi * x / n)
    for m in range(n):
        th = abs(rf[m])
        C = np.cos(th / 2)
        S = 1j * np.exp(1j * np.angle(rf[m])) * np.sin(th / 2)
        a, b = C * a - np.conj(S) * zi * b, S * a + C * zi * b
    return a, b
