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


# ---- Bloch-equation simulator with relaxation (bloch_simulation/blochC.c, blochH.c) ----------------------
# NumPy restatement of calcrotmat (blochC.c:171-236), blochsim (:283-418) and blochsimfz (:422-512), vectorised
# over the (frequency, position) pairs.  Parity unpinned against reference outputs: the file holds the MEX
# gateway and includes mex.h, so it cannot be compiled here; what pins it is the closed loop with abrm above
# (no relaxation: the same rotation per sample) and the analytic free-precession / recovery solution.
GAMMA_C13 = 6726.1          # blochC.c:5   (rad/s/G)
GAMMA_H1 = 26754.0          # blochH.c:6
TWOPI_REF = 6.283185        # blochC.c:6 -- the reference's truncated constant, kept


def calcrotmat(nx, ny, nz):
    """Rotation by |n| about n (blochC.c:171-236) for arrays of axes; returns R with shape (..., 3, 3),
    R[..., i, j] = the reference's column-major rmat[i + 3 j]."""
    nx, ny, nz = np.broadcast_arrays(np.asarray(nx, float), np.asarray(ny, float), np.asarray(nz, float))
    phi = np.sqrt(nx * nx + ny * ny + nz * nz)
    safe = np.where(phi > 0, phi, 1.0)
    hp = phi / 2
    cp = np.cos(hp)
    sp = np.sin(hp) / safe
    ar, ai, br, bi = cp, -nz * sp, ny * sp, -nx * sp
    R = np.empty(phi.shape + (3, 3))
    R[..., 0, 0] = ar * ar - ai * ai - br * br + bi * bi
    R[..., 1, 0] = -2 * ar * ai - 2 * br * bi
    R[..., 2, 0] = -2 * ar * br + 2 * ai * bi
    R[..., 0, 1] = 2 * ar * ai - 2 * br * bi
    R[..., 1, 1] = ar * ar - ai * ai + br * br - bi * bi
    R[..., 2, 1] = -2 * ai * br - 2 * ar * bi
    R[..., 0, 2] = 2 * ar * br + 2 * ai * bi
    R[..., 1, 2] = 2 * ar * bi - 2 * ai * br
    R[..., 2, 2] = ar * ar + ai * ai - br * br - bi * bi
    R[phi == 0] = np.eye(3)
    return R


def blochsimfz(b1, grad, tsteps, t1, t2, df, pos, mode=0, m0=None, gamma=GAMMA_C13):
    """[mx, my, mz] = bloch(b1, grad, tsteps, t1, t2, df, pos, mode, mx0, my0, mz0)  (blochC.c:422-512).
    b1 complex (Gauss), grad (ntime, 3) or None (G/cm), tsteps interval lengths (s, scalar or ntime), df (Hz),
    pos (npos, 3) (cm), mode bit 0: steady state, bit 1: record every time point; m0 (nfreq, npos, 3) or None
    (= equilibrium [0 0 1]).  Returns m of shape (nfreq, npos, ntout, 3), ntout = ntime if mode & 2 else 1."""
    b1 = np.asarray(b1, dtype=np.complex128).ravel()
    nt = len(b1)
    grad = np.zeros((nt, 3)) if grad is None else np.asarray(grad, float).reshape(nt, -1)
    if grad.shape[1] < 3:
        grad = np.concatenate([grad, np.zeros((nt, 3 - grad.shape[1]))], 1)
    ts = np.broadcast_to(np.asarray(tsteps, float).ravel(), (nt,)) if np.size(tsteps) in (1, nt) else None
    df = np.asarray(df, float).ravel()
    pos = np.asarray(pos, float).reshape(-1, 3) if np.ndim(pos) > 0 and np.size(pos) % 3 == 0 and np.ndim(pos) == 2 \
        else np.stack([np.asarray(pos, float).ravel(), np.zeros(np.size(pos)), np.zeros(np.size(pos))], 1)
    nf, npos = len(df), len(pos)
    e1 = np.exp(-ts / t1)                                                 # :460-464
    e2 = np.exp(-ts / t2)
    m = np.zeros((nf, npos, 3))
    m[..., 2] = 1.0
    if m0 is not None:
        m = np.array(m0, dtype=float).reshape(nf, npos, 3)
    gpos = pos * gamma                                                    # :317-319 gammadx ...

    def run(mstart, sim_mode):
        """blochsim for every (f, p): sim_mode 0 endpoint, 1 steady state, 2 all time points."""
        A = np.broadcast_to(np.eye(3), (nf, npos, 3, 3)).copy()
        B = np.zeros((nf, npos, 3))
        mc = mstart.copy()
        out = np.zeros((nf, npos, nt, 3)) if sim_mode == 2 else None
        for t in range(nt):
            rotz = -((grad[t] @ gpos.T)[None, :] + df[:, None] * TWOPI_REF) * ts[t]      # :330-331
            rotx = -b1[t].real * gamma * ts[t]                                           # :332
            roty = b1[t].imag * gamma * ts[t]                                            # :333
            R = calcrotmat(rotx, roty, rotz)
            dec = np.array([e2[t], e2[t], e1[t]])
            if sim_mode == 1:
                A = dec[:, None] * (R @ A)                                               # :336-356
                B = dec * np.einsum("fpij,fpj->fpi", R, B)
                B[..., 2] += 1 - e1[t]
            else:
                mc = dec * np.einsum("fpij,fpj->fpi", R, mc)
                mc[..., 2] += 1 - e1[t]
                if sim_mode == 2:
                    out[:, :, t] = mc
        if sim_mode == 1:                                                                # :406-415  M = inv(I - A) B
            return np.linalg.solve(np.eye(3) - A, B[..., None])[..., 0]
        return out if sim_mode == 2 else mc

    if mode == 3:                                                                        # :477-490
        return run(run(m, 1), 2)
    r = run(m, mode)
    return r if mode == 2 else r[:, :, None, :]
