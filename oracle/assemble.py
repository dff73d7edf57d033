"""Oracle problem assembly (test infrastructure, see oracle/__init__.py).

Each `assemble_*` restates the assembly part of one reference designer and
returns the conic program the reference hands to its external solver, as
DENSE numpy arrays:

    minimise c'z   subject to   G z + s = h,   s in K
    K = R_+^l  x  Q_3^nq3  x  Q_big   (big cone optional)

Row order (the product's structured assembly uses the same order so the two
can be compared row by row in tests/test_assembly.py):
    fir_ap_cvx    : [A x<=U^2 | -A x<=-L^2 | A_stop x - rho<=0 | x1<=nP | -x1<=nP]
                    then Q3 cones i=2..n  (t=(n-i+1)P ; x_i ; x_{n+i-1})
    fir_qp_cvx    : Q3 band cones, Q3 transition cones, Q3 per-tap cones,
                    then the big cone (E ; x)
    fir_linprog   : [A x<=U | -A x<=-L]
    fir_qprog_phs : [Au x<=Bu | -Al x<=-Bl] then the big cone (t ; x), min t
                    (same minimiser as the reference's min 1/2 x'x)
"""
import numpy as np


class EarlyFail(Exception):
    """The reference returns status='Failed', h=[] before calling a solver."""


def matlab_linspace(d1, d2, n):
    # MATLAB linspace.m: y = d1 + (0:n1).*(d2-d1)./n1 ; y(1)=d1 ; y(end)=d2
    n1 = n - 1
    y = d1 + (np.arange(n, dtype=np.float64) * (d2 - d1)) / n1
    y[0] = d1
    y[-1] = d2
    return y


def _bands(w, f, a, d):
    """Band / transition split shared by all designers.

    fir_ap_cvx.m:51-82 == fir_qp_cvx.m:41-76 == ss/fir_linprog.m:111-132,166-174.
    Returns idx_band (concatenated in band order, 0-based), amp, dev per
    in-band sample, idx_tran, U_tran, L_tran scalars.
    """
    idx_band = []
    amp_band = []
    dev_band = []
    nband = len(f) // 2
    for b in range(nband):
        lo, hi = f[2 * b], f[2 * b + 1]
        idx = np.nonzero((w >= lo) & (w <= hi))[0]
        idx_band.append(idx)
        if lo == hi:
            amp = np.full(idx.shape, a[2 * b], dtype=np.float64)
        else:
            amp = a[2 * b] + (a[2 * b + 1] - a[2 * b]) * ((w[idx] - lo) / (hi - lo))
        amp_band.append(amp)
        dev_band.append(np.full(idx.shape, d[b], dtype=np.float64))
    idx_band = np.concatenate(idx_band) if idx_band else np.zeros(0, dtype=int)
    amp_band = np.concatenate(amp_band) if amp_band else np.zeros(0)
    dev_band = np.concatenate(dev_band) if dev_band else np.zeros(0)
    tmp = np.ones(len(w), dtype=bool)
    tmp[idx_band] = False
    idx_tran = np.nonzero(tmp)[0]
    return idx_band, amp_band, dev_band, idx_tran


# --------------------------------------------------------------------------
# fir_ap_cvx
# --------------------------------------------------------------------------
def assemble_fir_ap_cvx(n, f, a, d, obj=0.0, Peak=1e-3, grid_m=0):
    """fir_ap_cvx.m:44-142,160-169.  grid_m=0 -> reference rule m=2*n*15."""
    f = np.asarray(f, dtype=np.float64) * np.pi            # :44
    a = np.asarray(a, dtype=np.float64)
    d = np.asarray(d, dtype=np.float64)
    if obj < 0:
        raise ValueError("invalid input of obj")           # :171-173
    epsilon = 1e-10                                         # :40
    m0 = grid_m if grid_m else 2 * n * 15                   # :45-46
    w = matlab_linspace(-np.pi, np.pi, m0)                  # :47
    w = np.sort(np.concatenate([w, f]), kind="stable")      # :48
    idx_band, amp, dev, idx_tran = _bands(w, f, a, d)       # :51-69
    U_band = amp + dev
    L_band = amp - dev
    if len(idx_tran):                                       # :74-82
        U_tran = np.full(len(idx_tran), U_band.max())
        L_tran = np.full(len(idx_tran), min(0.0, L_band.min()))
    else:
        U_tran = np.zeros(0)
        L_tran = np.zeros(0)
    w = np.concatenate([w[idx_band], w[idx_tran]])          # :86-91
    m = len(w)
    j = np.arange(1, n)
    wj = np.outer(w, j)
    A = np.hstack([np.ones((m, 1)), 2 * np.cos(wj), 2 * np.sin(wj)])   # :100
    U_b = np.concatenate([U_band, U_tran]) ** 2             # :104-105
    L_b = np.concatenate([L_band, L_tran])                  # :109
    L_b[L_b < 0] = 0.0                                      # :110-111
    L_b = L_b ** 2                                          # :112
    L_b[L_b < epsilon ** 2] = epsilon ** 2                  # :115-116
    sq = np.sqrt(U_b)
    idx_stop = np.nonzero(sq < (sq.min() + 1e-2))[0]        # :125
    ns = len(idx_stop)
    N = 2 * n                                               # x(2n-1), ripple_stop
    nl = 2 * m + ns + 2
    nq3 = n - 1
    R = nl + 3 * nq3
    G = np.zeros((R, N))
    h = np.zeros(R)
    G[0:m, : 2 * n - 1] = A                                 # A_b=[A_U;-A_L] :119
    h[0:m] = U_b
    G[m:2 * m, : 2 * n - 1] = -A
    h[m:2 * m] = -L_b                                       # b=[U_b -L_b] :120
    G[2 * m:2 * m + ns, : 2 * n - 1] = A[idx_stop]          # :165
    G[2 * m:2 * m + ns, 2 * n - 1] = -1.0
    # i=1: norm(x(1)) <= n*Peak  (a 1-D norm is an absolute value)  :136,167
    r0 = 2 * m + ns
    G[r0, 0] = 1.0
    h[r0] = n * Peak
    G[r0 + 1, 0] = -1.0
    h[r0 + 1] = n * Peak
    for i in range(2, n + 1):                               # :137-142,166-168
        r = nl + 3 * (i - 2)
        h[r] = (n - i + 1) * Peak
        G[r + 1, i - 1] = -1.0                              # x(i)
        G[r + 2, n + i - 2] = -1.0                          # x(n+i-1)
    c = np.zeros(N)
    c[0] = 1.0                                              # minimize x(1)+obj*ripple_stop :162
    c[2 * n - 1] = obj
    return dict(c=c, G=G, h=h, l=nl, nq3=nq3, big=0,
                meta=dict(w=w, U_b=U_b, L_b=L_b, idx_stop=idx_stop, m=m,
                          n_band=len(idx_band), A=A))


# --------------------------------------------------------------------------
# fir_qp_cvx
# --------------------------------------------------------------------------
def assemble_fir_qp_cvx(n, f, a, d, k=100.0, obj=0.0, grid_m=0):
    """fir_qp_cvx.m:34-139,145-191.  obj scalar -> model A, 2 entries -> model B."""
    f = np.asarray(f, dtype=np.float64) * np.pi            # :34
    a = np.asarray(a, dtype=np.float64)
    d = np.asarray(d, dtype=np.float64)
    obj = np.atleast_1d(np.asarray(obj, dtype=np.float64))
    if len(obj) not in (1, 2):
        raise ValueError("invalid input of obj")           # :194-196
    m0 = grid_m if grid_m else n * 10                       # :35-36
    w = matlab_linspace(-np.pi, np.pi, m0)                  # :37
    w = np.sort(np.concatenate([w, f]), kind="stable")      # :38
    idx_band, M_band, D_band, idx_tran = _bands(w, f, a, d)  # :41-63
    wband = w[idx_band]                                     # :79-80 (no reorder of w)
    wtran = w[idx_tran]
    mb, mt = len(wband), len(wtran)
    t = np.arange(n)
    modelB = len(obj) == 2
    ne = 3 if modelB else 2
    N = 2 * n + ne
    iE = 2 * n + (1 if modelB else 0)
    iP = iE + 1
    nq3 = mb + mt + n
    R = 3 * nq3 + 1 + 2 * n
    G = np.zeros((R, N))
    h = np.zeros(R)

    def blocks(ws):                                         # Ai :99,107
        wt = np.outer(ws, t)
        c_, s_ = np.cos(wt), np.sin(wt)
        return np.hstack([c_, s_]), np.hstack([-s_, c_])

    r1, r2 = blocks(wband)
    Hd = M_band * np.exp(1j * (k * wband ** 2 - wband * (n - 1) / 2))   # :118
    rows = np.arange(mb) * 3
    if modelB:
        G[rows, 2 * n] = -D_band                            # <= D_band(i)*delta :176
    else:
        h[rows] = D_band                                    # <= D_band(i) :151
    G[rows + 1, : 2 * n] = -r1
    h[rows + 1] = -Hd.real
    G[rows + 2, : 2 * n] = -r2
    h[rows + 2] = -Hd.imag
    r1, r2 = blocks(wtran)
    rows = 3 * mb + np.arange(mt) * 3
    h[rows] = 1.1 if modelB else 1 + d.max() * 5            # :181 / :156
    G[rows + 1, : 2 * n] = -r1
    G[rows + 2, : 2 * n] = -r2
    for i in range(n):                                      # norm(F_i x)<=Peak :160-162
        r = 3 * (mb + mt) + 3 * i
        G[r, iP] = -1.0
        G[r + 1, i] = -1.0
        G[r + 2, n + i] = -1.0
    r = 3 * nq3                                             # norm(x)<=E_total :165
    G[r, iE] = -1.0
    G[r + 1 + np.arange(2 * n), np.arange(2 * n)] = -1.0
    c = np.zeros(N)
    if modelB:
        c[2 * n] = 1.0                                      # delta+obj1*E+obj2*Peak :172
        c[iE] = obj[0]
        c[iP] = obj[1]
    else:
        c[iE] = 1.0                                         # E_total+obj*Peak :147
        c[iP] = obj[0]
    return dict(c=c, G=G, h=h, l=0, nq3=nq3, big=1 + 2 * n,
                meta=dict(w=w, wband=wband, wtran=wtran, Hd=Hd, D_band=D_band,
                          M_band=M_band))


# --------------------------------------------------------------------------
# ss/fir_linprog
# --------------------------------------------------------------------------
def assemble_fir_linprog(n, f, a, d, grid_m=0):
    """ss/fir_linprog.m:46-132,161-240.  Raises EarlyFail for :66-75."""
    f = np.asarray(f, dtype=np.float64) * np.pi            # :46
    a = np.asarray(a, dtype=np.float64)
    d = np.asarray(d, dtype=np.float64)
    real_filter = not (f.min() < 0)                         # :47-51
    odd_filter = (n & 1) == 1                               # :56-60
    if not odd_filter:                                      # :66-75
        idx = np.nonzero(np.abs(f) == np.pi)[0]
        if np.any(a[idx] == 1):
            raise EarlyFail("n odd and frequency spec 1 at fs/2")
    nhalf = (n + 1) // 2                                    # :79
    nx = nhalf
    if not real_filter:                                     # :82-88
        nx = 2 * nhalf - 1 if odd_filter else 2 * nhalf
    if real_filter:                                         # :97-103
        m0 = grid_m if grid_m else 15 * n
        w = matlab_linspace(0.0, np.pi, m0)
    else:
        m0 = grid_m if grid_m else 2 * 15 * n
        w = matlab_linspace(-np.pi, np.pi, m0)
    w = np.sort(np.concatenate([w, f]), kind="stable")      # :107
    idx_band, amp, dev, idx_tran = _bands(w, f, a, d)       # :111-132
    U_band = amp + dev
    L_band = amp - dev
    if len(idx_tran):                                       # :166-174
        U_tran = np.full(len(idx_tran), U_band.max())
        L_tran = np.full(len(idx_tran), min(0.0, L_band.min()))
    else:
        U_tran = np.zeros(0)
        L_tran = np.zeros(0)
    nb = len(idx_band)
    w = np.concatenate([w[idx_band], w[idx_tran]])          # :178-183
    m = len(w)
    if odd_filter:                                          # :195-213
        jj = np.arange(1, nhalf)
        Acos = np.hstack([np.ones((m, 1)), 2 * np.cos(np.outer(w, jj))])
        Asin = 2 * np.sin(np.outer(w, jj))
    else:
        jj = np.arange(nhalf) + 0.5
        Acos = 2 * np.cos(np.outer(w, jj))
        Asin = 2 * np.sin(np.outer(w, jj))
    A = Acos if real_filter else np.hstack([Acos, Asin])    # :217
    U_b = np.concatenate([U_band, U_tran])                  # :222
    L_b = np.concatenate([L_band, L_tran])                  # :227
    G = np.vstack([A, -A])                                  # :231
    h = np.concatenate([U_b, -L_b])                         # :232
    c = A[nb:].sum(axis=0)                                  # fmin :240
    return dict(c=c, G=G, h=h, l=2 * m, nq3=0, big=0,
                meta=dict(w=w, U_b=U_b, L_b=L_b, nhalf=nhalf, nx=nx, A=A,
                          real_filter=real_filter, odd_filter=odd_filter,
                          n_band=nb))


def fill_h_linprog(x, nhalf, real_filter, odd_filter):
    """ss/fir_linprog.m:274-296 (column vector of n taps)."""
    x = np.asarray(x, dtype=np.float64)
    if real_filter:
        if odd_filter:
            return np.concatenate([x[:0:-1], x]).astype(np.complex128)
        return np.concatenate([x[::-1], x]).astype(np.complex128)
    if odd_filter:
        hh = x[:nhalf] + 1j * np.concatenate([[0.0], x[nhalf:]])
        return np.concatenate([np.conj(hh[:0:-1]), hh])
    hh = x[:nhalf] + 1j * x[nhalf:]
    return np.concatenate([np.conj(hh[::-1]), hh])


# --------------------------------------------------------------------------
# ss/fir_qprog_phs
# --------------------------------------------------------------------------
def assemble_fir_qprog_phs(n, f, ac, dc, grid_m=0):
    """ss/fir_qprog_phs.m:49-128,178-342 with min 1/2 x'x restated as
    min t s.t. ||x||<=t.  Raises EarlyFail for :193-202, ValueError for the
    reference's error() calls."""
    f = np.asarray(f, dtype=np.float64)
    ac = np.asarray(ac, dtype=np.complex128)
    dc = np.asarray(dc, dtype=np.complex128)
    nband = len(f) // 2                                     # :49
    for b in range(nband):                                  # :53-57
        if ac[2 * b] != ac[2 * b + 1]:
            raise ValueError("Does not support sloped bands")
    a = np.abs(ac[0::2])                                    # :61,64
    aphs = np.angle(a)                                      # :65  (angle(abs(.)) == 0)
    d = np.abs(dc)                                          # :66
    dphs = np.angle(dc)                                     # :67
    for b in range(nband):                                  # :74-80
        if (a[b] + d[b]) * (a[b] - d[b]) < 0:
            if a[b] != 0 or dphs[b] != 0:
                raise ValueError("Bands straddling 0 must have a = 0, angle(d) = 0")
    err_tol = 0.05                                          # :85
    dphs = dphs.copy()
    for b in range(nband):                                  # :86-98
        if a[b] != 0:
            magerr_inner = (a[b] - d[b]) * (1.0 / np.cos(dphs[b]) - 1)
            if magerr_inner >= 2 * d[b]:
                dphs[b] = 0.99 * np.arccos((a[b] - d[b]) / (a[b] + d[b]))
    n_phs_tran = int(np.ceil(2 * np.pi / np.arccos(1 - err_tol)))       # :103
    amax = (a + d).max()                                    # :104
    phs_tran = list(np.arange(n_phs_tran + 1) / n_phs_tran * 2 * np.pi)  # :105
    phs_band = []
    for b in range(nband):                                  # :106-123
        if a[b] == 0:
            phs_band.append(np.arange(n_phs_tran + 1) / n_phs_tran * 2 * np.pi)
        else:
            phs_tol = np.arccos(1 - (err_tol * 2 * d[b]))
            n_phs = int(np.ceil(2 * dphs[b] / phs_tol))
            if n_phs < 1:
                # reference: (0:0)/0 = NaN, then phs_band{band}(2) indexes out of range
                raise ValueError("passband needs a non-zero phase ripple")
            phs_band.append((np.arange(n_phs + 1) / n_phs * 2 - 1) * dphs[b] + aphs[b])
            if (a[b] + d[b]) >= amax * (1 - err_tol):
                phs_tran += [aphs[b] - dphs[b], aphs[b] + dphs[b]]
    phs_tran = np.mod(np.asarray(phs_tran), 2 * np.pi)      # :127
    phs_tran = np.unique(np.concatenate([phs_tran, [0.0, 2 * np.pi]]))   # :128
    f = f * np.pi                                           # :178
    odd_filter = (n & 1) == 1                               # :183-187
    if not odd_filter:                                      # :193-202
        idx = np.nonzero(np.abs(f) == np.pi)[0]
        if np.any(np.abs(ac[idx]) != 0):
            raise EarlyFail("n odd and frequency spec non-zero at fs/2")
    nhalf = (n + 1) // 2                                    # :206
    m0 = grid_m if grid_m else 2 * 15 * n                   # :213,218
    w = matlab_linspace(-np.pi, np.pi, m0)                  # :219
    w = np.sort(np.concatenate([w, f]), kind="stable")      # :223
    if odd_filter:                                          # :227-231
        t = np.arange(-(nhalf - 1), nhalf, dtype=np.float64)
    else:
        t = np.arange(-nhalf, nhalf, dtype=np.float64) + 0.5
    W = np.exp(-1j * np.outer(w, t))
    Au, Bu, Al, Bl = [], [], [], []
    idx_band = []
    inph = lambda Wt: np.hstack([Wt.real, -Wt.imag])        # in-phase part
    quad = lambda Wt: np.hstack([Wt.imag, Wt.real])         # quadrature part
    for b in range(nband):                                  # :238-274
        idx = np.nonzero((w >= f[2 * b]) & (w <= f[2 * b + 1]))[0]
        idx_band.append(idx)
        pb = phs_band[b]
        phs_diff = np.angle(np.exp(1j * pb[1]) * np.exp(-1j * pb[0]))   # :244
        a_mid = (a[b] + d[b]) * np.cos(phs_diff / 2)        # :246
        for pi_ in range(len(pb) - 1):                      # :247-253
            phs_mid = pb[pi_] + phs_diff / 2
            Au.append(inph(W[idx] * np.exp(-1j * phs_mid)))
            Bu.append(np.full(len(idx), a_mid))
        if a[b] != 0:                                       # :257-273
            Al.append(inph(W[idx] * np.exp(-1j * aphs[b])))
            Bl.append(np.full(len(idx), a[b] - d[b]))
            Au.append(quad(W[idx] * np.exp(-1j * pb[-1])))
            Bu.append(np.zeros(len(idx)))
            Al.append(quad(W[idx] * np.exp(-1j * pb[0])))
            Bl.append(np.zeros(len(idx)))
    idx_band = np.concatenate(idx_band)
    tmp = np.ones(len(w), dtype=bool)                       # :278-280
    tmp[idx_band] = False
    idx_tran = np.nonzero(tmp)[0]
    for i in range(len(phs_tran) - 1):                      # :306-313
        phs_diff = phs_tran[i + 1] - phs_tran[i]
        phs_mid = phs_tran[i] + phs_diff / 2
        Au.append(inph(W[idx_tran] * np.exp(-1j * phs_mid)))
        Bu.append(np.full(len(idx_tran), amax * np.cos(phs_diff / 2)))
    Au = np.vstack(Au)
    Bu = np.concatenate(Bu)
    if Al:
        Al = np.vstack(Al)
        Bl = np.concatenate(Bl)
    else:
        Al = np.zeros((0, 2 * n))
        Bl = np.zeros(0)
    A = np.vstack([Au, -Al])                                # :318
    B = np.concatenate([Bu, -Bl])                           # :319
    p = A.shape[0]
    N = 2 * n + 1
    R = p + 1 + 2 * n
    G = np.zeros((R, N))
    h = np.zeros(R)
    G[:p, : 2 * n] = A
    h[:p] = B
    G[p, 2 * n] = -1.0                                      # (t ; x) in Q_{2n+1}
    G[p + 1 + np.arange(2 * n), np.arange(2 * n)] = -1.0
    c = np.zeros(N)
    c[2 * n] = 1.0
    return dict(c=c, G=G, h=h, l=p, nq3=0, big=1 + 2 * n,
                meta=dict(w=w, t=t, A=A, B=B, phs_tran=phs_tran,
                          phs_band=phs_band, idx_tran=idx_tran))
