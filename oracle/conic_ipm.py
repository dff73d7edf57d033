"""Oracle solver core (test infrastructure, see oracle/__init__.py).

Dense primal-dual interior-point method for

    minimise c'x   s.t.  G x + s = h,  s in K = R_+^l x Q_3^nq3 x Q_big

written from the published mathematics, not from any file of the reference
(the reference delegates this step to CVX/SDPT3, linprog and quadprog, which
are not vendored -- fir_ap_cvx.m:160-169, fir_qp_cvx.m:145-191,
ss/fir_linprog.m:245-251, ss/fir_qprog_phs.m:339-342):

  * homogeneous self-dual embedding (tau, kappa) so that infeasible
    instances end with a Farkas certificate instead of diverging
    (L. Vandenberghe, "The CVXOPT linear and quadratic cone program solvers",
    2010, section 6-7; A. Domahidi et al., "ECOS", ECC 2013);
  * Nesterov-Todd scaling for the second-order cones;
  * Mehrotra predictor-corrector, step fraction 0.99, sigma = min((1-alpha_aff)^3, 0.25): on the
    fir_ap_cvx programs the affine step is short (0.05-0.5) at every scale -- thousands of positivity rows
    S(w_i) >= 1e-20 sit at slacks spread over many decades -- so the plain rule spends every fifth iteration
    on an almost pure centring step; the cap saves 10-18 % of the iterations on them and changes no verdict
    (DESIGN.md section 5);
  * KKT systems reduced to the normal equations  (G' W^-2 G) dx = rhs  and
    solved by dense Cholesky, with one or two steps of iterative refinement
    on the un-regularised reduced system.

The product's HIP solver implements the same iteration (same initial point,
same step rule, same stopping rule), so the two agree to rounding and the
parity tests can compare taps at 1e-6 relative l-inf.
"""
import numpy as np

STEP = 0.99
SIGMA_MAX = 0.25              # cap of Mehrotra's centring parameter (see solve())
SIGMA_MAX_CORR = 0.05         # ... where a centrality corrector follows the direction (programs with orthant rows; see solve())
# one centrality corrector per iteration (J. Gondzio, "Multiple centrality corrections in a primal-dual method for linear
# programming", Comput. Optim. Appl. 6, 1996), on the rows of the non-negative orthant: see solve()
CORR_DELTA = 0.5              # the corrector aims at the step alpha + CORR_DELTA (capped at 1)
CORR_BMIN, CORR_BMAX = 0.1, 10.0      # box of the complementarity products, in units of the target sigma * mu
CORR_ACCEPT = 1.01            # the corrected direction is taken when its step is at least this factor longer
CORR_ETA = 1.0                # ... and its (unrefined) solve leaves no more than CORR_ETA ||rx|| of the dual equation
# end game (round 6): an iterate that meets the stopping rule is kept, and the iteration goes on until the gap measures are
# POLISH times smaller (or POLISH_MAX more iterations): see solve().  Measured on the 58-tap S-C13 case (the oracle on 1 / 2 / 4 / 8
# BLAS threads, i.e. four roundings of one path): taps 1e-6 apart without an end game, 4e-8 with POLISH = 0.1 (+0.5 iterations),
# 1e-8 with 0.01 (+1.75).  0.1 was tried in the product: BASELINE config 4's design (j = 0, Peak = 1e-4; n = 200, ~55 iterations) then ends
# 1.4e-6 from the oracle's taps -- outside north_star's 1e-6 -- and the headline batch gains nothing (49.0 against 49.2 iterations: its
# end game runs into POLISH_MAX either way); stopping the end game when an iteration gains less than a factor five was tried with it.
POLISH = 1e-2
POLISH_MAX = 3
POLISH_SWEEPS = 1             # final approach and end game: refinement sweeps on top of the controller's count (see solve())
POLISH_APPROACH = 30.0         # ... the final approach: a gap measure within this factor of its tolerance
STATUS_OPTIMAL = 0
STATUS_PRIMAL_INFEASIBLE = 1
STATUS_DUAL_INFEASIBLE = 2
STATUS_MAXIT = 3
STATUS_NUMERICAL = 4


class _Cone:
    """K = R_+^l x Q_3^nq3 x Q_big acting on flat vectors of length R."""

    def __init__(self, l, nq3, big):
        self.l, self.nq3, self.big = l, nq3, big
        self.o3 = l
        self.ob = l + 3 * nq3
        self.R = self.ob + big
        self.degree = l + nq3 + (1 if big else 0)

    def split(self, v):
        return (v[: self.l], v[self.o3:self.ob].reshape(self.nq3, 3), v[self.ob:])

    def e(self):
        v = np.zeros(self.R)
        v[: self.l] = 1.0
        v[self.o3:self.ob:3] = 1.0
        if self.big:
            v[self.ob] = 1.0
        return v

    def min_residual(self, v):
        """max over cones of -(distance inside): <0 means strictly interior."""
        vl, vq, vb = self.split(v)
        t = -np.inf
        if self.l:
            t = max(t, -vl.min())
        if self.nq3:
            t = max(t, (np.hypot(vq[:, 1], vq[:, 2]) - vq[:, 0]).max())
        if self.big:
            t = max(t, np.linalg.norm(vb[1:]) - vb[0])
        return t


def _jres(v0, nrm1):
    # v0^2 - ||v1||^2 without the cancellation of the naive form
    return (v0 - nrm1) * (v0 + nrm1)


class _Scaling:
    """Nesterov-Todd scaling W (symmetric) for the current (s, z)."""

    def __init__(self, cone, s, z):
        self.cone = cone
        sl, sq, sb = cone.split(s)
        zl, zq, zb = cone.split(z)
        self.wl = np.sqrt(sl / zl)                 # W = diag(wl) on the LP rows
        self.dl = zl / sl                          # W^-2
        if cone.nq3:
            self.eta3, self.wb3 = self._soc(sq, zq)
        if cone.big:
            eb, wb = self._soc(sb[None, :], zb[None, :])
            self.etab, self.wbb = eb[0], wb[0]

    @staticmethod
    def _soc(s, z):
        ns = np.sqrt(np.sum(s[:, 1:] ** 2, axis=1))
        nz = np.sqrt(np.sum(z[:, 1:] ** 2, axis=1))
        a = np.sqrt(_jres(s[:, 0], ns))
        b = np.sqrt(_jres(z[:, 0], nz))
        sb = s / a[:, None]
        zb = z / b[:, None]
        gamma = np.sqrt((1.0 + np.sum(sb * zb, axis=1)) / 2.0)
        wbar = np.empty_like(s)
        wbar[:, 0] = (sb[:, 0] + zb[:, 0]) / (2 * gamma)
        wbar[:, 1:] = (sb[:, 1:] - zb[:, 1:]) / (2 * gamma[:, None])
        eta = np.sqrt(a / b)
        return eta, wbar

    @staticmethod
    def _soc_apply(eta, wbar, u, inverse):
        w0 = wbar[:, 0]
        w1 = wbar[:, 1:]
        u0 = u[:, 0]
        u1 = u[:, 1:]
        dot = np.sum(w1 * u1, axis=1)
        out = np.empty_like(u)
        if inverse:
            out[:, 0] = (w0 * u0 - dot) / eta
            out[:, 1:] = (u1 + ((-u0 + dot / (1 + w0)))[:, None] * w1) / eta[:, None]
        else:
            out[:, 0] = (w0 * u0 + dot) * eta
            out[:, 1:] = (u1 + ((u0 + dot / (1 + w0)))[:, None] * w1) * eta[:, None]
        return out

    def apply(self, v, inverse=False):
        c = self.cone
        out = np.empty_like(v)
        vl, vq, vb = c.split(v)
        out[: c.l] = vl / self.wl if inverse else vl * self.wl
        if c.nq3:
            out[c.o3:c.ob] = self._soc_apply(self.eta3, self.wb3, vq, inverse).ravel()
        if c.big:
            out[c.ob:] = self._soc_apply(np.array([self.etab]), self.wbb[None, :],
                                         vb[None, :], inverse)[0]
        return out

    # ---- eigen form of the second-order-cone blocks ---------------------------------------
    # W^-2 = eta^-2 (2 u u' - J), u = J wbar, has the eigenpairs
    #   lam_p = (w0+|w1|)^2/eta^2  on  e_p = (1, -what)/sqrt2     ("strong" when the cone is nearly active)
    #   lam_0 = 1/eta^2            on  {0} x what^perp
    #   lam_m = 1/(eta^2 (w0+|w1|)^2)  on  e_m = (1, what)/sqrt2
    # Applying it as  sum_k lam_k e_k e_k'  has no cancellation between the three scales.
    @staticmethod
    def _soc_eig(eta, wbar):
        w0 = wbar[:, 0]
        n1 = np.sqrt(np.sum(wbar[:, 1:] ** 2, axis=1))
        what = np.zeros_like(wbar[:, 1:])
        nz = n1 > 0
        what[nz] = wbar[nz, 1:] / n1[nz, None]
        what[~nz, 0] = 1.0
        g = (w0 + n1) ** 2
        ep = np.concatenate([np.ones((len(w0), 1)), -what], axis=1) / np.sqrt(2.0)
        em = np.concatenate([np.ones((len(w0), 1)), what], axis=1) / np.sqrt(2.0)
        return g / eta ** 2, 1.0 / eta ** 2, 1.0 / (eta ** 2 * g), ep, em

    @staticmethod
    def _soc_eig_apply(fp, f0, fm, ep, em, V):
        """sum_k f_k e_k e_k' V for V of shape (ncones, d, ncols)."""
        cp = np.einsum("ka,kan->kn", ep, V)
        cm = np.einsum("ka,kan->kn", em, V)
        rest = V - ep[:, :, None] * cp[:, None, :] - em[:, :, None] * cm[:, None, :]
        return (ep[:, :, None] * (fp[:, None] * cp)[:, None, :] + em[:, :, None] * (fm[:, None] * cm)[:, None, :]
                + f0[:, None, None] * rest)

    def eig_weights(self):
        """(LP weights, (lam_p, lam_0, lam_m, e_p, e_m) of the Q3 cones, the same of the big cone or None)."""
        c = self.cone
        q3 = self._soc_eig(self.eta3, self.wb3) if c.nq3 else None
        bg = self._soc_eig(np.array([self.etab]), self.wbb[None, :]) if c.big else None
        return self.dl, q3, bg

    def eig_apply(self, V, power, cap=None):
        """W^(2*power) V (power = -1: W^-2, +1: W^2) in eigen form; with `cap`, every eigenvalue of W^-2
        above cap is replaced by cap (the capped scaling W_w of the extended-precision solve)."""
        c = self.cone
        vec = V.ndim == 1
        if vec:
            V = V[:, None]
        out = np.empty_like(V)
        dl, q3, bg = self.eig_weights()

        def f(lam):
            lam = np.minimum(lam, cap) if cap is not None else lam
            return lam if power < 0 else 1.0 / lam
        out[: c.l] = f(dl)[:, None] * V[: c.l]
        if c.nq3:
            lp, l0, lm, ep, em = q3
            out[c.o3:c.ob] = self._soc_eig_apply(f(lp), f(l0), f(lm), ep, em,
                                                 V[c.o3:c.ob].reshape(c.nq3, 3, -1)).reshape(3 * c.nq3, -1)
        if c.big:
            lp, l0, lm, ep, em = bg                      # lam_0 has multiplicity big-2: never capped
            out[c.ob:] = self._soc_eig_apply(f(lp), l0 if power < 0 else 1.0 / l0, f(lm), ep, em, V[None, c.ob:])[0]
        return out[:, 0] if vec else out

    def inv2(self, V):
        """W^-2 V for a vector (R,) or a matrix (R, k)."""
        c = self.cone
        vec = V.ndim == 1
        if vec:
            V = V[:, None]
        out = np.empty_like(V)
        out[: c.l] = self.dl[:, None] * V[: c.l]
        if c.nq3:
            Vq = V[c.o3:c.ob].reshape(c.nq3, 3, -1)
            u = self.wb3 * np.array([1.0, -1.0, -1.0])        # J wbar
            uv = np.einsum("ka,kan->kn", u, Vq)
            o = 2 * u[:, :, None] * uv[:, None, :]
            o[:, 0, :] -= Vq[:, 0, :]
            o[:, 1:, :] += Vq[:, 1:, :]
            out[c.o3:c.ob] = (o / (self.eta3 ** 2)[:, None, None]).reshape(3 * c.nq3, -1)
        if c.big:
            Vb = V[c.ob:]
            u = self.wbb.copy()
            u[1:] = -u[1:]
            o = 2 * np.outer(u, u @ Vb)
            o[0] -= Vb[0]
            o[1:] += Vb[1:]
            out[c.ob:] = o / self.etab ** 2
        return out[:, 0] if vec else out


def _soc_prod(u, v):
    out = np.empty_like(u)
    out[:, 0] = np.sum(u * v, axis=1)
    out[:, 1:] = u[:, :1] * v[:, 1:] + v[:, :1] * u[:, 1:]
    return out


def _soc_div(lam, d):
    """x with lam o x = d."""
    l0 = lam[:, 0]
    l1 = lam[:, 1:]
    a = _jres(l0, np.sqrt(np.sum(l1 ** 2, axis=1)))
    ld = np.sum(l1 * d[:, 1:], axis=1)
    out = np.empty_like(d)
    out[:, 0] = (l0 * d[:, 0] - ld) / a
    out[:, 1:] = (d[:, 1:] - out[:, :1] * l1) / l0[:, None]
    return out


def _soc_target(v, mut):
    """Centrality corrector on second-order cones: t such that the eigenvalues v0 +- ||v1|| of v + t lie in
    [CORR_BMIN, CORR_BMAX] * mut (each eigenvalue's move bounded below by -CORR_BMAX mut), in the Jordan frame of v."""
    n1 = np.sqrt(np.sum(v[:, 1:] ** 2, axis=1))
    e1, e2 = v[:, 0] + n1, v[:, 0] - n1
    d1 = np.maximum(np.minimum(np.maximum(e1, CORR_BMIN * mut), CORR_BMAX * mut) - e1, -CORR_BMAX * mut)
    d2 = np.maximum(np.minimum(np.maximum(e2, CORR_BMIN * mut), CORR_BMAX * mut) - e2, -CORR_BMAX * mut)
    out = np.empty_like(v)
    out[:, 0] = 0.5 * (d1 + d2)
    out[:, 1:] = (0.5 * (d1 - d2) / np.where(n1 > 0, n1, 1.0))[:, None] * v[:, 1:]
    return out


def _cone_prod(c, u, v):
    out = np.empty_like(u)
    ul, uq, ub = c.split(u)
    vl, vq, vb = c.split(v)
    out[: c.l] = ul * vl
    if c.nq3:
        out[c.o3:c.ob] = _soc_prod(uq, vq).ravel()
    if c.big:
        out[c.ob:] = _soc_prod(ub[None, :], vb[None, :])[0]
    return out


def _cone_div(c, lam, d):
    out = np.empty_like(d)
    ll, lq, lb = c.split(lam)
    dl, dq, db = c.split(d)
    out[: c.l] = dl / ll
    if c.nq3:
        out[c.o3:c.ob] = _soc_div(lq, dq).ravel()
    if c.big:
        out[c.ob:] = _soc_div(lb[None, :], db[None, :])[0]
    return out


def _soc_step(lam, d):
    """max over cones of (||rho_1|| - rho_0), rho = T(lam) d, T lam = e."""
    l0 = lam[:, 0]
    l1 = lam[:, 1:]
    a = np.sqrt(_jres(l0, np.sqrt(np.sum(l1 ** 2, axis=1))))
    lb0 = l0 / a
    lb1 = l1 / a[:, None]
    dot = np.sum(lb1 * d[:, 1:], axis=1)
    rho0 = (lb0 * d[:, 0] - dot) / a
    rho1 = (d[:, 1:] + ((-d[:, 0] + dot / (1 + lb0)))[:, None] * lb1) / a[:, None]
    return (np.sqrt(np.sum(rho1 ** 2, axis=1)) - rho0).max()


def _max_step(c, lam, d):
    """t such that lam + alpha d in K  <=>  alpha <= 1/t (t<=0: unbounded)."""
    ll, lq, lb = c.split(lam)
    dl, dq, db = c.split(d)
    t = -np.inf
    if c.l:
        t = max(t, (-dl / ll).max())
    if c.nq3:
        t = max(t, _soc_step(lq, dq))
    if c.big:
        t = max(t, _soc_step(lb[None, :], db[None, :]))
    return t


MAX_SWEEPS = 8
REFTOL = 1e-11
REFETA = 1e-1                 # ... or this fraction of the iterate's own dual residual, whichever is larger (see solve())
WALL_ITERS = 3
INACC_FEAS = 1e-6
INACC_GAP = 1.22e-4           # CVX's reduced tolerance eps^(1/4): what 'Inaccurate/Solved' means in the reference
STATUS_OPTIMAL_INACCURATE = 5


PIVTOL = 1e-13
CHOL_NB = 64


def chol_piv(H):
    """Blocked right-looking Cholesky (64-wide panels, the device kernel's schedule) with
    the interior-point pivot rule: a pivot that is not above PIVTOL * H_jj (i.e. pure
    rounding noise) is replaced by H_jj itself.  By Cauchy-Schwarz the rest of that column of
    the Schur complement is then at noise level too, so the column is effectively decoupled
    and the factor stays a non-singular (CG-usable) preconditioner instead of dividing by
    noise.  Returns (L, number of replaced pivots)."""
    N = H.shape[0]
    L = np.tril(H).copy()
    d0 = np.diag(H).copy()
    nfix = 0
    for k0 in range(0, N, CHOL_NB):
        k1 = min(N, k0 + CHOL_NB)
        D = L[k0:k1, k0:k1]
        for j in range(k1 - k0):
            p = D[j, j]
            if not (p > PIVTOL * d0[k0 + j]):
                p = max(d0[k0 + j], 1e-300)
                nfix += 1
            r = np.sqrt(p)
            D[j, j] = r
            D[j + 1:, j] /= r
            D[j + 1:, j + 1:] -= np.tril(np.outer(D[j + 1:, j], D[j + 1:, j]))
        if k1 < N:
            # panel by forward substitution X D' = A (multiplying by inv(D) instead is not
            # backward stable and breaks the factorisation on the near-singular late iterates);
            # numpy only: mixing numpy's and scipy's OpenBLAS thread pools costs ~30 ms per switch
            X = L[k1:, k0:k1]
            for j in range(k1 - k0):
                X[:, j] = (X[:, j] - X[:, :j] @ D[j, :j]) / D[j, j]
            L[k1:, k1:] -= np.tril(L[k1:, k0:k1] @ L[k1:, k0:k1].T)
    return L, nfix


def next_sweeps(norm_lists, nsweep, tol):
    """Refinement controller shared with the device solver.  norm_lists: for every KKT
    solve of the iteration the dual-equation residual norms [n_0 .. n_nsweep] (n_0 after the
    Cholesky solve, n_k after CG iteration k).  If every solve reached `tol` after k
    iterations the next IPM iteration runs k of them (possibly none), otherwise one more
    (at most MAX_SWEEPS)."""
    need = 0
    for norms in norm_lists:
        k = next((i for i, v in enumerate(norms) if v <= tol), None)
        if k is None:
            return min(MAX_SWEEPS, nsweep + 1)
        need = max(need, k)
    return need


def solve(c, G, h, l, nq3=0, big=0, max_iter=200, feastol=1e-8, abstol=1e-10,
          reltol=1e-8, refine=2, verbose=False, history=None, ddkkt=None, start=None, corrector=True):
    """Returns dict(status, x, s, z, iters, pcost, dcost, gap, pres, dres).

    Stopping rule (all quantities of the de-homogenised point x/tau ...):
      pres = ||Gx+s-h||/max(1,||h||), dres = ||G'z+c||/max(1,||c||),
      gap  = s'z, relgap = gap/max(|pcost|,|dcost|)  -> optimal when
      pres,dres <= feastol and (gap <= abstol or relgap <= reltol).
    Certificates: primal infeasible when h'z<0 and ||G'z||/(-h'z) <= feastol;
    dual infeasible when c'x<0 and ||Gx+s||/(-c'x) <= feastol.
    """
    c = np.asarray(c, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    G = np.asarray(G, dtype=np.float64)
    cone = _Cone(l, nq3, big)
    R, N = G.shape
    assert R == cone.R
    nrm_h = max(1.0, np.linalg.norm(h))
    nrm_c = max(1.0, np.linalg.norm(c))
    e = cone.e()

    def factor(Wm):
        if ddkkt is not None and Wm is not None:
            st = factor_dd(Wm)
            if st is not None:
                return None, st
        H = G.T @ (Wm.inv2(G) if Wm is not None else G)
        H = 0.5 * (H + H.T)
        if not np.all(np.isfinite(H)):
            raise FloatingPointError("non-finite normal matrix")
        L, nfix = chol_piv(H)
        chol_fixes[0] += nfix
        return H, np.linalg.inv(L)          # M = L^-1, as the device solver keeps it

    sweep_log = []
    chol_fixes = [0]

    def cho_solve(M, b):
        return M.T @ (M @ b)

    # ---- extended-precision KKT solve (ddkkt = dict(theta=..., nref=...)) ----------------------
    # Every eigenvalue of W^-2 above cap = theta * (median weight) is split  lam = cap + excess.  The capped
    # scaling W_w goes through the ordinary double-precision Gram product; the excess parts
    # H_s = U' X U  (U: one row G'e per strong eigen-direction, X = diag(excess)) are accumulated, and
    # H = H_w + H_s is factorised, in double-double arithmetic (oracle/ddlin.c), so the weakly weighted
    # directions survive next to weights 1e16 times larger.  The multipliers of the strong directions,
    # zeta = X (U dx - e_p'bz), are evaluated in double-double from the double-double dx.  Around this
    # solver runs plain iterative refinement on the augmented system [0 G'; G -W^2] in double precision,
    # whose residuals involve W^2 (tiny on the strong directions), never W^-2.
    def factor_dd(Wm):
        from . import ddlin
        dl, q3, bg = Wm.eig_weights()
        typ = [dl] + ([q3[1]] if q3 is not None else []) + ([bg[1]] if bg is not None else [])
        # "typical" weight: geometric mean of the LP weights and the middle eigenvalues of the cones (a sum
        # reduction on the device, where a median would need a selection)
        cap = ddkkt.get("theta", 1e6) * float(np.exp(np.mean(np.log(np.concatenate(typ)))))
        # strong eigen-directions: (offset of the cone in R, eigenvector over the cone's rows, excess weight)
        rowsU, X, sel = [], [], []
        if cone.l:
            i = np.nonzero(dl > cap)[0]
            sel.append(("l", i, None))
            rowsU.append(G[i]); X.append(dl[i] - cap)
        if cone.nq3:
            lp, l0, lm, ep, em = q3
            e0 = np.stack([np.zeros(cone.nq3), -ep[:, 2], ep[:, 1]], 1) * np.sqrt(2.0)   # (0, what_perp)
            G3 = G[cone.o3:cone.ob].reshape(cone.nq3, 3, N)
            for lam, ev in ((lp, ep), (l0, e0), (lm, em)):
                i = np.nonzero(lam > cap)[0]
                sel.append(("q", i, ev[i]))
                rowsU.append(np.einsum("ka,kan->kn", ev[i], G3[i])); X.append(lam[i] - cap)
        if cone.big:
            for lam, ev in ((bg[0], bg[3]), (bg[2], bg[4])):
                i = np.nonzero(lam > cap)[0]
                sel.append(("b", i, ev[i]))
                rowsU.append(ev[i] @ G[cone.ob:]); X.append(lam[i] - cap)
        U = np.ascontiguousarray(np.concatenate(rowsU, 0))
        X = np.ascontiguousarray(np.concatenate(X))
        if len(X) == 0:
            return None                                        # nothing above the cap: the plain solve is exact enough
        Hw = G.T @ Wm.eig_apply(G, -1, cap)
        Hw = 0.5 * (Hw + Hw.T)
        if not np.all(np.isfinite(Hw)):
            raise FloatingPointError("non-finite normal matrix")
        if ddkkt.get("form") == "cap":
            # CAPACITANCE (saddle-point) form, plain double precision (round 4): the strong directions are kept as nearly-equality
            # constraints instead of being folded into the matrix --
            #     [H_w  U'; U  -X^-1] [dx; zeta] = [rhs_w; t],   zeta = X (U dx - t)
            # -- so no number of the size of X ever meets one of the size of H_w: H_w = L L' in double, Y = L^-1 U' (N x k),
            # S = X^-1 + Y'Y (k x k: a Cholesky of k <= 588 instead of a double-double one of np = 1088), zeta = S^-1 (U y - t),
            # dx = H_w^-1 (rhs_w - U' zeta).  The refinement on the augmented system around it is unchanged.
            Lw, nfix = chol_piv(Hw)
            chol_fixes[0] += nfix
            Mw = np.linalg.inv(Lw)
            Y = Mw @ U.T                                        # N x k
            S = Y.T @ Y + np.diag(1.0 / X)
            S = 0.5 * (S + S.T)
            Ls, nfs = chol_piv(S)
            chol_fixes[0] += nfs
            st = dict(Wm=Wm, cap=cap, U=U, X=X, k=len(X), nfix=nfix + nfs, Mw=Mw, Ms=np.linalg.inv(Ls), Y=Y, form="cap")
        else:
            st = None
        Hh = np.ascontiguousarray(Hw)
        Hl = np.zeros_like(Hh)
        if len(X) and st is None:
            ddlin.rank_k(U, X, Hh, Hl)
        if st is None:
            d0 = np.ascontiguousarray(np.diag(Hh)).copy()
            nfix = ddlin.chol(Hh, Hl, ddkkt.get("pivtol", 1e-28), d0)
            chol_fixes[0] += nfix

        def comp(V):                                           # e' V_cone for the strong directions
            out = []
            for kind, i, ev in sel:
                if kind == "l":
                    out.append(V[i])
                elif kind == "q":
                    out.append(np.einsum("ka,kan->kn", ev, V[cone.o3:cone.ob].reshape(cone.nq3, 3, -1)[i]))
                else:
                    out.append(ev @ V[cone.ob:])
            return np.concatenate(out, 0)

        def spread(Z, ncol):                                   # rows of R from strong-direction multipliers
            o = np.zeros((R, ncol))
            oq = o[cone.o3:cone.ob].reshape(cone.nq3, 3, ncol)
            k0 = 0
            for kind, i, ev in sel:
                k1 = k0 + len(i)
                if kind == "l":
                    o[i] += Z[k0:k1]
                elif kind == "q":
                    np.add.at(oq, i, ev[:, :, None] * Z[k0:k1][:, None, :])
                elif len(i):
                    o[cone.ob:] += ev[0][:, None] * Z[k0][None, :]
                k0 = k1
            return o
        if st is not None:
            st.update(comp=comp, spread=spread)
            return st
        return dict(Wm=Wm, cap=cap, U=U, X=X, Lh=Hh, Ll=Hl, comp=comp, spread=spread, k=len(X), nfix=nfix)

    dd_log = []

    def kkt_solve_dd(st, bx, bz):
        from . import ddlin
        Wm, U, X, cap = st["Wm"], st["U"], st["X"], st["cap"]
        vec = bx.ndim == 1
        BX = bx[:, None].copy() if vec else bx
        BZ = bz[:, None].copy() if vec else bz
        ncol = BX.shape[1]
        DX = np.zeros((N, ncol)); DZ = np.zeros((R, ncol)); GDX = np.zeros((R, ncol))
        norms = []
        nref = ddkkt.get("nref", 3 if st.get("form") == "cap" else 2)       # the capacitance form takes a third pass (solver.hip: why)
        for it in range(nref + 1):
            r1 = BX - G.T @ DZ
            r2 = BZ - GDX + Wm.eig_apply(DZ, +1)
            n1 = float(np.max(np.sqrt(np.sum(r1 * r1, axis=0))))
            n2 = float(np.max(np.abs(r2)))
            norms.append((n1, n2))
            if it == nref:
                break
            t = np.ascontiguousarray(st["comp"](r2))                         # k x ncol
            if st.get("form") == "cap":
                rw = r1 + G.T @ Wm.eig_apply(r2, -1, cap)                    # rhs_w: the capped scaling only
                Mw, Ms, Y = st["Mw"], st["Ms"], st["Y"]
                v = Mw @ rw                                                  # L^-1 rhs_w
                zeta = Ms.T @ (Ms @ (Y.T @ v - t))                           # S^-1 (U H_w^-1 rhs_w - t)
                dxh = Mw.T @ (v - Y @ zeta)                                  # H_w^-1 (rhs_w - U' zeta)
                Gd = G @ dxh
                dz = Wm.eig_apply(Gd - r2, -1, cap) + st["spread"](zeta, ncol)
                DX += dxh; DZ += dz; GDX += Gd
                continue
            rh = np.ascontiguousarray(r1 + G.T @ Wm.eig_apply(r2, -1, cap))
            rl = np.zeros_like(rh)
            if st["k"]:
                yh, yl = ddlin.vec_mul_d(t.ravel(), np.zeros(t.size), np.repeat(X, ncol))
                ddlin.cols_times_acc(U, yh.reshape(-1, ncol), yl.reshape(-1, ncol), rh, rl)
            ddlin.cho_solve(st["Lh"], st["Ll"], rh, rl)                      # rh, rl <- delta x (dd)
            dxh = rh
            Gd = G @ dxh
            dz = Wm.eig_apply(Gd - r2, -1, cap)
            if st["k"]:
                uh, ul = ddlin.rows_times(U, rh, rl)
                uh, ul = ddlin.vec_sub(uh.ravel(), ul.ravel(), t.ravel(), np.zeros(t.size))
                zh, zl = ddlin.vec_mul_d(uh, ul, np.repeat(X, ncol))
                dz = dz + st["spread"]((zh + zl).reshape(-1, ncol), ncol)
            DX += dxh; DZ += dz; GDX += Gd
        dd_log.append(norms)
        return (DX[:, 0], DZ[:, 0], GDX[:, 0]) if vec else (DX, DZ, GDX)

    def kkt_solve(Wm, H, cf, bx, bz, plain=False):
        """[0 G'; G -W^2][dx; dz] = [bx; bz] for one or two right-hand sides (columns);
        returns (dx, dz, G dx).  plain: the Cholesky solve alone -- no refinement sweeps, nothing for
        the sweep controller -- and, fourth, the norm of its dual-equation residual ||bx - G'dz|| (the
        centrality corrector's solve: the caller drops a correction whose residual would spoil the
        iterate's own).

        dz is kept as an explicit vector and corrected incrementally, so the dual
        equation G'dz = bx is driven to rounding level even when ||H|| eps is large
        (the residual of an increment scales with the increment, not with dx)."""
        if isinstance(cf, dict):
            return kkt_solve_dd(cf, bx, bz)
        wbz = Wm.inv2(bz) if Wm is not None else bz
        rhs = bx + G.T @ wbz
        dx = cho_solve(cf, rhs)
        Gdx = G @ dx
        dz = (Wm.inv2(Gdx) if Wm is not None else Gdx) - wbz
        if plain:
            return dx, dz, Gdx, float(np.linalg.norm(bx - G.T @ dz))
        # Preconditioned conjugate gradients on (G' W^-2 G) dx = rhs with the operator applied
        # exactly through G (two passes over the rows) and M'M as preconditioner; dz and G dx
        # are carried along, so the dual equation G'dz = bx ends at the CG residual.
        vec = bx.ndim == 1
        col = (lambda v: v[:, None]) if vec else (lambda v: v)
        r = col(bx - G.T @ dz).copy()
        DX, DZ, GDX = col(dx).copy(), col(dz).copy(), col(Gdx).copy()
        z_ = cho_solve(cf, r)
        p = z_.copy()
        rz_ = np.sum(r * z_, axis=0)
        norms = [float(np.max(np.sqrt(np.sum(r * r, axis=0))))]
        for _ in range(nsweep[0]):
            Gp = G @ p
            Wp = Wm.inv2(Gp) if Wm is not None else Gp
            Hp = G.T @ Wp
            pHp = np.sum(p * Hp, axis=0)
            alpha_ = np.where(pHp > 0, rz_ / np.where(pHp > 0, pHp, 1.0), 0.0)
            DX += alpha_ * p
            GDX += alpha_ * Gp
            DZ += alpha_ * Wp
            r -= alpha_ * Hp
            norms.append(float(np.max(np.sqrt(np.sum(r * r, axis=0)))))
            z_ = cho_solve(cf, r)
            rz_new = np.sum(r * z_, axis=0)
            beta_ = np.where(rz_ > 0, rz_new / np.where(rz_ > 0, rz_, 1.0), 0.0)
            p = z_ + beta_ * p
            rz_ = rz_new
        dx, dz, Gdx = (DX[:, 0], DZ[:, 0], GDX[:, 0]) if vec else (DX, DZ, GDX)
        sweep_log.append(norms)
        return dx, dz, Gdx

    # ---- initial point (W = I) ------------------------------------------
    nsweep = [int(refine)]
    nsweep_ctl = int(refine)   # the sweep controller's count; nsweep[0] = what the solves of the iteration run (see POLISH_SWEEPS)
    H, cf = factor(None)
    X0, Z0, _ = kkt_solve(None, H, cf, np.stack([np.zeros(N), -c], 1), np.stack([h, np.zeros(R)], 1))
    x = X0[:, 0]                                            # min ||Gx-h||
    s = -Z0[:, 0]                                           # h - Gx
    ts = cone.min_residual(s)
    if ts >= -1e-8 * max(1.0, np.linalg.norm(s)):
        s = s + (1.0 + ts) * e
    z = Z0[:, 1]                                            # G'z=-c, least norm
    tz = cone.min_residual(z)
    if tz >= -1e-8 * max(1.0, np.linalg.norm(z)):
        z = z + (1.0 + tz) * e
    tau, kappa = 1.0, 1.0
    if start is not None:
        # STUDY ONLY (tools/exp/warmstart_study.py): a warm start of the homogeneous embedding in the manner
        # of Skajaa, Andersen & Ye -- the convex combination of a neighbouring problem's optimum (x*, s*, z*) with a cold,
        # well-centred point: start = dict(x, s, z, lam[, cold]).  cold = "e": (0, e, e, 1, 1) as in their paper; "own": this
        # solver's own initial point computed above.  The device takes no starting point (measured, not adopted).
        lam_ws = float(start["lam"])
        if start.get("cold", "e") == "e":
            xc, sc, zc = np.zeros(N), e.copy(), e.copy()
        else:
            xc, sc, zc = x, s, z
        x = lam_ws * np.asarray(start["x"], dtype=np.float64) + (1 - lam_ws) * xc
        s = lam_ws * np.asarray(start["s"], dtype=np.float64) + (1 - lam_ws) * sc
        z = lam_ws * np.asarray(start["z"], dtype=np.float64) + (1 - lam_ws) * zc
        tau = 1.0
        kappa = float(s @ z) / (cone.degree + 1)
    status = STATUS_MAXIT
    it = 0
    info = {}
    best = (np.inf, None, None)
    wall = 0                  # consecutive iterations past the numerical wall (see below)
    ncorr = [0, 0]            # corrector solves, corrected directions taken
    opt_best = (None, None, None, None, None)      # end game: the best iterate that met the stopping rule
    first_opt = None          # ... and the iteration of the first one
    fixes_seen = 0
    for it in range(max_iter + 1):
        rx = G.T @ z + c * tau
        rz = G @ x + s - h * tau
        cx, hz = c @ x, h @ z
        rt = kappa + cx + hz
        sz = s @ z
        mu = (sz + kappa * tau) / (cone.degree + 1)
        pcost, dcost = cx / tau, -hz / tau
        gap = sz / tau ** 2
        pres = np.linalg.norm(rz) / tau / nrm_h
        dres = np.linalg.norm(rx) / tau / nrm_c
        den = max(abs(pcost), abs(dcost))
        relgap = gap / den if den > 0 else np.inf
        hresx = np.linalg.norm(rx - c * tau)                # ||G'z||
        hresz = np.linalg.norm(rz + h * tau)                # ||Gx+s||
        pinfres = hresx / (-hz) if hz < 0 else np.inf
        dinfres = hresz / (-cx) if cx < 0 else np.inf
        info = dict(iters=it, pcost=pcost, dcost=dcost, gap=gap, relgap=relgap,
                    pres=pres, dres=dres, tau=tau, kappa=kappa, mu=mu,
                    pinfres=pinfres, dinfres=dinfres)
        if history is not None:
            history.append(dict(info))
        if verbose:
            print("%3d pcost % .10e dcost % .10e gap %.2e pres %.1e dres %.1e k/t %.1e mu %.1e"
                  % (it, pcost, dcost, gap, pres, dres, kappa / tau, mu),
                  " ".join("[%s]" % " ".join("%.0e/%.0e" % t for t in nn) for nn in dd_log[-2:]) if dd_log and verbose > 1 else "")
        finite = np.isfinite(pres) and np.isfinite(dres) and np.isfinite(gap) and tau > 0
        if finite and pres <= feastol and dres <= feastol and (gap <= abstol or relgap <= reltol):
            # End game.  The iterate meets the stopping rule; its distance to the optimum is of the order of its gap, so two
            # solvers that stop one iteration apart -- or the same solver on another machine: the iteration amplifies rounding
            # differences, most of all with the centrality corrector -- return points 1e-8 apart, and fir_ap_cvx's spectral
            # factorisation amplifies that by 1e1 ... 1e6 in the taps.  The last iterations converge fast (step 0.9-0.99), so the
            # iteration goes on while it pays: until the gap measures are POLISH times below the tolerances, for at most
            # POLISH_MAX more iterations, and the BEST iterate that met the rule is returned (the last one unless rounding
            # has the final word).  Two or three iterations for an answer that no longer depends on the path.
            merit_o = min(relgap / reltol, gap / max(abstol, 1e-300))
            if opt_best[0] is None or merit_o < opt_best[0]:
                opt_best = (merit_o, x / tau, dict(info), s / tau, z / tau)
            if first_opt is None:
                first_opt = it
            if gap <= POLISH * abstol or relgap <= POLISH * reltol:
                status = STATUS_OPTIMAL
                break
        if first_opt is not None and it >= first_opt + POLISH_MAX:
            # (also when this iterate no longer meets the rule: the end game has had its iterations)
            status = STATUS_OPTIMAL
            break
        # The solves of the final approach (a gap measure within POLISH_APPROACH of its tolerance) and of the end game are the worst
        # conditioned of the whole iteration, mu falls by up to 100 x per iteration there (sigma ~ 1e-4), and the sweep controller
        # answers one iteration late: POLISH_SWEEPS more sweeps than it asks for, so that an iterate does not miss the stopping
        # rule by its dual residual when its gap is already there (fuzz seed 55, fir_qprog_phs: dres 5.7e-10 -> 1.7e-8 behind the
        # first iterate that met the rule; a 29-tap fir_qprog_phs: 1.1e-9 -> 9.5e-8 one iteration BEFORE, then the numerical wall).
        # Measured on the device, one box (tools/gpu_polish_ab2.py): no extra sweeps 305.6 designs/s on the headline batch and 17 of
        # 600 fuzz designs retried in extended precision; + 2 from a factor 1000 on: 271.1 and none; + 1 from a factor 30 on: 298.6, none.
        approach = finite and (gap <= POLISH_APPROACH * abstol or relgap <= POLISH_APPROACH * reltol)
        nsweep[0] = min(MAX_SWEEPS, nsweep_ctl + (POLISH_SWEEPS if (approach or first_opt is not None) else 0))
        if not finite:
            status = STATUS_NUMERICAL
            break
        # Farkas certificates; when tau has collapsed relative to kappa a looser
        # certificate is accepted (the dual residual has a rounding floor ~1e-10)
        collapsed = kappa / tau >= 1e6
        if first_opt is None and (pinfres <= feastol or (collapsed and pinfres <= 1e-5)):
            status = STATUS_PRIMAL_INFEASIBLE
            break
        if first_opt is None and (dinfres <= feastol or (collapsed and dinfres <= 1e-5)):
            status = STATUS_DUAL_INFEASIBLE
            break
        # best iterate so far for the reduced-accuracy exit: among the iterates whose residuals
        # meet the reduced feasibility tolerance, the one with the smallest gap measure
        if pres <= INACC_FEAS and dres <= INACC_FEAS:
            merit = min(relgap, gap / max(abstol, 1e-300) * reltol)
            if merit < best[0]:
                best = (merit, x / tau, dict(info))
        if it == max_iter:
            break
        # numerical wall: the last factorisation had to replace pivots AND the residuals are out of the
        # reduced-accuracy range -- three such iterations in a row and the normal matrix has lost the weak
        # directions for good; stop instead of iterating on noise to max_iter
        last_fixes = chol_fixes[0] - fixes_seen
        fixes_seen = chol_fixes[0]
        wall = wall + 1 if (last_fixes > 0 and (pres > INACC_FEAS or dres > INACC_FEAS)) else 0
        if wall >= WALL_ITERS:
            status = STATUS_NUMERICAL
            break
        try:
            Wm = _Scaling(cone, s, z)
            lam = Wm.apply(z)
            H, cf = factor(Wm)
            sweep_log.clear()
            # constant system [x1 z1] and affine system in one two-column solve
            XB, ZB, GB = kkt_solve(Wm, H, cf, np.stack([-c, -rx], 1), np.stack([h, s - rz], 1))
        except FloatingPointError:
            status = STATUS_NUMERICAL
            break
        x1, z1, Gx1 = XB[:, 0], ZB[:, 0], GB[:, 0]
        if not (np.all(np.isfinite(XB)) and np.all(np.isfinite(ZB))):
            status = STATUS_NUMERICAL
            break
        wz1 = Wm.apply(z1)
        den_t = kappa / tau + wz1 @ wz1

        def direction(sigma, dk_c, x2, z2, Gx2):
            bt = -(1 - sigma) * rt
            dtau = (dk_c / tau - bt + c @ x2 + h @ z2) / den_t
            dx = x2 + dtau * x1
            dz = z2 + dtau * z1
            # ds from the primal equation G dx + ds - h dtau = -(1-sigma) rz: the primal
            # residual then shrinks by exactly (1 - alpha (1-sigma)) per step and the
            # rounding of the scaled solve lands in the complementarity row, where it
            # is harmless.
            ds = -(1 - sigma) * rz - Gx2 - dtau * (Gx1 - h)
            wdz = Wm.apply(dz)
            dss = Wm.apply(ds, inverse=True)                  # W^-1 ds
            dkap = (dk_c - kappa * dtau) / tau
            return dx, ds, dz, dtau, dkap, dss, wdz

        def step_of(dss, wdz, dtau, dkap, frac):
            t = max(0.0, _max_step(cone, lam, dss), _max_step(cone, lam, wdz),
                    -dtau / tau, -dkap / kappa)
            return 1.0 if t == 0.0 else min(1.0, frac / t)

        ll = _cone_prod(cone, lam, lam)
        # affine direction: lam \ (-lam o lam) = -lam, W lam = s  ->  bz = s - rz (in the batch above)
        dxa, dsa, dza, dta, dka, dssa, wdza = direction(0.0, -kappa * tau, XB[:, 1], ZB[:, 1], GB[:, 1])
        alpha_a = step_of(dssa, wdza, dta, dka, 1.0)
        # (the cap: 0.25 keeps Mehrotra's rule from over-centring -- measured in round 2 --; where the centrality corrector below looks
        #  after the outliers itself, the direction can aim lower still: 0.05.  Nine S-C13 instances of 58 ... 200 taps in this oracle:
        #  417 iterations at 0.25, 384 at 0.1, 365 at 0.05, 362 at 0.02, 346 at 0 -- and 433 ... 465 at 0.4 ... 1.0; the device's
        #  headline batch 49.2 -> 41.6 iterations per design, no verdict of 800 fuzz specs changed: DESIGN.md section 5a)
        # (not for a program with a big cone, fir_qprog_phs: its quadratic objective is flat around the minimiser, the faster end of
        #  the iteration left device and oracle 1 ... 3e-6 apart in the taps on 9 of its 600 fuzz specs instead of 1)
        sigma = min((1.0 - alpha_a) ** 3, SIGMA_MAX_CORR if (corrector and cone.l > 0 and cone.big == 0) else SIGMA_MAX)
        ds_c = sigma * mu * e - ll - _cone_prod(cone, dssa, wdza)
        dk_c = sigma * mu - kappa * tau - dka * dta
        lds = _cone_div(cone, lam, ds_c)
        try:
            x2, z2, Gx2 = kkt_solve(Wm, H, cf, -(1 - sigma) * rx, -(1 - sigma) * rz - Wm.apply(lds))
        except FloatingPointError:
            status = STATUS_NUMERICAL
            break
        dx, ds, dz, dtau, dkap, dss, wdz = direction(sigma, dk_c, x2, z2, Gx2)
        alpha = step_of(dss, wdz, dtau, dkap, STEP)
        ncorr_ok = 0
        # (the big cone's products are corrected along with the orthant rows', on the plain path: see below)
        corr_big = cone.big > 0 and cone.l > 0 and not isinstance(cf, dict)
        if corrector and cone.l > 0:
            # Centrality corrector (round 6).  The fir_ap_cvx programs keep thousands of rows S(w_i) >= 1e-20 whose products
            # s_i z_i are spread over five decades around mu; the predictor-corrector direction is then cut at alpha = 0.01-0.3
            # by a handful of outliers.  At the trial step at = min(1, alpha + CORR_DELTA) the products of the orthant rows
            # v = (lam + at dss)(lam + at wdz) are projected onto the box [CORR_BMIN, CORR_BMAX] * sigma mu; the difference
            # (bounded below by -CORR_BMAX sigma mu) is one more complementarity right-hand side for the factorisation at hand:
            # [0 G'; G -W^2][dxk; dzk] = [0; -W (lam \ t)].  The corrected direction is taken when its step is longer by
            # CORR_ACCEPT.  The 3-row cones and the (tau, kappa) pair are left alone.  Measured over eight S-C13
            # instances (n = 80 ... 200): 380 -> 301 iterations (-21 %) for one more Cholesky solve per iteration -- without
            # refinement sweeps: with them the count is 303; with the residual guard below 317 of 392 (both with the end game;
            # DESIGN.md section 5).
            # The big cone (the quadratic objective's epigraph in fir_qprog_phs) gets the same projection for the two eigenvalues
            # v0 +- ||v1|| of the Jordan product (lam + at dss) o (lam + at wdz) (_soc_target): fir_qprog_phs 25 -> 19, 28 -> 21
            # iterations.  Not on the extended-precision path, and not in a program WITHOUT orthant rows (fir_qp_cvx: its
            # frequency rows are 3-row cones): correcting all its cones saves a quarter of its iterations (H-1 dual band,
            # n = 256 / 384 / 512: 55 -> 41, 82 -> 59, 79 -> 68) -- measured, built on the device and taken back: the optimum of
            # E + obj Peak is flat, two paths that part end 1e-5 ... 3e-4 apart in the taps at 1e-10 in the objective, and the
            # comparisons that hold device, oracle, shards and the two extended-precision forms together at 1e-6 ... 1e-4 rely
            # on all of them walking ONE path.  fir_ap_cvx's 3-row cones are its n spike constraints
            # beside 6 n ... 60 n orthant rows: correcting them changes nothing (14 instances: 536 against 533 iterations).
            at = min(1.0, alpha + CORR_DELTA)
            mut = sigma * mu
            ll_ = cone.l
            v = (lam[:ll_] + at * dss[:ll_]) * (lam[:ll_] + at * wdz[:ll_])
            tt = np.minimum(np.maximum(v, CORR_BMIN * mut), CORR_BMAX * mut) - v
            tt = np.maximum(tt, -CORR_BMAX * mut)
            bzk = np.zeros(R)
            bzk[:ll_] = -Wm.wl * (tt / lam[:ll_])
            if corr_big:
                ub, wb = (lam + at * dss)[cone.ob:], (lam + at * wdz)[cone.ob:]
                tc = np.zeros(R)
                tc[cone.ob:] = _soc_target(_soc_prod(ub[None, :], wb[None, :]), mut)[0]
                bzk[cone.ob:] = -Wm.apply(_cone_div(cone, lam, tc))[cone.ob:]
            try:
                rk = kkt_solve(Wm, H, cf, np.zeros(N), bzk, plain=True)
            except FloatingPointError:
                status = STATUS_NUMERICAL
                break
            xk, zk, Gxk = rk[:3]
            # (the corrector's solve runs without refinement sweeps: a third of a full solve.  What it leaves of the dual equation,
            #  ||G'dzk||, goes into the next iterate's dual residual; a correction that would put more there than the iterate's own
            #  ||rx|| -- late iterations, when the factorisation alone is no longer accurate -- is dropped.  The extended-precision
            #  solve returns three values: its correction is refined like every other solve)
            nk = rk[3] if len(rk) > 3 else 0.0
            cand = direction(sigma, dk_c, x2 + xk, z2 + zk, Gx2 + Gxk)
            alpha_c = step_of(cand[5], cand[6], cand[3], cand[4], STEP)
            if nk <= max(REFTOL * nrm_c, CORR_ETA * float(np.linalg.norm(rx))) and alpha_c >= CORR_ACCEPT * alpha:
                dx, ds, dz, dtau, dkap, dss, wdz = cand
                alpha = alpha_c
                ncorr_ok = 1
            ncorr[0] += 1
            ncorr[1] += ncorr_ok
        if sweep_log:                                         # (iterations on the extended-precision path keep the count)
            # inexact-Newton forcing term: while the iterate's own dual residual ||rx|| is large there is no point in
            # driving the linear system's residual twelve digits below it -- the controller asks for REFETA * ||rx||
            # or the absolute floor, whichever is larger.  Same iterates at the end (the floor rules once ||rx|| is
            # small), 30-40 % fewer refinement sweeps over a solve.
            nsweep_ctl = next_sweeps(sweep_log, nsweep[0], max(REFTOL * nrm_c, REFETA * float(np.linalg.norm(rx))))
        if history is not None:
            sl_, zl_ = s[:cone.l], z[:cone.l]
            _ts, _tz = _max_step(cone, lam, dss), _max_step(cone, lam, wdz)
            history[-1].update(alpha_s=min(1.0, 0.99 / _ts) if _ts > 0 else 1.0, alpha_z=min(1.0, 0.99 / _tz) if _tz > 0 else 1.0)
            history[-1].update(alpha=alpha, alpha_a=alpha_a, sigma=sigma,
                               cmin=(sl_ * zl_).min() / mu if cone.l else 1.0,
                               cmax=(sl_ * zl_).max() / mu if cone.l else 1.0)
        x = x + alpha * dx
        s = s + alpha * ds
        z = z + alpha * dz
        tau = tau + alpha * dtau
        kappa = kappa + alpha * dkap
        if not (np.isfinite(tau) and tau > 0 and np.all(np.isfinite(x))):
            status = STATUS_NUMERICAL
            break
    out = dict(status=status, x=x / tau, s=s / tau, z=z / tau, chol_fixes=chol_fixes[0], correctors=ncorr[0], correctors_taken=ncorr[1])
    out.update(info)
    if opt_best[0] is not None:
        # an iterate met the stopping rule: the best of them is the answer, however the end game ended (its target, its
        # iteration cap, max_iter, the numerical wall, a non-finite iterate)
        out.update(opt_best[2])
        out.update(status=STATUS_OPTIMAL, x=opt_best[1], s=opt_best[3], z=opt_best[4], iters=it)
        return out
    if status in (STATUS_MAXIT, STATUS_NUMERICAL) and best[1] is not None:
        bi = best[2]
        # the reference accepts CVX's 'Inaccurate/Solved' (fir_ap_cvx.m:176): reduced tolerances
        if bi["pres"] <= INACC_FEAS and bi["dres"] <= INACC_FEAS and (bi["relgap"] <= INACC_GAP or bi["gap"] <= abstol):
            out.update(bi)
            out["status"] = STATUS_OPTIMAL_INACCURATE
            out["x"] = best[1]
    return out
