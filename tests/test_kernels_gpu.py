"""GPU tests of the individual HIP kernels through the C ABI test hooks (include/mbfir.h)."""
import numpy as np
import pytest
from conftest import relinf

import mbfir
from oracle import specfact

pytestmark = pytest.mark.gpu


# (the last one: 23 x 24 / 2 = 276 tiles -- above 256 tiles the plan takes the split of the frequency rows whose workgroups fill their
#  last round of slots best instead of one round, gram.hip gram_plan: what BASELINE config 5's 528 tiles run with)
@pytest.mark.parametrize("m,nt,nw", [(7, 5, 1), (100, 37, 1), (1000, 200, 3), (5000, 399, 1), (3001, 129, 3), (20000, 260, 1), (700, 2900, 1)])
def test_gram_matches_numpy(m, nt, nw):
    """K2: T_w = A' diag(d_w) A on the fp64 matrix cores; fp64 reference A.T*d @ A, tol 1e-13."""
    rng = np.random.default_rng(m + nt)
    A = rng.standard_normal((m, nt))
    d = rng.random((nw, m)) * 10.0 ** rng.integers(-6, 6, size=(nw, m))      # wide dynamic range like z/s
    T = mbfir.test_gram(A, d)
    ref = np.stack([(A.T * d[w]) @ A for w in range(nw)])
    assert relinf(T, ref) <= 1e-13
    assert np.abs(T - np.transpose(T, (0, 2, 1))).max() == 0.0              # exactly symmetric


def test_gram_of_trig_rows_is_toeplitz_plus_hankel():
    """Structural oracle for K2: cos/cos block = (C(j-k)+C(j+k))/2 computed from 2N moments."""
    n, m = 40, 2500
    w = np.sort(np.random.default_rng(5).uniform(-np.pi, np.pi, m))
    j = np.arange(n)
    A = np.cos(np.outer(w, j))
    d = np.random.default_rng(6).random(m) + 0.01
    T = mbfir.test_gram(A, d)[0]
    C = np.array([np.sum(d * np.cos(l * w)) for l in range(2 * n)])
    ref = 0.5 * (C[np.abs(j[:, None] - j[None, :])] + C[j[:, None] + j[None, :]])
    assert relinf(T, ref) <= 1e-12


@pytest.mark.parametrize("n", [1, 50, 64, 65, 200, 449, 1023])
def test_cholesky_and_inverse(n):
    """K4: H = L L', M = L^-1 (blocked Cholesky + recursive-doubling inverse)."""
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n + 20, n))
    H = B.T @ B + 0.1 * np.eye(n)
    L, M = mbfir.test_chol(H)
    Lr = np.linalg.cholesky(H)
    assert relinf(L, Lr) <= 1e-12
    assert np.abs(np.triu(L, 1)).max() == 0 and np.abs(np.triu(M, 1)).max() == 0
    assert np.abs(M @ Lr - np.eye(n)).max() <= 1e-11
    b = rng.standard_normal(n)
    x = M.T @ (M @ b)
    assert np.linalg.norm(H @ x - b) <= 1e-10 * np.linalg.norm(b)


@pytest.mark.parametrize("n,k", [(40, 0), (100, 7), (257, 60), (700, 300)])
def test_double_double_solve_matches_the_oracles_dd_kernels(n, k):
    """Extended-precision KKT kernels (ddlin.hip): H = H_w + U' X U with X up to 1e16 x the scale of H_w, its
    Cholesky factor and the two triangular solves, all in double-double.  Checked against the oracle's C twin
    (oracle/ddlin.c) and against the residual in exact rational-free form: with cond(H) ~ 1e18 a double
    solver loses everything, the dd solve must still reproduce b to 1e-12 through the split operator."""
    from oracle import ddlin
    rng = np.random.default_rng(n + k)
    B = rng.standard_normal((n + 30, n))
    Hw = B.T @ B
    Hw = 0.5 * (Hw + Hw.T)
    U = rng.standard_normal((k, n))
    if k:
        U[1::3] = U[0::3][: len(U[1::3])] * (1 + 1e-9 * rng.standard_normal((len(U[1::3]), 1)))     # nearly dependent rows
    X = 10.0 ** rng.uniform(8, 16, k)
    b = rng.standard_normal((2, n))
    bl = 1e-17 * rng.standard_normal((2, n))
    xh, xl, nfix = mbfir.test_ddsolve(Hw, U, X, b, bl)
    # oracle twin
    Hh, Hl = Hw.copy(), np.zeros_like(Hw)
    if k:
        ddlin.rank_k(np.ascontiguousarray(U), X, Hh, Hl)
    d0 = np.diag(Hh).copy()
    nfo = ddlin.chol(Hh, Hl, 1e-28, d0)
    Bh, Bl = np.ascontiguousarray(b.T), np.ascontiguousarray(bl.T)
    ddlin.cho_solve(Hh, Hl, Bh, Bl)
    assert nfix == nfo == 0
    ref = (Bh + Bl).T
    assert np.abs((xh + xl) - ref).max() <= 1e-24 * np.abs(ref).max() * max(1.0, X.max() if k else 1.0)
    # residual through the split operator  b - H_w x - U' X (U x), the strong part evaluated in double-double
    # (a double evaluation of U x would be pure rounding noise next to weights of 1e16)
    Xh, Xl = np.ascontiguousarray(xh.T), np.ascontiguousarray(xl.T)
    strong = np.zeros((n, 2))
    if k:
        Uc = np.ascontiguousarray(U)
        uh, ul = ddlin.rows_times(Uc, Xh, Xl)
        yh, yl = ddlin.vec_mul_d(uh.ravel(), ul.ravel(), np.repeat(X, 2))
        sh_, sl_ = np.zeros((n, 2)), np.zeros((n, 2))
        ddlin.cols_times_acc(Uc, yh.reshape(-1, 2), yl.reshape(-1, 2), sh_, sl_)
        strong = sh_ + sl_
    r = b.T - Hw @ (Xh + Xl) - strong
    assert np.abs(r).max() <= 1e-10 * (np.abs(b).max() + np.abs(Hw).max() * np.abs(Xh).max() * n)


@pytest.mark.parametrize("n", [64, 200, 449, 1023, 2100])
def test_cholesky_split_step_equals_the_fused_step(n, monkeypatch):
    """Lock-step batches run the factorisation in other forms than the fused step of a single design -- the split step
    (the row blocks read the stored image of L_kk instead of repeating its 64 pivots) and the single-launch form --
    forced here on a single matrix: same factor and inverse as the fused step, bit for bit."""
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n + 20, n))
    H = B.T @ B + 0.1 * np.eye(n)
    monkeypatch.setenv("MBFIR_CHOL_SPLIT", "0")
    L0, M0 = mbfir.test_chol(H)
    monkeypatch.setenv("MBFIR_CHOL_SPLIT", "1")
    L1, M1 = mbfir.test_chol(H)
    assert np.array_equal(L0, L1) and np.array_equal(M0, M1)
    assert relinf(L1, np.linalg.cholesky(H)) <= 1e-12
    # the whole factorisation in ONE launch (k_chol_dag: ticket-ordered tasks, per-tile dependency counters, strips of
    # tiles per task): the same arithmetic per tile in the same order -- bit for bit again; a second time with the
    # images of the diagonal blocks poisoned before the build (a block that read one before its diagonal block has
    # published it would produce NaN, not last build's plausible numbers)
    monkeypatch.setenv("MBFIR_CHOL_SPLIT", "4")
    L4, M4 = mbfir.test_chol(H)
    assert np.array_equal(L0, L4) and np.array_equal(M0, M4)


def test_cholesky_ill_conditioned_scaling():
    """IPM-like matrix: A' D A with D spanning ten decades (cond ~1e9)."""
    rng = np.random.default_rng(11)
    n, m = 300, 4000
    A = np.cos(np.outer(rng.uniform(-3, 3, m), np.arange(n)))
    d = 10.0 ** rng.uniform(-5, 5, m)
    H = (A.T * d) @ A
    H += 1e-9 * np.abs(H).max() * np.eye(n)
    L, M = mbfir.test_chol(H)
    Lr = np.linalg.cholesky(H)
    assert relinf(L @ L.T, H) <= 1e-12
    assert relinf(L, Lr) <= 1e-6
    b = rng.standard_normal(n)
    x = M.T @ (M @ b)
    # explicit-inverse solve: eps*cond(H) ~ 3e-7 at best; LAPACK trtri gives 2e-5 on this matrix
    assert np.linalg.norm(H @ x - b) <= 2e-4 * np.linalg.norm(b)


@pytest.mark.parametrize("n", [2, 16, 58, 100, 257])
def test_spectral_factorisation_matches_oracle(n):
    """K7: device fmp2/mag2mp (fir_ap_cvx.m:264-304) against the numpy restatement."""
    rng = np.random.default_rng(n)
    h0 = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    r = np.correlate(h0, h0, mode="full")[n - 1:]
    x = np.concatenate([[r[0].real], r[1:].real, r[1:].imag])
    hg = mbfir.test_specfact(x, n)
    ho = specfact.fmp2(specfact.x_to_r(x, n))
    assert relinf(hg, ho) <= 1e-10


def test_fp64_peak_microbenchmark_runs():
    mf, va = mbfir.mfma_peak()
    assert 5.0 < mf < 200.0 and 5.0 < va < 200.0


def test_double_double_solve_by_block_inverses_agrees_with_the_substitution(monkeypatch):
    """The triangular solves of the extended-precision path multiply by the inverses of the 64 x 64 diagonal blocks of the
    dd factor (k_dd_blockinv, k_dd_trsv_bi: 34 chain steps at np = 1088, none with a substitution in it);
    MBFIR_DD_BLOCKINV=0 is the substitution on 32-row blocks (k_dd_trsv_mw).  Same factor, and solutions that agree far
    below anything the refinement around them could see -- here with strong weights up to 1e14 on top of H_w."""
    rng = np.random.default_rng(21)
    n, k = 700, 120
    B = rng.standard_normal((n + 30, n))
    Hw = B.T @ B
    Hw = 0.5 * (Hw + Hw.T)
    U = rng.standard_normal((k, n))
    X = 10.0 ** rng.uniform(8, 14, k)
    b = rng.standard_normal((2, n))
    bl = np.zeros((2, n))
    xh1, xl1, nf1 = mbfir.test_ddsolve(Hw, U, X, b, bl)
    monkeypatch.setenv("MBFIR_DD_BLOCKINV", "0")
    xh0, xl0, nf0 = mbfir.test_ddsolve(Hw, U, X, b, bl)
    assert nf0 == nf1 == 0
    d = np.abs((xh1 - xh0) + (xl1 - xl0)).max()
    assert d <= 1e-22 * np.abs(xh0).max() * X.max()
    assert not np.array_equal(xl1, xl0) or np.array_equal(xh1, xh0)          # (two different orders of summation: the low words differ)


def test_double_double_factorisation_is_the_same_with_32_and_64_wide_tiles(monkeypatch):
    """The dd rank-k kernels (k_dd_syrk, k_ddchol_update) run on 32 x 32 tiles up to np = 2048 (four times the workgroups
    of a quarter the length: the trailing matrix of np = 1088 is 51 tiles of 64 x 64 on average, a fifth of the chip) and
    on 64 x 64 tiles beyond; MBFIR_DD_TILE forces either.  Every entry's sum runs in the same order: same factor and
    same solution bit for bit."""
    rng = np.random.default_rng(33)
    n, k = 449, 70
    B = rng.standard_normal((n + 30, n))
    Hw = B.T @ B
    Hw = 0.5 * (Hw + Hw.T)
    U = rng.standard_normal((k, n))
    X = 10.0 ** rng.uniform(8, 15, k)
    b = rng.standard_normal((2, n))
    bl = np.zeros((2, n))
    out = {}
    for tile in ("32", "64"):
        monkeypatch.setenv("MBFIR_DD_TILE", tile)
        out[tile] = mbfir.test_ddsolve(Hw, U, X, b, bl, factor=True)
    for a32, a64 in zip(out["32"], out["64"]):
        assert np.array_equal(a32, a64)


def test_double_double_solve_with_one_right_hand_side_equals_the_first_of_two():
    """k_dd_trsv_mw<1> against k_dd_trsv_mw<2>: the blocks' sums run in the same order per right-hand side, so the
    single solve reproduces column 0 of the double solve bit for bit."""
    rng = np.random.default_rng(9)
    n, k = 300, 40
    B = rng.standard_normal((n + 30, n))
    Hw = B.T @ B
    Hw = 0.5 * (Hw + Hw.T)
    U = rng.standard_normal((k, n))
    X = 10.0 ** rng.uniform(8, 14, k)
    b = rng.standard_normal((2, n))
    bl = np.zeros((2, n))
    xh2, xl2, _ = mbfir.test_ddsolve(Hw, U, X, b, bl)
    xh1, xl1, _ = mbfir.test_ddsolve(Hw, U, X, b[:1], bl[:1])
    assert np.array_equal(xh1[0], xh2[0]) and np.array_equal(xl1[0], xl2[0])


@pytest.mark.parametrize("n,nl", [(130, 3), (449, 5), (1023, 8)])
def test_cholesky_of_a_lock_step_batch_every_form_bit_identical(n, nl):
    """K4 as a lock-step batch runs it: `nl` different matrices factorised TOGETHER (mbfir_test_chol_lanes), in every form
    of chol_inv_launch -- fused per-step (0), split per-step with the device flag (1), split in two launches (2), the whole
    factorisation in ONE launch with ticket-ordered tasks (4, the default for >= 3 lanes).  Every lane's L and L^-1 are the
    same bit for bit in all forms and equal to the single-matrix factorisation; the stored transpose equals the inverse
    factor (checked inside the hook); a masked lane leaves the others untouched."""
    rng = np.random.default_rng(100 * n + nl)
    Hs = []
    for b in range(nl):
        B = rng.standard_normal((n + 20, n))
        Hs.append(B.T @ B + (0.1 + 0.05 * b) * np.eye(n))
    Hs = np.array(Hs)
    ref = [mbfir.test_chol(H) for H in Hs]                      # one matrix at a time (fused step)
    outs = {form: mbfir.test_chol_lanes(Hs, form=form) for form in (0, 1, 2, 4, -1)}
    for form, (L, M) in outs.items():
        for b in range(nl):
            assert np.array_equal(L[b], ref[b][0]) and np.array_equal(M[b], ref[b][1]), (form, b)
    for b in range(nl):
        assert relinf(ref[b][0], np.linalg.cholesky(Hs[b])) <= 1e-12
    mask = [1] * nl
    mask[1] = 0
    L, M = mbfir.test_chol_lanes(Hs, form=4, mask=mask)
    for b in range(nl):
        if mask[b]:
            assert np.array_equal(L[b], ref[b][0]) and np.array_equal(M[b], ref[b][1])
        else:
            assert not L[b].any() and not M[b].any()
