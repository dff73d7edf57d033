/* Double-double ("dd", ~32 significant digits) dense kernels for the oracle's extended-precision
 * normal-equation solve (test infrastructure, see oracle/__init__.py; used by oracle/conic_ipm.py).
 *
 * Written from the published error-free transformations (T. J. Dekker 1971; D. E. Knuth TAOCP 2,
 * 4.2.2 "TwoSum"; Hida, Li, Bailey, "Algorithms for quad-double precision floating point arithmetic",
 * ARITH-15, 2001) -- nothing here follows a file of the reference, which delegates the solve to
 * CVX/SDPT3 (fir_qp_cvx.m:145-191).
 *
 * Build: see oracle/Makefile (-O2 -mfma -ffp-contract=off: contractions would break TwoSum).
 * All matrices row-major, lower triangle referenced.  A dd number is the unevaluated sum hi + lo.
 */
#include <math.h>
#include <stddef.h>

typedef struct { double h, l; } dd;

static inline dd two_sum(double a, double b) {
    double s = a + b, bb = s - a;
    dd r = { s, (a - (s - bb)) + (b - bb) };
    return r;
}
static inline dd quick_two_sum(double a, double b) {
    double s = a + b;
    dd r = { s, b - (s - a) };
    return r;
}
static inline dd two_prod(double a, double b) {
    double p = a * b;
    dd r = { p, fma(a, b, -p) };
    return r;
}
static inline dd dd_add(dd a, dd b) {
    dd s = two_sum(a.h, b.h), t = two_sum(a.l, b.l);
    s.l += t.h;
    s = quick_two_sum(s.h, s.l);
    s.l += t.l;
    return quick_two_sum(s.h, s.l);
}
static inline dd dd_neg(dd a) { dd r = { -a.h, -a.l }; return r; }
static inline dd dd_mul(dd a, dd b) {
    dd p = two_prod(a.h, b.h);
    p.l += a.h * b.l + a.l * b.h;
    return quick_two_sum(p.h, p.l);
}
static inline dd dd_mul_d(dd a, double b) {
    dd p = two_prod(a.h, b);
    p.l += a.l * b;
    return quick_two_sum(p.h, p.l);
}
static inline dd dd_div(dd a, dd b) {
    double q1 = a.h / b.h;
    dd r = dd_add(a, dd_neg(dd_mul_d(b, q1)));
    double q2 = r.h / b.h;
    r = dd_add(r, dd_neg(dd_mul_d(b, q2)));
    double q3 = r.h / b.h;
    dd q = quick_two_sum(q1, q2);
    dd q3d = { q3, 0.0 };
    return dd_add(q, q3d);
}
static inline dd dd_sqrt(dd a) {
    /* Karp's trick: sqrt(a) = a*x + [a - (a*x)^2] * x / 2, x = 1/sqrt(a.h) */
    if (a.h <= 0.0) { dd z = { 0.0, 0.0 }; return z; }
    double x = 1.0 / sqrt(a.h), ax = a.h * x;
    dd sq = two_prod(ax, ax);
    dd e = dd_add(a, dd_neg(sq));
    return two_sum(ax, e.h * (x * 0.5));
}

/* H (dd, N x N, lower) += sum_r x[r] * u_r u_r'   with u_r = row r of U (k x N, plain doubles) */
void dd_rank_k(int N, int k, const double *U, const double *x, double *Hh, double *Hl) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j <= i; ++j) {
            dd acc = { Hh[(size_t)i * N + j], Hl[(size_t)i * N + j] };
            for (int r = 0; r < k; ++r) {
                double ui = U[(size_t)r * N + i], uj = U[(size_t)r * N + j];
                if (ui == 0.0 || uj == 0.0) continue;
                dd p = two_prod(ui, uj);
                acc = dd_add(acc, dd_mul_d(p, x[r]));
            }
            Hh[(size_t)i * N + j] = acc.h;
            Hl[(size_t)i * N + j] = acc.l;
        }
    }
}

/* In-place lower Cholesky in dd (left-looking by columns).  Pivot rule of conic_ipm.chol_piv: a pivot
 * not above pivtol * d0[j] is replaced by d0[j].  Returns the number of replaced pivots. */
int dd_chol(int N, double *Hh, double *Hl, double pivtol, const double *d0) {
    int nfix = 0;
    for (int j = 0; j < N; ++j) {
#pragma omp parallel for schedule(static)
        for (int i = j; i < N; ++i) {
            const double *ah = Hh + (size_t)i * N, *al = Hl + (size_t)i * N;
            const double *bh = Hh + (size_t)j * N, *bl = Hl + (size_t)j * N;
            dd acc = { ah[j], al[j] };
            for (int c = 0; c < j; ++c) {
                dd a = { ah[c], al[c] }, b = { bh[c], bl[c] };
                acc = dd_add(acc, dd_neg(dd_mul(a, b)));
            }
            /* stash: column j of row i holds the updated entry; the diagonal is finished below */
            ((double *)ah)[j] = acc.h;
            ((double *)al)[j] = acc.l;
        }
        dd p = { Hh[(size_t)j * N + j], Hl[(size_t)j * N + j] };
        if (!(p.h > pivtol * d0[j])) {
            p.h = d0[j] > 1e-300 ? d0[j] : 1e-300;
            p.l = 0.0;
            ++nfix;
        }
        dd r = dd_sqrt(p);
        Hh[(size_t)j * N + j] = r.h;
        Hl[(size_t)j * N + j] = r.l;
#pragma omp parallel for schedule(static)
        for (int i = j + 1; i < N; ++i) {
            dd v = { Hh[(size_t)i * N + j], Hl[(size_t)i * N + j] };
            v = dd_div(v, r);
            Hh[(size_t)i * N + j] = v.h;
            Hl[(size_t)i * N + j] = v.l;
        }
    }
    return nfix;
}

/* x = (L L')^-1 b for nrhs right-hand sides stored as columns of B (N x nrhs, row-major), all dd */
void dd_cho_solve(int N, int nrhs, const double *Lh, const double *Ll, double *Bh, double *Bl) {
#pragma omp parallel for schedule(static)
    for (int q = 0; q < nrhs; ++q) {
        for (int i = 0; i < N; ++i) {
            dd acc = { Bh[(size_t)i * nrhs + q], Bl[(size_t)i * nrhs + q] };
            for (int c = 0; c < i; ++c) {
                dd a = { Lh[(size_t)i * N + c], Ll[(size_t)i * N + c] };
                dd b = { Bh[(size_t)c * nrhs + q], Bl[(size_t)c * nrhs + q] };
                acc = dd_add(acc, dd_neg(dd_mul(a, b)));
            }
            dd d = { Lh[(size_t)i * N + i], Ll[(size_t)i * N + i] };
            acc = dd_div(acc, d);
            Bh[(size_t)i * nrhs + q] = acc.h;
            Bl[(size_t)i * nrhs + q] = acc.l;
        }
        for (int i = N - 1; i >= 0; --i) {
            dd acc = { Bh[(size_t)i * nrhs + q], Bl[(size_t)i * nrhs + q] };
            for (int c = i + 1; c < N; ++c) {
                dd a = { Lh[(size_t)c * N + i], Ll[(size_t)c * N + i] };
                dd b = { Bh[(size_t)c * nrhs + q], Bl[(size_t)c * nrhs + q] };
                acc = dd_add(acc, dd_neg(dd_mul(a, b)));
            }
            dd d = { Lh[(size_t)i * N + i], Ll[(size_t)i * N + i] };
            acc = dd_div(acc, d);
            Bh[(size_t)i * nrhs + q] = acc.h;
            Bl[(size_t)i * nrhs + q] = acc.l;
        }
    }
}

/* Y (k x nrhs, dd) = U (k x N) X (N x nrhs, dd) */
void dd_rows_times(int k, int N, int nrhs, const double *U, const double *Xh, const double *Xl,
                   double *Yh, double *Yl) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < k; ++r)
        for (int q = 0; q < nrhs; ++q) {
            dd acc = { 0.0, 0.0 };
            for (int c = 0; c < N; ++c) {
                double u = U[(size_t)r * N + c];
                if (u == 0.0) continue;
                dd x = { Xh[(size_t)c * nrhs + q], Xl[(size_t)c * nrhs + q] };
                acc = dd_add(acc, dd_mul_d(x, u));
            }
            Yh[(size_t)r * nrhs + q] = acc.h;
            Yl[(size_t)r * nrhs + q] = acc.l;
        }
}

/* X (N x nrhs, dd) += U' Y   with Y (k x nrhs, dd) */
void dd_cols_times_acc(int k, int N, int nrhs, const double *U, const double *Yh, const double *Yl,
                       double *Xh, double *Xl) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < N; ++c)
        for (int q = 0; q < nrhs; ++q) {
            dd acc = { Xh[(size_t)c * nrhs + q], Xl[(size_t)c * nrhs + q] };
            for (int r = 0; r < k; ++r) {
                double u = U[(size_t)r * N + c];
                if (u == 0.0) continue;
                dd y = { Yh[(size_t)r * nrhs + q], Yl[(size_t)r * nrhs + q] };
                acc = dd_add(acc, dd_mul_d(y, u));
            }
            Xh[(size_t)c * nrhs + q] = acc.h;
            Xl[(size_t)c * nrhs + q] = acc.l;
        }
}

/* elementwise helpers on dd vectors of length n */
void dd_vec_mul_d(int n, const double *ah, const double *al, const double *b, double *oh, double *ol) {
    for (int i = 0; i < n; ++i) {
        dd a = { ah[i], al[i] };
        dd r = dd_mul_d(a, b[i]);
        oh[i] = r.h; ol[i] = r.l;
    }
}
void dd_vec_sub(int n, const double *ah, const double *al, const double *bh, const double *bl, double *oh, double *ol) {
    for (int i = 0; i < n; ++i) {
        dd a = { ah[i], al[i] }, b = { -bh[i], -bl[i] };
        dd r = dd_add(a, b);
        oh[i] = r.h; ol[i] = r.l;
    }
}
