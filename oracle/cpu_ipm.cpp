// C++ / OpenMP twin of oracle/conic_ipm.py -- TEST INFRASTRUCTURE (see oracle/__init__.py): the CPU baseline that
// SURVEY.md section 8(d) asks for ("the build's own C++ fp64 CPU solver, same algorithm, OpenMP, at 1 thread and at all
// cores on identical specs").  bench.py's cpu_baseline leg times it; tests hold it to the NumPy oracle.  It is the
// DENSE algorithm (G stored, normal matrix by a blocked Gram product), like the NumPy oracle and unlike the product's
// lattice path.  Written from the same published mathematics as conic_ipm.py (homogeneous self-dual embedding,
// Nesterov-Todd scaling, Mehrotra predictor-corrector with sigma = min((1-alpha_aff)^3, 0.25), normal equations +
// Cholesky + preconditioned CG on the exact operator); nothing here follows a file of the reference, which delegates
// the solve to CVX / linprog / quadprog (fir_ap_cvx.m:160-169, fir_qp_cvx.m:145-191, ss/fir_linprog.m:245-251,
// ss/fir_qprog_phs.m:339-342).  The extended-precision KKT solve of conic_ipm.py is not mirrored (the benchmarked
// programs do not need it).
//
// Build: oracle/Makefile  (g++ -O3 -march=x86-64-v3 -fopenmp -shared -fPIC)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>
#include <immintrin.h>
#include <omp.h>

namespace {

typedef std::vector<double> vec;
const double POLISH = 1e-2;      // conic_ipm.py: the end game
const int POLISH_MAX = 3, POLISH_SWEEPS = 1;
const double POLISH_APPROACH = 30.0;      // conic_ipm.py: the final approach
const double CORR_DELTA = 0.5, CORR_BMIN = 0.1, CORR_BMAX = 10.0, CORR_ACCEPT = 1.01, CORR_ETA = 1.0;      // conic_ipm.py: the centrality corrector
const double STEP = 0.99, SIGMA_MAX = 0.25, SIGMA_MAX_CORR = 0.05 /* where the centrality corrector follows: programs with orthant rows (conic_ipm.py) */, REFTOL = 1e-11, REFETA = 1e-1, INACC_FEAS = 1e-6, INACC_GAP = 1.22e-4, PIVTOL = 1e-13;
const int MAX_SWEEPS = 8, WALL_ITERS = 3, NB = 64;
enum { ST_OPTIMAL = 0, ST_PINF = 1, ST_DINF = 2, ST_MAXIT = 3, ST_NUMERICAL = 4, ST_INACC = 5 };

struct Cone {
    int l, nq3, big, o3, ob, R, degree;
};
inline double jres(double v0, double n1) { return (v0 - n1) * (v0 + n1); }

// max over cones of -(distance inside)
double min_residual(const Cone& K, const double* v) {
    double t = -std::numeric_limits<double>::infinity();
    for (int i = 0; i < K.l; ++i) t = std::max(t, -v[i]);
    for (int c = 0; c < K.nq3; ++c) {
        const double* q = v + K.o3 + 3 * c;
        t = std::max(t, std::hypot(q[1], q[2]) - q[0]);
    }
    if (K.big) {
        double n = 0;
        for (int i = 1; i < K.big; ++i) n += v[K.ob + i] * v[K.ob + i];
        t = std::max(t, std::sqrt(n) - v[K.ob]);
    }
    return t;
}
void add_e(const Cone& K, double* v, double a) {
    for (int i = 0; i < K.l; ++i) v[i] += a;
    for (int c = 0; c < K.nq3; ++c) v[K.o3 + 3 * c] += a;
    if (K.big) v[K.ob] += a;
}

// NT scaling of one second-order cone of dimension d: eta, wbar
void soc_scaling(const double* s, const double* z, int d, double& eta, double* wbar) {
    double ns = 0, nz = 0, sz = 0;
    for (int i = 1; i < d; ++i) { ns += s[i] * s[i]; nz += z[i] * z[i]; }
    ns = std::sqrt(ns); nz = std::sqrt(nz);
    const double a = std::sqrt(jres(s[0], ns)), b = std::sqrt(jres(z[0], nz));
    for (int i = 0; i < d; ++i) sz += (s[i] / a) * (z[i] / b);
    const double gamma = std::sqrt((1.0 + sz) / 2.0);
    wbar[0] = (s[0] / a + z[0] / b) / (2 * gamma);
    for (int i = 1; i < d; ++i) wbar[i] = (s[i] / a - z[i] / b) / (2 * gamma);
    eta = std::sqrt(a / b);
}
// out = W u (inverse = false) or W^-1 u
void soc_apply(double eta, const double* wbar, const double* u, double* out, int d, bool inverse) {
    double dot = 0;
    for (int i = 1; i < d; ++i) dot += wbar[i] * u[i];
    const double w0 = wbar[0], u0 = u[0];
    if (inverse) {
        const double f = -u0 + dot / (1 + w0);
        for (int i = 1; i < d; ++i) out[i] = (u[i] + f * wbar[i]) / eta;
        out[0] = (w0 * u0 - dot) / eta;
    } else {
        const double f = u0 + dot / (1 + w0);
        for (int i = 1; i < d; ++i) out[i] = (u[i] + f * wbar[i]) * eta;
        out[0] = (w0 * u0 + dot) * eta;
    }
}
// out = W^-2 v
void soc_inv2(double eta, const double* wbar, const double* v, double* out, int d) {
    double uv = wbar[0] * v[0];
    for (int i = 1; i < d; ++i) uv -= wbar[i] * v[i];
    const double e2 = 1.0 / (eta * eta);
    out[0] = (2 * wbar[0] * uv - v[0]) * e2;
    for (int i = 1; i < d; ++i) out[i] = (2 * (-wbar[i]) * uv + v[i]) * e2;
}
void soc_prod(const double* u, const double* v, double* out, int d) {
    double t = 0;
    for (int i = 0; i < d; ++i) t += u[i] * v[i];
    for (int i = 1; i < d; ++i) out[i] = u[0] * v[i] + v[0] * u[i];
    out[0] = t;
}
void soc_div(const double* lam, const double* dd, double* out, int d) {       // x with lam o x = dd
    double n = 0, ld = 0;
    for (int i = 1; i < d; ++i) { n += lam[i] * lam[i]; ld += lam[i] * dd[i]; }
    const double a = jres(lam[0], std::sqrt(n));
    out[0] = (lam[0] * dd[0] - ld) / a;
    for (int i = 1; i < d; ++i) out[i] = (dd[i] - out[0] * lam[i]) / lam[0];
}
double soc_step(const double* lam, const double* dd, int d) {                // ||rho_1|| - rho_0, rho = T(lam) dd
    double n = 0;
    for (int i = 1; i < d; ++i) n += lam[i] * lam[i];
    const double a = std::sqrt(jres(lam[0], std::sqrt(n))), lb0 = lam[0] / a;
    double dot = 0;
    for (int i = 1; i < d; ++i) dot += (lam[i] / a) * dd[i];
    const double rho0 = (lb0 * dd[0] - dot) / a, f = -dd[0] + dot / (1 + lb0);
    double t = 0;
    for (int i = 1; i < d; ++i) { const double r = (dd[i] + f * lam[i] / a) / a; t += r * r; }
    return std::sqrt(t) - rho0;
}

struct Scaling {
    const Cone* K = nullptr;
    vec wl, dl, eta3, wb3, wbb;
    double etab = 1;
    void build(const Cone& Kc, const vec& s, const vec& z) {
        K = &Kc;
        wl.resize(Kc.l); dl.resize(Kc.l); eta3.resize(Kc.nq3); wb3.resize(3 * (size_t)Kc.nq3); wbb.resize(Kc.big);
        for (int i = 0; i < Kc.l; ++i) { wl[i] = std::sqrt(s[i] / z[i]); dl[i] = z[i] / s[i]; }
        for (int c = 0; c < Kc.nq3; ++c) soc_scaling(&s[Kc.o3 + 3 * c], &z[Kc.o3 + 3 * c], 3, eta3[c], &wb3[3 * c]);
        if (Kc.big) soc_scaling(&s[Kc.ob], &z[Kc.ob], Kc.big, etab, wbb.data());
    }
    void apply(const double* v, double* out, bool inverse) const {
        for (int i = 0; i < K->l; ++i) out[i] = inverse ? v[i] / wl[i] : v[i] * wl[i];
        for (int c = 0; c < K->nq3; ++c) soc_apply(eta3[c], &wb3[3 * c], v + K->o3 + 3 * c, out + K->o3 + 3 * c, 3, inverse);
        if (K->big) soc_apply(etab, wbb.data(), v + K->ob, out + K->ob, K->big, inverse);
    }
    void inv2(const double* v, double* out) const {
        for (int i = 0; i < K->l; ++i) out[i] = dl[i] * v[i];
        for (int c = 0; c < K->nq3; ++c) soc_inv2(eta3[c], &wb3[3 * c], v + K->o3 + 3 * c, out + K->o3 + 3 * c, 3);
        if (K->big) soc_inv2(etab, wbb.data(), v + K->ob, out + K->ob, K->big);
    }
};
void cone_prod(const Cone& K, const double* u, const double* v, double* out) {
    for (int i = 0; i < K.l; ++i) out[i] = u[i] * v[i];
    for (int c = 0; c < K.nq3; ++c) soc_prod(u + K.o3 + 3 * c, v + K.o3 + 3 * c, out + K.o3 + 3 * c, 3);
    if (K.big) soc_prod(u + K.ob, v + K.ob, out + K.ob, K.big);
}
void cone_div(const Cone& K, const double* lam, const double* d, double* out) {
    for (int i = 0; i < K.l; ++i) out[i] = d[i] / lam[i];
    for (int c = 0; c < K.nq3; ++c) soc_div(lam + K.o3 + 3 * c, d + K.o3 + 3 * c, out + K.o3 + 3 * c, 3);
    if (K.big) soc_div(lam + K.ob, d + K.ob, out + K.ob, K.big);
}
double max_step(const Cone& K, const double* lam, const double* d) {
    double t = -std::numeric_limits<double>::infinity();
    for (int i = 0; i < K.l; ++i) t = std::max(t, -d[i] / lam[i]);
    for (int c = 0; c < K.nq3; ++c) t = std::max(t, soc_step(lam + K.o3 + 3 * c, d + K.o3 + 3 * c, 3));
    if (K.big) t = std::max(t, soc_step(lam + K.ob, d + K.ob, K.big));
    return t;
}

// ---- dense kernels (row-major G, R x N) ---------------------------------------------------------------------
struct Dense {
    int R, N;
    const double* G;
    // out (R) = G v
    void mul(const double* v, double* out) const {
#pragma omp parallel for schedule(static)
        for (int r = 0; r < R; ++r) {
            const double* g = G + (size_t)r * N;
            double a = 0;
            for (int j = 0; j < N; ++j) a += g[j] * v[j];
            out[r] = a;
        }
    }
    // out (N) = G' u : per-thread partial sums over row blocks, folded in thread order (deterministic per thread count)
    void mulT(const double* u, double* out) const {
        const int nt = omp_get_max_threads();
        std::vector<double> part((size_t)nt * N, 0.0);
#pragma omp parallel
        {
            double* p = part.data() + (size_t)omp_get_thread_num() * N;
#pragma omp for schedule(static)
            for (int r = 0; r < R; ++r) {
                const double* g = G + (size_t)r * N;
                const double a = u[r];
                if (a != 0.0)
                    for (int j = 0; j < N; ++j) p[j] += a * g[j];
            }
        }
        for (int j = 0; j < N; ++j) {
            double a = 0;
            for (int t = 0; t < nt; ++t) a += part[(size_t)t * N + j];
            out[j] = a;
        }
    }
};

// ---- Gram product H = B'B on a panel-packed B ------------------------------------------------------------------
// B is kept in column panels of 8: element (r, j) at ((j / 8) * R + r) * 8 + j % 8 (columns padded with zeros), so a
// panel is one contiguous stream of cache lines.  32 x 32 tiles of the lower triangle run in parallel; inside a tile the
// rows go in blocks of 128 (8 panels x 8 KB stay in L2) through a 4 x 8 register kernel (AVX2 + FMA: 8 accumulators).
inline size_t pk(int R, int r, int j) { return ((size_t)(j >> 3) * R + r) * 8 + (j & 7); }

inline void mk4x8(const double* A, int aoff, const double* Bq, int nr, double* C, int ldc) {
    __m256d c00 = _mm256_setzero_pd(), c01 = c00, c10 = c00, c11 = c00, c20 = c00, c21 = c00, c30 = c00, c31 = c00;
    for (int r = 0; r < nr; ++r) {
        const __m256d b0 = _mm256_loadu_pd(Bq + 8 * r), b1 = _mm256_loadu_pd(Bq + 8 * r + 4);
        const double* a = A + 8 * r + aoff;
        __m256d t = _mm256_broadcast_sd(a);
        c00 = _mm256_fmadd_pd(t, b0, c00); c01 = _mm256_fmadd_pd(t, b1, c01);
        t = _mm256_broadcast_sd(a + 1);
        c10 = _mm256_fmadd_pd(t, b0, c10); c11 = _mm256_fmadd_pd(t, b1, c11);
        t = _mm256_broadcast_sd(a + 2);
        c20 = _mm256_fmadd_pd(t, b0, c20); c21 = _mm256_fmadd_pd(t, b1, c21);
        t = _mm256_broadcast_sd(a + 3);
        c30 = _mm256_fmadd_pd(t, b0, c30); c31 = _mm256_fmadd_pd(t, b1, c31);
    }
    _mm256_storeu_pd(C, _mm256_add_pd(_mm256_loadu_pd(C), c00)); _mm256_storeu_pd(C + 4, _mm256_add_pd(_mm256_loadu_pd(C + 4), c01));
    C += ldc;
    _mm256_storeu_pd(C, _mm256_add_pd(_mm256_loadu_pd(C), c10)); _mm256_storeu_pd(C + 4, _mm256_add_pd(_mm256_loadu_pd(C + 4), c11));
    C += ldc;
    _mm256_storeu_pd(C, _mm256_add_pd(_mm256_loadu_pd(C), c20)); _mm256_storeu_pd(C + 4, _mm256_add_pd(_mm256_loadu_pd(C + 4), c21));
    C += ldc;
    _mm256_storeu_pd(C, _mm256_add_pd(_mm256_loadu_pd(C), c30)); _mm256_storeu_pd(C + 4, _mm256_add_pd(_mm256_loadu_pd(C + 4), c31));
}

#ifdef __AVX512F__
// the same product on 512-bit registers (built with -march=native on a machine that has them: oracle/Makefile,
// _ref/libcpu_ipm_native.so): 8 x 8 block, 8 accumulators, one panel-row load and eight broadcast-FMAs per row; every
// element sums its products in the same order as in mk4x8, so the two kernels agree bit for bit
inline void mk8x8(const double* A, const double* Bq, int nr, double* C, int ldc) {
    __m512d c0 = _mm512_setzero_pd(), c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int r = 0; r < nr; ++r) {
        const __m512d b = _mm512_loadu_pd(Bq + 8 * r);
        const double* a = A + 8 * r;
        c0 = _mm512_fmadd_pd(_mm512_set1_pd(a[0]), b, c0); c1 = _mm512_fmadd_pd(_mm512_set1_pd(a[1]), b, c1);
        c2 = _mm512_fmadd_pd(_mm512_set1_pd(a[2]), b, c2); c3 = _mm512_fmadd_pd(_mm512_set1_pd(a[3]), b, c3);
        c4 = _mm512_fmadd_pd(_mm512_set1_pd(a[4]), b, c4); c5 = _mm512_fmadd_pd(_mm512_set1_pd(a[5]), b, c5);
        c6 = _mm512_fmadd_pd(_mm512_set1_pd(a[6]), b, c6); c7 = _mm512_fmadd_pd(_mm512_set1_pd(a[7]), b, c7);
    }
    const __m512d cs[8] = {c0, c1, c2, c3, c4, c5, c6, c7};
    for (int i = 0; i < 8; ++i) _mm512_storeu_pd(C + (size_t)i * ldc, _mm512_add_pd(_mm512_loadu_pd(C + (size_t)i * ldc), cs[i]));
}
#endif
extern "C" const char* cpu_ipm_isa(void) {
#ifdef __AVX512F__
    return "avx512f (8 x 8 zmm Gram kernel)";
#else
    return "avx2+fma (4 x 8 ymm Gram kernel)";
#endif
}

void gram(const double* Bp, int R, int N, double* H) {
    const int npan = (N + 7) / 8, nt = (npan + 3) / 4, RB = 128;
    std::vector<std::pair<int, int>> tiles;
    for (int i = 0; i < nt; ++i)
        for (int j = 0; j <= i; ++j) tiles.push_back({i, j});
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t q = 0; q < tiles.size(); ++q) {
        const int pi0 = tiles[q].first * 4, pj0 = tiles[q].second * 4, npi = std::min(4, npan - pi0), npj = std::min(4, npan - pj0);
        alignas(64) double C[32 * 32];
        std::memset(C, 0, sizeof(C));
        for (int r0 = 0; r0 < R; r0 += RB) {
            const int nr = std::min(RB, R - r0);
            for (int pa = 0; pa < npi; ++pa) {
                const double* A = Bp + ((size_t)(pi0 + pa) * R + r0) * 8;
                for (int pb = 0; pb < npj; ++pb) {
                    const double* Bq = Bp + ((size_t)(pj0 + pb) * R + r0) * 8;
#ifdef __AVX512F__
                    mk8x8(A, Bq, nr, C + (pa * 8) * 32 + pb * 8, 32);
#else
                    mk4x8(A, 0, Bq, nr, C + (pa * 8) * 32 + pb * 8, 32);
                    mk4x8(A, 4, Bq, nr, C + (pa * 8 + 4) * 32 + pb * 8, 32);
#endif
                }
            }
        }
        for (int a = 0; a < 8 * npi; ++a)
            for (int b = 0; b < 8 * npj; ++b) {
                const int i = pi0 * 8 + a, j = pj0 * 8 + b;
                if (i < N && j < N) { H[(size_t)i * N + j] = C[a * 32 + b]; H[(size_t)j * N + i] = C[a * 32 + b]; }
            }
    }
}

// blocked right-looking Cholesky with the pivot rule of conic_ipm.chol_piv; L in the lower triangle of H
int chol_piv(double* H, int N) {
    vec d0(N);
    for (int i = 0; i < N; ++i) d0[i] = H[(size_t)i * N + i];
    int nfix = 0;
    for (int k0 = 0; k0 < N; k0 += NB) {
        const int k1 = std::min(N, k0 + NB), nb = k1 - k0;
        for (int j = k0; j < k1; ++j) {                       // diagonal block, unblocked
            double p = H[(size_t)j * N + j];
            if (!(p > PIVTOL * d0[j])) { p = std::max(d0[j], 1e-300); ++nfix; }
            const double r = std::sqrt(p);
            H[(size_t)j * N + j] = r;
            for (int i = j + 1; i < k1; ++i) H[(size_t)i * N + j] /= r;
            for (int i = j + 1; i < k1; ++i) {
                const double lij = H[(size_t)i * N + j];
                for (int c = j + 1; c <= i; ++c) H[(size_t)i * N + c] -= lij * H[(size_t)c * N + j];
            }
        }
        if (k1 >= N) break;
#pragma omp parallel for schedule(static)
        for (int i = k1; i < N; ++i) {                        // panel rows by forward substitution
            double* x = H + (size_t)i * N + k0;
            for (int j = 0; j < nb; ++j) {
                double v = x[j];
                const double* dj = H + (size_t)(k0 + j) * N + k0;
                for (int c = 0; c < j; ++c) v -= x[c] * dj[c];
                x[j] = v / dj[j];
            }
        }
#pragma omp parallel for schedule(dynamic, 8)
        for (int i = k1; i < N; ++i) {                        // trailing update, lower triangle
            const double* li = H + (size_t)i * N + k0;
            for (int j = k1; j <= i; ++j) {
                const double* lj = H + (size_t)j * N + k0;
                double a = 0;
                for (int c = 0; c < nb; ++c) a += li[c] * lj[c];
                H[(size_t)i * N + j] -= a;
            }
        }
    }
    return nfix;
}
// x = (L L')^-1 b
void cho_solve(const double* L, int N, const double* b, double* x) {
    for (int i = 0; i < N; ++i) {
        double v = b[i];
        const double* li = L + (size_t)i * N;
        for (int c = 0; c < i; ++c) v -= li[c] * x[c];
        x[i] = v / li[i];
    }
    for (int i = N - 1; i >= 0; --i) {
        double v = x[i];
        for (int c = i + 1; c < N; ++c) v -= L[(size_t)c * N + i] * x[c];
        x[i] = v / L[(size_t)i * N + i];
    }
}
double nrm2(const double* v, int n) { double a = 0; for (int i = 0; i < n; ++i) a += v[i] * v[i]; return std::sqrt(a); }
double dot(const double* a, const double* b, int n) { double t = 0; for (int i = 0; i < n; ++i) t += a[i] * b[i]; return t; }

}  // namespace

extern "C" {

// info_out[16]: status, iters, pcost, dcost, gap, relgap, pres, dres, chol_fixes, seconds_factor, seconds_total
int cpu_ipm_solve(int R, int N, const double* G, const double* h, const double* c, int l, int nq3, int big, int max_iter,
                  double feastol, double abstol, double reltol, int refine, int nthreads, double* x_out, double* info_out) {
    if (nthreads > 0) omp_set_num_threads(nthreads);
    const double t_begin = omp_get_wtime();
    double t_factor = 0;
    Cone K{l, nq3, big, l, l + 3 * nq3, l + 3 * nq3 + big, l + nq3 + (big ? 1 : 0)};
    if (K.R != R) return -1;
    Dense D{R, N, G};
    const double nrm_h = std::max(1.0, nrm2(h, R)), nrm_c = std::max(1.0, nrm2(c, N));
    const int Npad = (N + 7) / 8 * 8;
    vec H((size_t)N * N), B((size_t)R * Npad, 0.0), tmpR(R), tmpR2(R), tmpN(N), tmpN2(N);
    Scaling W;
    bool unit = true;
    int chol_fixes = 0, nsweep = refine, nsweep_ctl = refine;      // (nsweep_ctl: the controller's count; nsweep: what the iteration's solves run)
    std::vector<vec> sweep_log;
    auto winv2 = [&](const double* v, double* out) { if (unit) std::memcpy(out, v, sizeof(double) * R); else W.inv2(v, out); };
    auto factor = [&]() {
        const double t0 = omp_get_wtime();
        if (unit) {
#pragma omp parallel for schedule(static)
            for (int r = 0; r < R; ++r)
                for (int j = 0; j < N; ++j) B[pk(R, r, j)] = G[(size_t)r * N + j];
        } else {
            // B = W^-1 G, column by column of G = row-wise for the LP rows, cone blocks for the SOCs
#pragma omp parallel for schedule(static)
            for (int r = 0; r < K.l; ++r) {
                const double f = 1.0 / W.wl[r];
                for (int j = 0; j < N; ++j) B[pk(R, r, j)] = G[(size_t)r * N + j] * f;
            }
#pragma omp parallel for schedule(static)
            for (int cc = 0; cc < K.nq3; ++cc) {
                const size_t o = (size_t)(K.o3 + 3 * cc) * N;
                for (int j = 0; j < N; ++j) {
                    const double u[3] = {G[o + j], G[o + N + j], G[o + 2 * (size_t)N + j]};
                    double w[3];
                    soc_apply(W.eta3[cc], &W.wb3[3 * cc], u, w, 3, true);
                    const int r0 = K.o3 + 3 * cc;
                    B[pk(R, r0, j)] = w[0]; B[pk(R, r0 + 1, j)] = w[1]; B[pk(R, r0 + 2, j)] = w[2];
                }
            }
            if (K.big) {
#pragma omp parallel
                {
                    vec u(K.big), w(K.big);
#pragma omp for schedule(static)
                    for (int j = 0; j < N; ++j) {
                        for (int i = 0; i < K.big; ++i) u[i] = G[(size_t)(K.ob + i) * N + j];
                        soc_apply(W.etab, W.wbb.data(), u.data(), w.data(), K.big, true);
                        for (int i = 0; i < K.big; ++i) B[pk(R, K.ob + i, j)] = w[i];
                    }
                }
            }
        }
        gram(B.data(), R, N, H.data());
        chol_fixes += chol_piv(H.data(), N);
        t_factor += omp_get_wtime() - t0;
    };
    // [0 G'; G -W^2][dx; dz] = [bx; bz] for ONE right-hand side; dz explicit, refined by PCG on the exact operator
    double plain_norm = 0;               // ||bx - G'dz|| of the last plain solve (the corrector's residual guard)
    auto kkt_solve = [&](const double* bx, const double* bz, double* dx, double* dz, double* gdx, bool plain = false) {
        vec wbz(R), rhs(N), r(N), zz(N), p(N), Gp(R), Wp(R), Hp(N);
        winv2(bz, wbz.data());
        D.mulT(wbz.data(), rhs.data());
        for (int j = 0; j < N; ++j) rhs[j] += bx[j];
        cho_solve(H.data(), N, rhs.data(), dx);
        D.mul(dx, gdx);
        winv2(gdx, dz);
        for (int i = 0; i < R; ++i) dz[i] -= wbz[i];
        D.mulT(dz, r.data());
        for (int j = 0; j < N; ++j) r[j] = bx[j] - r[j];
        if (plain) { plain_norm = nrm2(r.data(), N); return; }       // (the corrector's solve: no sweeps, nothing for the controller)
        vec norms{nrm2(r.data(), N)};
        if (nsweep > 0) {
            cho_solve(H.data(), N, r.data(), zz.data());
            p = zz;
            double rz = dot(r.data(), zz.data(), N);
            for (int it = 0; it < nsweep; ++it) {
                D.mul(p.data(), Gp.data());
                winv2(Gp.data(), Wp.data());
                D.mulT(Wp.data(), Hp.data());
                const double pHp = dot(p.data(), Hp.data(), N), al = pHp > 0 ? rz / pHp : 0.0;
                for (int j = 0; j < N; ++j) { dx[j] += al * p[j]; r[j] -= al * Hp[j]; }
                for (int i = 0; i < R; ++i) { gdx[i] += al * Gp[i]; dz[i] += al * Wp[i]; }
                norms.push_back(nrm2(r.data(), N));
                cho_solve(H.data(), N, r.data(), zz.data());
                const double rzn = dot(r.data(), zz.data(), N), be = rz > 0 ? rzn / rz : 0.0;
                for (int j = 0; j < N; ++j) p[j] = zz[j] + be * p[j];
                rz = rzn;
            }
        }
        sweep_log.push_back(norms);
    };
    auto next_sweeps = [&](double tol) {
        int need = 0;
        for (const vec& nl : sweep_log) {
            int k = -1;
            for (size_t i = 0; i < nl.size(); ++i)
                if (nl[i] <= tol) { k = int(i); break; }
            if (k < 0) return std::min(MAX_SWEEPS, nsweep + 1);
            need = std::max(need, k);
        }
        return need;
    };
    // ---- initial point (W = I) ---------------------------------------------------------------------------------
    vec x(N), s(R), z(R), zero_n(N, 0.0), zero_r(R, 0.0), negc(N), gd(R), zt(R), xt(N);
    for (int j = 0; j < N; ++j) negc[j] = -c[j];
    factor();
    kkt_solve(zero_n.data(), h, x.data(), zt.data(), gd.data());               // min ||Gx - h||, s = h - Gx
    for (int i = 0; i < R; ++i) s[i] = -zt[i];
    double ts = min_residual(K, s.data());
    if (ts >= -1e-8 * std::max(1.0, nrm2(s.data(), R))) add_e(K, s.data(), 1.0 + ts);
    kkt_solve(negc.data(), zero_r.data(), xt.data(), z.data(), gd.data());     // G'z = -c, least norm
    double tz = min_residual(K, z.data());
    if (tz >= -1e-8 * std::max(1.0, nrm2(z.data(), R))) add_e(K, z.data(), 1.0 + tz);
    unit = false;
    double tau = 1, kappa = 1;
    int status = ST_MAXIT, it = 0, wall = 0, fixes_seen = 0;
    double pcost = 0, dcost = 0, gap = 0, relgap = 0, pres = 0, dres = 0, best_merit = 1e300;
    double best_info[6] = {0, 0, 0, 0, 0, 0};
    vec xbest;
    int first_opt = -1;                  // end game: the first iteration whose iterate met the stopping rule ...
    double opt_merit = 1e300, opt_info[6] = {0, 0, 0, 0, 0, 0};
    vec xopt;                            // ... and the best such iterate
    vec rx(N), rz(R), lam(R), x1(N), z1(R), g1(R), x2(N), z2(R), g2(R), bxa(N), bza(R), wz1(R), dsa(R), dza(R), dssa(R), wdza(R),
        ll(R), dsc(R), lds(R), wlds(R), bxc(N), bzc(R), ds(R), dz(R), dss(R), wdz(R), dxv(N), pr(R), bzk(R), xk(N), zk(R), gk(R), dsk(R), dzk(R),
        dssk(R), wdzk(R);
    for (it = 0; it <= max_iter; ++it) {
        D.mulT(z.data(), rx.data());
        for (int j = 0; j < N; ++j) rx[j] += c[j] * tau;
        D.mul(x.data(), rz.data());
        for (int i = 0; i < R; ++i) rz[i] += s[i] - h[i] * tau;
        const double cx = dot(c, x.data(), N), hz = dot(h, z.data(), R), rt = kappa + cx + hz, sz = dot(s.data(), z.data(), R);
        const double mu = (sz + kappa * tau) / (K.degree + 1);
        pcost = cx / tau; dcost = -hz / tau; gap = sz / (tau * tau);
        pres = nrm2(rz.data(), R) / tau / nrm_h; dres = nrm2(rx.data(), N) / tau / nrm_c;
        const double den = std::max(std::fabs(pcost), std::fabs(dcost));
        relgap = den > 0 ? gap / den : std::numeric_limits<double>::infinity();
        double hresx = 0, hresz = 0;
        for (int j = 0; j < N; ++j) { const double v = rx[j] - c[j] * tau; hresx += v * v; }
        for (int i = 0; i < R; ++i) { const double v = rz[i] + h[i] * tau; hresz += v * v; }
        const double pinf = hz < 0 ? std::sqrt(hresx) / (-hz) : 1e300, dinf = cx < 0 ? std::sqrt(hresz) / (-cx) : 1e300;
        const bool finite = std::isfinite(pres) && std::isfinite(dres) && std::isfinite(gap) && tau > 0;
        if (finite && pres <= feastol && dres <= feastol && (gap <= abstol || relgap <= reltol)) {
            // end game (conic_ipm.py): keep the best iterate that meets the rule, go on until the gap measures are POLISH times
            // below the tolerances or POLISH_MAX more iterations have passed
            const double merit_o = std::min(relgap / reltol, gap / std::max(abstol, 1e-300));
            if (first_opt < 0 || merit_o < opt_merit) {
                opt_merit = merit_o;
                xopt.assign(N, 0.0);
                for (int j = 0; j < N; ++j) xopt[j] = x[j] / tau;
                const double oi[6] = {pcost, dcost, gap, relgap, pres, dres};
                std::memcpy(opt_info, oi, sizeof(oi));
            }
            if (first_opt < 0) first_opt = it;
            if (gap <= POLISH * abstol || relgap <= POLISH * reltol) { status = ST_OPTIMAL; break; }
        }
        if (first_opt >= 0 && it >= first_opt + POLISH_MAX) { status = ST_OPTIMAL; break; }      // (whether or not this iterate still meets the rule)
        {   // final approach and end game: POLISH_SWEEPS sweeps on top of the controller's count (conic_ipm.py)
            const bool approach = finite && (gap <= POLISH_APPROACH * abstol || relgap <= POLISH_APPROACH * reltol);
            nsweep = std::min(MAX_SWEEPS, nsweep_ctl + ((approach || first_opt >= 0) ? POLISH_SWEEPS : 0));
        }
        if (!finite) { status = ST_NUMERICAL; break; }
        const bool collapsed = kappa / tau >= 1e6;
        if (first_opt < 0 && (pinf <= feastol || (collapsed && pinf <= 1e-5))) { status = ST_PINF; break; }
        if (first_opt < 0 && (dinf <= feastol || (collapsed && dinf <= 1e-5))) { status = ST_DINF; break; }
        if (pres <= INACC_FEAS && dres <= INACC_FEAS) {
            const double merit = std::min(relgap, gap / std::max(abstol, 1e-300) * reltol);
            if (merit < best_merit) {
                best_merit = merit;
                xbest.assign(N, 0.0);
                for (int j = 0; j < N; ++j) xbest[j] = x[j] / tau;
                const double bi[6] = {pcost, dcost, gap, relgap, pres, dres};
                std::memcpy(best_info, bi, sizeof(bi));
            }
        }
        if (it == max_iter) break;
        const int last_fixes = chol_fixes - fixes_seen;
        fixes_seen = chol_fixes;
        wall = (last_fixes > 0 && (pres > INACC_FEAS || dres > INACC_FEAS)) ? wall + 1 : 0;
        if (wall >= WALL_ITERS) { status = ST_NUMERICAL; break; }
        W.build(K, s, z);
        W.apply(z.data(), lam.data(), false);
        factor();
        sweep_log.clear();
        // constant system [x1 z1]: bx = -c, bz = h ; affine system: bx = -rx, bz = s - rz
        kkt_solve(negc.data(), h, x1.data(), z1.data(), g1.data());
        for (int j = 0; j < N; ++j) bxa[j] = -rx[j];
        for (int i = 0; i < R; ++i) bza[i] = s[i] - rz[i];
        kkt_solve(bxa.data(), bza.data(), x2.data(), z2.data(), g2.data());
        W.apply(z1.data(), wz1.data(), false);
        const double den_t = kappa / tau + dot(wz1.data(), wz1.data(), R);
        double dtau = 0, dkap = 0;
        auto direction = [&](double sigma, double dk_c, const double* xx2, const double* zz2, const double* gg2, double* dsv, double* dzv,
                             double* dssv, double* wdzv) {
            const double bt = -(1 - sigma) * rt;
            dtau = (dk_c / tau - bt + dot(c, xx2, N) + dot(h, zz2, R)) / den_t;
            for (int i = 0; i < R; ++i) {
                dzv[i] = zz2[i] + dtau * z1[i];
                dsv[i] = -(1 - sigma) * rz[i] - gg2[i] - dtau * (g1[i] - h[i]);
            }
            W.apply(dzv, wdzv, false);
            W.apply(dsv, dssv, true);
            dkap = (dk_c - kappa * dtau) / tau;
        };
        auto step_of = [&](const double* dssv, const double* wdzv, double frac) {
            const double t = std::max(std::max(0.0, max_step(K, lam.data(), dssv)),
                                      std::max(max_step(K, lam.data(), wdzv), std::max(-dtau / tau, -dkap / kappa)));
            return t == 0.0 ? 1.0 : std::min(1.0, frac / t);
        };
        direction(0.0, -kappa * tau, x2.data(), z2.data(), g2.data(), dsa.data(), dza.data(), dssa.data(), wdza.data());
        const double dta = dtau, dka = dkap;
        const double alpha_a = step_of(dssa.data(), wdza.data(), 1.0);
        const double sigma = std::min((1 - alpha_a) * (1 - alpha_a) * (1 - alpha_a), (K.l > 0 && K.big == 0) ? SIGMA_MAX_CORR : SIGMA_MAX);
        cone_prod(K, lam.data(), lam.data(), ll.data());
        cone_prod(K, dssa.data(), wdza.data(), pr.data());
        for (int i = 0; i < R; ++i) dsc[i] = -ll[i] - pr[i];
        add_e(K, dsc.data(), sigma * mu);
        const double dk_c = sigma * mu - kappa * tau - dka * dta;
        cone_div(K, lam.data(), dsc.data(), lds.data());
        W.apply(lds.data(), wlds.data(), false);
        for (int j = 0; j < N; ++j) bxc[j] = -(1 - sigma) * rx[j];
        for (int i = 0; i < R; ++i) bzc[i] = -(1 - sigma) * rz[i] - wlds[i];
        kkt_solve(bxc.data(), bzc.data(), x2.data(), z2.data(), g2.data());
        direction(sigma, dk_c, x2.data(), z2.data(), g2.data(), ds.data(), dz.data(), dss.data(), wdz.data());
        double alpha = step_of(dss.data(), wdz.data(), STEP);
        if (K.l > 0) {
            // one centrality corrector (conic_ipm.py: CORR_*): trial step alpha + CORR_DELTA, the complementarity products
            // projected onto [CORR_BMIN, CORR_BMAX] sigma mu -- the orthant rows' products and the two eigenvalues of the big
            // cone's Jordan product --, one more solve with the factorisation at hand, the corrected direction taken when its
            // step is CORR_ACCEPT times longer
            const double at = std::min(1.0, alpha + CORR_DELTA), mut = sigma * mu, dtau0 = dtau, dkap0 = dkap;
            for (int i = 0; i < R; ++i) bzk[i] = 0.0;
            for (int i = 0; i < K.l; ++i) {
                const double v = (lam[i] + at * dss[i]) * (lam[i] + at * wdz[i]);
                double tt = std::min(std::max(v, CORR_BMIN * mut), CORR_BMAX * mut) - v;
                tt = std::max(tt, -CORR_BMAX * mut);
                bzk[i] = -W.wl[i] * (tt / lam[i]);
            }
            if (K.big > 0) {
                // t in the Jordan frame of v = (lam + at dss) o (lam + at wdz), then bz = -W (lam \ t)      (ll, pr, dsc: scratch)
                const int off = K.ob, d = K.big;
                for (int i = 0; i < R; ++i) dsc[i] = 0.0;
                double* u = &ll[off];
                double* w = &pr[off];
                for (int i = 0; i < d; ++i) { u[i] = lam[off + i] + at * dss[off + i]; w[i] = lam[off + i] + at * wdz[off + i]; }
                double* v = &dsc[off];
                soc_prod(u, w, v, d);
                double n1 = 0;
                for (int i = 1; i < d; ++i) n1 += v[i] * v[i];
                n1 = std::sqrt(n1);
                const double e1 = v[0] + n1, e2 = v[0] - n1;
                const double d1 = std::max(std::min(std::max(e1, CORR_BMIN * mut), CORR_BMAX * mut) - e1, -CORR_BMAX * mut);
                const double d2 = std::max(std::min(std::max(e2, CORR_BMIN * mut), CORR_BMAX * mut) - e2, -CORR_BMAX * mut);
                const double f = 0.5 * (d1 - d2) / (n1 > 0 ? n1 : 1.0);
                v[0] = 0.5 * (d1 + d2);
                for (int i = 1; i < d; ++i) v[i] = f * v[i];
                cone_div(K, lam.data(), dsc.data(), pr.data());
                W.apply(pr.data(), ll.data(), false);
                for (int i = K.ob; i < R; ++i) bzk[i] = -ll[i];
            }
            kkt_solve(zero_n.data(), bzk.data(), xk.data(), zk.data(), gk.data(), true);
            for (int j = 0; j < N; ++j) xk[j] += x2[j];
            for (int i = 0; i < R; ++i) { zk[i] += z2[i]; gk[i] += g2[i]; }
            direction(sigma, dk_c, xk.data(), zk.data(), gk.data(), dsk.data(), dzk.data(), dssk.data(), wdzk.data());
            const double alpha_c = step_of(dssk.data(), wdzk.data(), STEP);
            if (plain_norm <= std::max(REFTOL * nrm_c, CORR_ETA * nrm2(rx.data(), N)) && alpha_c >= CORR_ACCEPT * alpha) {
                alpha = alpha_c;
                x2.swap(xk); ds.swap(dsk); dz.swap(dzk);
            } else { dtau = dtau0; dkap = dkap0; }
        }
        nsweep_ctl = next_sweeps(std::max(REFTOL * nrm_c, REFETA * nrm2(rx.data(), N)));   // forcing term, see conic_ipm.py
        for (int j = 0; j < N; ++j) x[j] += alpha * (x2[j] + dtau * x1[j]);
        for (int i = 0; i < R; ++i) { s[i] += alpha * ds[i]; z[i] += alpha * dz[i]; }
        tau += alpha * dtau; kappa += alpha * dkap;
        if (!(std::isfinite(tau) && tau > 0)) { status = ST_NUMERICAL; break; }
    }
    for (int j = 0; j < N; ++j) x_out[j] = x[j] / tau;
    double out[6] = {pcost, dcost, gap, relgap, pres, dres};
    if (first_opt >= 0) {                // an iterate met the stopping rule: the best of them is the answer, however the end game ended
        status = ST_OPTIMAL;
        std::memcpy(out, opt_info, sizeof(out));
        std::memcpy(x_out, xopt.data(), sizeof(double) * N);
    } else if ((status == ST_MAXIT || status == ST_NUMERICAL) && !xbest.empty() && best_info[4] <= INACC_FEAS && best_info[5] <= INACC_FEAS &&
        (best_info[3] <= INACC_GAP || best_info[2] <= abstol)) {
        status = ST_INACC;
        std::memcpy(out, best_info, sizeof(out));
        std::memcpy(x_out, xbest.data(), sizeof(double) * N);
    }
    info_out[0] = status; info_out[1] = it;
    for (int q = 0; q < 6; ++q) info_out[2 + q] = out[q];
    info_out[8] = chol_fixes; info_out[9] = t_factor; info_out[10] = omp_get_wtime() - t_begin;
    info_out[11] = omp_get_max_threads();
    return status;
}

}  // extern "C"
