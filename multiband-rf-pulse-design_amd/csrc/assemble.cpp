// Host-side problem assembly of the four convex FIR designers into the
// structured TrigProgram (program.h).  O(m) scalar work per design; the dense
// trig matrix itself is generated on the device (solver.hip).
//
// Follows the reference's rules (cited per function); the row order is the one
// documented in oracle/assemble.py so tests can compare the two row by row.
#include "program.h"
#include <algorithm>
#include <cmath>
#include <complex>

namespace mbfir {
namespace {

const double PI = 3.14159265358979323846;

// MATLAB linspace.m: y = d1 + (0:n1).*(d2-d1)./n1 with exact end points.
std::vector<double> linspace(double d1, double d2, int n) {
    std::vector<double> y(n);
    int n1 = n - 1;
    for (int k = 0; k < n; ++k) y[k] = d1 + (double(k) * (d2 - d1)) / double(n1);
    if (n > 0) { y[0] = d1; y[n - 1] = d2; }
    return y;
}

struct Bands {
    std::vector<double> w;          // sorted grid incl. band edges
    std::vector<int> idx_band;      // concatenated in band order
    std::vector<double> amp, dev;   // per in-band sample
    std::vector<int> idx_tran;
};

// fir_ap_cvx.m:47-69 == fir_qp_cvx.m:37-63 == ss/fir_linprog.m:97-132 == ss/fir_qprog_phs.m:219-223,239,278-280
void split_bands(double lo, double hi, int m0, const std::vector<double>& fpi, const double* a,
                 const double* d, Bands& B) {
    B.w = linspace(lo, hi, m0);
    B.w.insert(B.w.end(), fpi.begin(), fpi.end());
    std::sort(B.w.begin(), B.w.end());
    int nband = int(fpi.size()) / 2;
    std::vector<char> inband(B.w.size(), 0);
    for (int b = 0; b < nband; ++b) {
        double flo = fpi[2 * b], fhi = fpi[2 * b + 1];
        for (size_t i = 0; i < B.w.size(); ++i) {
            if (B.w[i] >= flo && B.w[i] <= fhi) {
                B.idx_band.push_back(int(i));
                inband[i] = 1;
                double amp = (flo == fhi || a == nullptr)
                                 ? (a ? a[2 * b] : 0.0)
                                 : a[2 * b] + (a[2 * b + 1] - a[2 * b]) * ((B.w[i] - flo) / (fhi - flo));
                B.amp.push_back(amp);
                B.dev.push_back(d ? d[b] : 0.0);
            }
        }
    }
    for (size_t i = 0; i < B.w.size(); ++i)
        if (!inband[i]) B.idx_tran.push_back(int(i));
}

void tran_limits(const std::vector<double>& U_band, const std::vector<double>& L_band, double& U_tran,
                 double& L_tran) {
    // fir_ap_cvx.m:74-82: U = max(U_band), L = min(0, min(L_band))
    U_tran = *std::max_element(U_band.begin(), U_band.end());
    L_tran = std::min(0.0, *std::min_element(L_band.begin(), L_band.end()));
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// fir_ap_cvx.m:44-142,160-169
int assemble_ap(int n, int nband, const double* f, const double* a, const double* d, double obj,
                double peak, int grid_m, TrigProgram& P, std::string& err) {
    if (n < 2 || nband < 1) { err = "not enough input"; return -1; }
    if (obj < 0) { err = "invalid input of obj"; return -1; }            // :171-173
    const double epsilon = 1e-10;                                         // :40
    std::vector<double> fpi(2 * nband);
    for (int i = 0; i < 2 * nband; ++i) fpi[i] = f[i] * PI;               // :44
    int m0 = grid_m > 0 ? grid_m : 2 * n * 15;                            // :45-46
    Bands B;
    split_bands(-PI, PI, m0, fpi, a, d, B);
    if (B.idx_band.empty()) { err = "no grid sample inside any band"; return -1; }
    size_t nb = B.idx_band.size(), nt = B.idx_tran.size();
    std::vector<double> U_band(nb), L_band(nb);
    for (size_t i = 0; i < nb; ++i) { U_band[i] = B.amp[i] + B.dev[i]; L_band[i] = B.amp[i] - B.dev[i]; }
    double U_tran, L_tran;
    tran_limits(U_band, L_band, U_tran, L_tran);
    int m = int(nb + nt);
    P = TrigProgram();
    P.which = DES_AP; P.n = n; P.Nt = 2 * n - 1; P.Ne = 1; P.Mf = m; P.quad = false;
    P.w.resize(m);
    std::vector<double> U_b(m), L_b(m);
    for (size_t i = 0; i < nb; ++i) { P.w[i] = B.w[B.idx_band[i]]; U_b[i] = U_band[i]; L_b[i] = L_band[i]; }  // :86-91
    for (size_t i = 0; i < nt; ++i) { P.w[nb + i] = B.w[B.idx_tran[i]]; U_b[nb + i] = U_tran; L_b[nb + i] = L_tran; }
    for (int i = 0; i < m; ++i) {
        U_b[i] = U_b[i] * U_b[i];                                          // :105
        double lo = L_b[i] < 0 ? 0.0 : L_b[i];                             // :110-111
        lo = lo * lo;                                                      // :112
        if (lo < epsilon * epsilon) lo = epsilon * epsilon;                // :115-116
        L_b[i] = lo;
    }
    // A = [1, 2cos(w j), 2 sin(w j)], j=1..n-1                            // :100
    P.col_kind.resize(P.Nt); P.col_tau.resize(P.Nt); P.col_scale.resize(P.Nt);
    P.pcol.assign(P.Nt, 0); P.psign.assign(P.Nt, 0.0);
    P.col_kind[0] = 0; P.col_tau[0] = 0; P.col_scale[0] = 1;
    for (int j = 1; j < n; ++j) {
        P.col_kind[j] = 0; P.col_tau[j] = j; P.col_scale[j] = 2;
        P.col_kind[n - 1 + j] = 1; P.col_tau[n - 1 + j] = j; P.col_scale[n - 1 + j] = 2;
    }
    double sqmin = 1e300;
    for (int i = 0; i < m; ++i) sqmin = std::min(sqmin, std::sqrt(U_b[i]));
    std::vector<int> idx_stop;
    for (int i = 0; i < m; ++i)
        if (std::sqrt(U_b[i]) < sqmin + 1e-2) idx_stop.push_back(i);       // :125
    for (int i = 0; i < m; ++i) P.add_row(i, -1, 1.0, 0.0, 0, 0, 0, U_b[i]);       // A x <= U^2   :119-120,164
    for (int i = 0; i < m; ++i) P.add_row(i, -1, -1.0, 0.0, 0, 0, 0, -L_b[i]);     // -A x <= -L^2
    for (int i : idx_stop) P.add_row(i, -1, 1.0, 0.0, -1.0, 0, 0, 0.0);            // A_stop x <= ripple_stop :165
    P.add_row(-1, 0, 1.0, 0.0, 0, 0, 0, n * peak);                                  // |x(1)| <= n*Peak :136,167
    P.add_row(-1, 0, -1.0, 0.0, 0, 0, 0, n * peak);
    P.l = int(P.h.size());
    for (int i = 2; i <= n; ++i) {                                                  // :137-142,166-168
        P.add_row(-1, -1, 0.0, 0.0, 0, 0, 0, (n - i + 1) * peak);
        P.add_row(-1, i - 1, -1.0, 0.0, 0, 0, 0, 0.0);
        P.add_row(-1, n + i - 2, -1.0, 0.0, 0, 0, 0, 0.0);
    }
    P.nq3 = n - 1; P.big = 0; P.R = int(P.h.size());
    P.c.assign(P.N(), 0.0);
    P.c[0] = 1.0; P.c[P.Nt] = obj;                                                  // :162
    return 0;
}

// ---------------------------------------------------------------------------------------------
// fir_qp_cvx.m:34-139,145-191
int assemble_qp(int n, int nband, const double* f, const double* a, const double* d, double kquad,
                const double* obj, int nobj, int grid_m, TrigProgram& P, std::string& err) {
    if (n < 1 || nband < 1) { err = "not enough input"; return -1; }
    if (nobj != 1 && nobj != 2) { err = "invalid input of obj"; return -1; }        // :194-196
    std::vector<double> fpi(2 * nband);
    for (int i = 0; i < 2 * nband; ++i) fpi[i] = f[i] * PI;                         // :34
    int m0 = grid_m > 0 ? grid_m : n * 10;                                          // :35-36
    Bands B;
    split_bands(-PI, PI, m0, fpi, a, d, B);
    size_t mb = B.idx_band.size(), mt = B.idx_tran.size();
    bool modelB = nobj == 2;
    P = TrigProgram();
    P.which = DES_QP; P.n = n; P.Nt = 2 * n; P.Ne = modelB ? 3 : 2; P.Mf = int(mb + mt); P.quad = true;
    const int iD = 0, iE = modelB ? 1 : 0, iP = iE + 1;                             // y = [delta] E Peak
    P.w.resize(P.Mf);
    for (size_t i = 0; i < mb; ++i) P.w[i] = B.w[B.idx_band[i]];                    // wband :79
    for (size_t i = 0; i < mt; ++i) P.w[mb + i] = B.w[B.idx_tran[i]];               // wtran :80
    // r1 = [cos(w t), sin(w t)], r2 = [-sin(w t), cos(w t)], t = 0..n-1             // :99,107
    P.col_kind.resize(P.Nt); P.col_tau.resize(P.Nt); P.col_scale.assign(P.Nt, 1.0);
    P.pcol.resize(P.Nt); P.psign.resize(P.Nt);
    for (int k = 0; k < n; ++k) {
        P.col_kind[k] = 0; P.col_tau[k] = k; P.col_kind[n + k] = 1; P.col_tau[n + k] = k;
        P.pcol[k] = n + k; P.psign[k] = -1.0;
        P.pcol[n + k] = k; P.psign[n + k] = 1.0;
    }
    double dmax = d[0];
    for (int b = 1; b < nband; ++b) dmax = std::max(dmax, d[b]);
    auto ey3 = [&](int idx, double v, double* e) { e[0] = e[1] = e[2] = 0; if (idx >= 0) e[idx] = v; };
    double e[3];
    for (size_t i = 0; i < mb; ++i) {                                               // :150-152 / :175-177
        double wi = P.w[i];
        std::complex<double> Hd = B.amp[i] * std::exp(std::complex<double>(0, kquad * wi * wi - wi * (n - 1) / 2.0));  // :118
        if (modelB) { ey3(iD, -B.dev[i], e); P.add_row(-1, -1, 0, 0, e[0], e[1], e[2], 0.0); }
        else P.add_row(-1, -1, 0, 0, 0, 0, 0, B.dev[i]);
        P.add_row(int(i), -1, -1.0, 0.0, 0, 0, 0, -Hd.real());
        P.add_row(int(i), -1, 0.0, -1.0, 0, 0, 0, -Hd.imag());
    }
    for (size_t i = 0; i < mt; ++i) {                                               // :155-157 / :180-182
        P.add_row(-1, -1, 0, 0, 0, 0, 0, modelB ? 1.1 : 1 + dmax * 5);
        P.add_row(int(mb + i), -1, -1.0, 0.0, 0, 0, 0, 0.0);
        P.add_row(int(mb + i), -1, 0.0, -1.0, 0, 0, 0, 0.0);
    }
    for (int i = 0; i < n; ++i) {                                                   // :160-162
        ey3(iP, -1.0, e);
        P.add_row(-1, -1, 0, 0, e[0], e[1], e[2], 0.0);
        P.add_row(-1, i, -1.0, 0, 0, 0, 0, 0.0);
        P.add_row(-1, n + i, -1.0, 0, 0, 0, 0, 0.0);
    }
    P.l = 0; P.nq3 = int(mb + mt) + n;
    ey3(iE, -1.0, e);                                                               // norm(x) <= E_total :165
    P.add_row(-1, -1, 0, 0, e[0], e[1], e[2], 0.0);
    for (int j = 0; j < 2 * n; ++j) P.add_row(-1, j, -1.0, 0, 0, 0, 0, 0.0);
    P.big = 1 + 2 * n; P.R = int(P.h.size());
    P.c.assign(P.N(), 0.0);
    if (modelB) { P.c[P.Nt + iD] = 1.0; P.c[P.Nt + iE] = obj[0]; P.c[P.Nt + iP] = obj[1]; }   // :172
    else { P.c[P.Nt + iE] = 1.0; P.c[P.Nt + iP] = obj[0]; }                                    // :147
    return 0;
}

// ---------------------------------------------------------------------------------------------
// ss/fir_linprog.m:46-132,161-240
int assemble_linprog(int n, int nband, const double* f, const double* a, const double* d, int grid_m,
                     TrigProgram& P, std::string& err) {
    if (n < 1 || nband < 1) { err = "not enough input"; return -1; }
    std::vector<double> fpi(2 * nband);
    double fmin = 1e300;
    for (int i = 0; i < 2 * nband; ++i) { fpi[i] = f[i] * PI; fmin = std::min(fmin, fpi[i]); }  // :46
    bool real_filter = !(fmin < 0);                                                 // :47-51
    bool odd_filter = (n & 1) == 1;                                                 // :56-60
    if (!odd_filter)                                                                // :66-75
        for (int i = 0; i < 2 * nband; ++i)
            if (std::fabs(fpi[i]) == PI && a[i] == 1.0) { err = "n odd and frequency spec 1 at fs/2"; return 3; }
    int nhalf = (n + 1) / 2;                                                        // :79
    int nx = nhalf;
    if (!real_filter) nx = odd_filter ? 2 * nhalf - 1 : 2 * nhalf;                  // :82-88
    Bands B;
    if (real_filter) split_bands(0.0, PI, grid_m > 0 ? grid_m : 15 * n, fpi, a, d, B);        // :97-99
    else split_bands(-PI, PI, grid_m > 0 ? grid_m : 2 * 15 * n, fpi, a, d, B);                // :100-103
    if (B.idx_band.empty()) { err = "no grid sample inside any band"; return -1; }
    size_t nb = B.idx_band.size(), nt = B.idx_tran.size();
    std::vector<double> U_band(nb), L_band(nb);
    for (size_t i = 0; i < nb; ++i) { U_band[i] = B.amp[i] + B.dev[i]; L_band[i] = B.amp[i] - B.dev[i]; }
    double U_tran, L_tran;
    tran_limits(U_band, L_band, U_tran, L_tran);                                    // :166-174
    int m = int(nb + nt);
    P = TrigProgram();
    P.which = DES_LINPROG; P.n = n; P.Nt = nx; P.Ne = 0; P.Mf = m; P.quad = false;
    P.nhalf = nhalf; P.real_filter = real_filter; P.odd_filter = odd_filter;
    P.w.resize(m);
    for (size_t i = 0; i < nb; ++i) P.w[i] = B.w[B.idx_band[i]];                    // :178-183
    for (size_t i = 0; i < nt; ++i) P.w[nb + i] = B.w[B.idx_tran[i]];
    P.col_kind.assign(nx, 0); P.col_tau.assign(nx, 0.0); P.col_scale.assign(nx, 2.0);
    P.pcol.assign(nx, 0); P.psign.assign(nx, 0.0);
    if (odd_filter) {                                                               // :199-200,206-208
        P.col_scale[0] = 1.0;
        for (int j = 0; j < nhalf; ++j) P.col_tau[j] = j;
        if (!real_filter)
            for (int j = 1; j < nhalf; ++j) { P.col_kind[nhalf - 1 + j] = 1; P.col_tau[nhalf - 1 + j] = j; }
    } else {                                                                        // :202,210-211
        for (int j = 0; j < nhalf; ++j) P.col_tau[j] = j + 0.5;
        if (!real_filter)
            for (int j = 0; j < nhalf; ++j) { P.col_kind[nhalf + j] = 1; P.col_tau[nhalf + j] = j + 0.5; }
    }
    for (int i = 0; i < m; ++i) P.add_row(i, -1, 1.0, 0, 0, 0, 0, i < int(nb) ? U_band[i] : U_tran);     // :221-232
    for (int i = 0; i < m; ++i) P.add_row(i, -1, -1.0, 0, 0, 0, 0, -(i < int(nb) ? L_band[i] : L_tran));
    P.l = 2 * m; P.nq3 = 0; P.big = 0; P.R = 2 * m;
    // c = fmin = sum(A(idx_tran,:),1) (:240): a sum over rows of the trig matrix, so it is
    // evaluated here in plain double like the reference does.
    P.c.assign(nx, 0.0);
    for (size_t i = 0; i < nt; ++i) {
        double wi = P.w[nb + i];
        for (int j = 0; j < nx; ++j) {
            double arg = wi * P.col_tau[j];
            P.c[j] += P.col_scale[j] * (P.col_kind[j] ? std::sin(arg) : std::cos(arg));
        }
    }
    if (odd_filter)
        { P.c[0] = double(nt); }   // the column of ones sums exactly
    return 0;
}

// ---------------------------------------------------------------------------------------------
// ss/fir_qprog_phs.m:49-128,178-342 ; min 1/2 x'x restated as min t s.t. ||x|| <= t.
int assemble_qprog_phs(int n, int nband, const double* f, const double* ac_re, const double* ac_im,
                       const double* dc_re, const double* dc_im, int grid_m, TrigProgram& P,
                       std::string& err) {
    if (n < 1 || nband < 1) { err = "not enough input"; return -1; }
    typedef std::complex<double> cd;
    for (int b = 0; b < nband; ++b)                                                 // :53-57
        if (ac_re[2 * b] != ac_re[2 * b + 1] || ac_im[2 * b] != ac_im[2 * b + 1]) {
            err = "Does not support sloped bands"; return -1;
        }
    std::vector<double> a(nband), aphs(nband), d(nband), dphs(nband);
    for (int b = 0; b < nband; ++b) {
        a[b] = std::abs(cd(ac_re[2 * b], ac_im[2 * b]));                            // :61,64
        aphs[b] = std::arg(cd(a[b], 0.0));                                          // :65 angle(abs(.))
        d[b] = std::abs(cd(dc_re[b], dc_im[b]));                                    // :66
        dphs[b] = std::arg(cd(dc_re[b], dc_im[b]));                                 // :67
    }
    for (int b = 0; b < nband; ++b)                                                 // :74-80
        if ((a[b] + d[b]) * (a[b] - d[b]) < 0 && (a[b] != 0 || dphs[b] != 0)) {
            err = "Bands straddling 0 must have a = 0, angle(d) = 0"; return -1;
        }
    const double err_tol = 0.05;                                                    // :85
    for (int b = 0; b < nband; ++b)                                                 // :86-98
        if (a[b] != 0) {
            double magerr_inner = (a[b] - d[b]) * (1.0 / std::cos(dphs[b]) - 1);
            if (magerr_inner >= 2 * d[b]) dphs[b] = 0.99 * std::acos((a[b] - d[b]) / (a[b] + d[b]));
        }
    int n_phs_tran = int(std::ceil(2 * PI / std::acos(1 - err_tol)));               // :103
    double amax = 0;
    for (int b = 0; b < nband; ++b) amax = std::max(amax, a[b] + d[b]);             // :104
    std::vector<double> phs_tran;
    for (int k = 0; k <= n_phs_tran; ++k) phs_tran.push_back(double(k) / n_phs_tran * 2 * PI);   // :105
    std::vector<std::vector<double>> phs_band(nband);
    for (int b = 0; b < nband; ++b) {                                               // :106-123
        if (a[b] == 0) {
            for (int k = 0; k <= n_phs_tran; ++k) phs_band[b].push_back(double(k) / n_phs_tran * 2 * PI);
        } else {
            double phs_tol = std::acos(1 - (err_tol * 2 * d[b]));
            int n_phs = int(std::ceil(2 * dphs[b] / phs_tol));
            if (n_phs < 1) { err = "passband needs a non-zero phase ripple"; return -1; }
            for (int k = 0; k <= n_phs; ++k) phs_band[b].push_back((double(k) / n_phs * 2 - 1) * dphs[b] + aphs[b]);
            if ((a[b] + d[b]) >= amax * (1 - err_tol)) {
                phs_tran.push_back(aphs[b] - dphs[b]);
                phs_tran.push_back(aphs[b] + dphs[b]);
            }
        }
    }
    for (double& p : phs_tran) { p = std::fmod(p, 2 * PI); if (p < 0) p += 2 * PI; }        // :127
    phs_tran.push_back(0.0); phs_tran.push_back(2 * PI);                            // :128
    std::sort(phs_tran.begin(), phs_tran.end());
    phs_tran.erase(std::unique(phs_tran.begin(), phs_tran.end()), phs_tran.end());
    std::vector<double> fpi(2 * nband);
    for (int i = 0; i < 2 * nband; ++i) fpi[i] = f[i] * PI;                         // :178
    bool odd_filter = (n & 1) == 1;                                                 // :183-187
    if (!odd_filter)                                                                // :193-202
        for (int i = 0; i < 2 * nband; ++i)
            if (std::fabs(fpi[i]) == PI && std::abs(cd(ac_re[i], ac_im[i])) != 0) {
                err = "n odd and frequency spec non-zero at fs/2"; return 3;
            }
    int nhalf = (n + 1) / 2;                                                        // :206
    int m0 = grid_m > 0 ? grid_m : 2 * 15 * n;                                      // :213,218
    Bands B;
    split_bands(-PI, PI, m0, fpi, nullptr, nullptr, B);                             // :219-223
    P = TrigProgram();
    P.which = DES_QPROG_PHS; P.n = n; P.Nt = 2 * n; P.Ne = 1; P.Mf = int(B.w.size()); P.quad = true;
    P.w = B.w;
    // W = exp(-i w t): r1 = [cos(w t), sin(w t)], r2 = [-sin(w t), cos(w t)]      // :227-231
    P.col_kind.resize(P.Nt); P.col_tau.resize(P.Nt); P.col_scale.assign(P.Nt, 1.0);
    P.pcol.resize(P.Nt); P.psign.resize(P.Nt);
    for (int k = 0; k < n; ++k) {
        double t = odd_filter ? double(k - (nhalf - 1)) : double(k - nhalf) + 0.5;
        P.col_kind[k] = 0; P.col_tau[k] = t; P.col_kind[n + k] = 1; P.col_tau[n + k] = t;
        P.pcol[k] = n + k; P.psign[k] = -1.0; P.pcol[n + k] = k; P.psign[n + k] = 1.0;
    }
    // rows: in-phase  [real(W e^{-i phi}), -imag(.)]  = cos(phi) r1 + sin(phi) r2
    //       quadrature [imag(W e^{-i phi}),  real(.)] = -sin(phi) r1 + cos(phi) r2
    struct Row { int fr; double al, be, h; };
    std::vector<Row> Au, Al;
    for (int b = 0; b < nband; ++b) {                                               // :238-274
        std::vector<int> idx;
        for (size_t i = 0; i < B.w.size(); ++i)
            if (B.w[i] >= fpi[2 * b] && B.w[i] <= fpi[2 * b + 1]) idx.push_back(int(i));
        const std::vector<double>& pb = phs_band[b];
        double phs_diff = std::arg(std::exp(cd(0, pb[1])) * std::exp(cd(0, -pb[0])));      // :244
        double a_mid = (a[b] + d[b]) * std::cos(phs_diff / 2);                      // :246
        for (size_t k = 0; k + 1 < pb.size(); ++k) {                                // :247-253
            double phs_mid = pb[k] + phs_diff / 2;
            for (int i : idx) Au.push_back({i, std::cos(phs_mid), std::sin(phs_mid), a_mid});
        }
        if (a[b] != 0) {                                                            // :257-273
            for (int i : idx) Al.push_back({i, std::cos(aphs[b]), std::sin(aphs[b]), a[b] - d[b]});
            for (int i : idx) Au.push_back({i, -std::sin(pb.back()), std::cos(pb.back()), 0.0});
            for (int i : idx) Al.push_back({i, -std::sin(pb[0]), std::cos(pb[0]), 0.0});
        }
    }
    for (size_t k = 0; k + 1 < phs_tran.size(); ++k) {                              // :306-313
        double phs_diff = phs_tran[k + 1] - phs_tran[k];
        double phs_mid = phs_tran[k] + phs_diff / 2;
        for (int i : B.idx_tran) Au.push_back({i, std::cos(phs_mid), std::sin(phs_mid), amax * std::cos(phs_diff / 2)});
    }
    for (const Row& r : Au) P.add_row(r.fr, -1, r.al, r.be, 0, 0, 0, r.h);          // A=[Au;-Al] :318-319
    for (const Row& r : Al) P.add_row(r.fr, -1, -r.al, -r.be, 0, 0, 0, -r.h);
    P.l = int(P.h.size()); P.nq3 = 0;
    P.add_row(-1, -1, 0, 0, -1.0, 0, 0, 0.0);                                       // (t ; x) in Q_{2n+1}
    for (int j = 0; j < 2 * n; ++j) P.add_row(-1, j, -1.0, 0, 0, 0, 0, 0.0);
    P.big = 1 + 2 * n; P.R = int(P.h.size());
    P.c.assign(P.N(), 0.0);
    P.c[P.Nt] = 1.0;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// 1 for the rows that a row-sharded solve REPLICATES on every rank: LP rows without a frequency, the rows of Q3 cones none of
// whose rows has one, the big cone.  (Sums over the rows count them on rank 0 only.)
std::vector<int> replicated_rows(const TrigProgram& Q) {
    std::vector<int> rep(Q.R, 0);
    for (int r = 0; r < Q.l; ++r) rep[r] = Q.freq[r] < 0 ? 1 : 0;
    for (int c = 0; c < Q.nq3; ++c) {
        const int r0 = Q.l + 3 * c;
        const int v = (Q.freq[r0] < 0 && Q.freq[r0 + 1] < 0 && Q.freq[r0 + 2] < 0) ? 1 : 0;
        rep[r0] = rep[r0 + 1] = rep[r0 + 2] = v;
    }
    for (int r = Q.l + 3 * Q.nq3; r < Q.R; ++r) rep[r] = 1;
    return rep;
}

TrigProgram shard_program(const TrigProgram& Q, int rank, int size) {
    if (size <= 1) return Q;
    TrigProgram P;
    P.which = Q.which; P.n = Q.n; P.Nt = Q.Nt; P.Ne = Q.Ne; P.quad = Q.quad;
    P.col_kind = Q.col_kind; P.col_tau = Q.col_tau; P.col_scale = Q.col_scale; P.pcol = Q.pcol; P.psign = Q.psign;
    P.c = Q.c; P.nhalf = Q.nhalf; P.real_filter = Q.real_filter; P.odd_filter = Q.odd_filter;
    // The frequencies are dealt out by FOLDED PAIRS (round 4): +w and -w share cos(w t) and differ in the sign of sin(w t), and
    // the lattice kernels run one recurrence for both (solver.hip analyse_lattice) -- but only when both sit on the same rank;
    // dealing single frequencies i % size put the partners on different ranks and doubled every rank's recurrence work.
    // Entries of the list sorted by |w| -- a pair within 2 ulp of each other with opposite signs, or a single frequency (band
    // edges, one-sided grids) -- go to the ranks in turn, so every rank gets the same band / transition mix.
    std::vector<int> owner(Q.Mf, 0);
    {
        std::vector<int> order(Q.Mf);
        for (int i = 0; i < Q.Mf; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return std::fabs(Q.w[a]) < std::fabs(Q.w[b]); });
        double wmax = 1.0;
        for (double w : Q.w) wmax = std::max(wmax, std::fabs(w));
        const double tol = 2 * 2.2204460492503131e-16 * wmax;
        int entry = 0;
        for (int q = 0; q < Q.Mf; ++entry) {
            const int i = order[q];
            owner[i] = entry % size;
            if (q + 1 < Q.Mf) {
                const int j = order[q + 1];
                if (std::fabs(Q.w[j]) - std::fabs(Q.w[i]) <= tol && (Q.w[i] < 0.0) != (Q.w[j] < 0.0)) { owner[j] = entry % size; q += 2; continue; }
            }
            q += 1;
        }
    }
    std::vector<int> fmap(Q.Mf, -1);
    for (int i = 0; i < Q.Mf; ++i)
        if (owner[i] == rank) { fmap[i] = int(P.w.size()); P.w.push_back(Q.w[i]); }
    P.Mf = int(P.w.size());
    // rows / cones with a frequency go to the rank that holds it; those WITHOUT one (identity rows, spike / per-tap cones, the
    // big cone) are REPLICATED on every rank (round 4): x and y are replicated, so every rank computes them identically, holds
    // their scaling and can assemble and factorise the whole normal matrix itself; their contributions to sums over the rows
    // are counted on rank 0 only (DProg::rep / own in solver.hip)
    auto mine = [&](int f) { return f < 0 || owner[f] == rank; };
    auto copy_row = [&](int r) {
        int f = Q.freq[r];
        P.add_row(f < 0 ? -1 : fmap[f], Q.col[r], Q.alpha[r], Q.beta[r], Q.ey[3 * r], Q.ey[3 * r + 1], Q.ey[3 * r + 2], Q.h[r]);
    };
    for (int r = 0; r < Q.l; ++r)
        if (mine(Q.freq[r])) copy_row(r);
    P.l = int(P.h.size());
    for (int c = 0; c < Q.nq3; ++c) {
        int r0 = Q.l + 3 * c, f = -1;
        for (int a = 0; a < 3; ++a) if (Q.freq[r0 + a] >= 0) f = Q.freq[r0 + a];
        if (!mine(f)) continue;
        for (int a = 0; a < 3; ++a) copy_row(r0 + a);
        P.nq3++;
    }
    if (Q.big) {
        for (int r = Q.l + 3 * Q.nq3; r < Q.R; ++r) copy_row(r);
        P.big = Q.big;
    }
    P.R = int(P.h.size());
    return P;
}

}  // namespace mbfir
