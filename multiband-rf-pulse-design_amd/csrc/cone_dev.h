// Device-side cone algebra for K = R_+^l x Q_3^nq3 x Q_big (Nesterov-Todd scaling).
// Thread-per-cone functions for the LP rows and the 3-dimensional cones; the single large cone
// (dimension 1+2n) is handled by one workgroup with block reductions (bigcone_* kernels in
// solver.hip).  Formulas follow the published NT scaling for second-order cones:
//   sbar = s/sqrt(s'Js), zbar = z/sqrt(z'Jz), gamma = sqrt((1+sbar'zbar)/2),
//   wbar = (sbar + J zbar)/(2 gamma), eta = (s'Js / z'Jz)^(1/4),
//   W = eta [w0, w1'; w1, I + w1 w1'/(1+w0)],  W^-2 = eta^-2 (2 (Jw)(Jw)' - J),  lambda = W z.
#pragma once
#include <hip/hip_runtime.h>

namespace mbfir {

__device__ __forceinline__ double jres(double v0, double n1) { return (v0 - n1) * (v0 + n1); }

struct Soc3 {
    double eta, w0, w1, w2;
};

__device__ __forceinline__ Soc3 soc3_scaling(const double s[3], const double z[3]) {
    double ns = sqrt(s[1] * s[1] + s[2] * s[2]), nz = sqrt(z[1] * z[1] + z[2] * z[2]);
    double a = sqrt(jres(s[0], ns)), b = sqrt(jres(z[0], nz));
    double sb0 = s[0] / a, sb1 = s[1] / a, sb2 = s[2] / a;
    double zb0 = z[0] / b, zb1 = z[1] / b, zb2 = z[2] / b;
    double gamma = sqrt((1.0 + (sb0 * zb0 + sb1 * zb1 + sb2 * zb2)) / 2.0);
    Soc3 W;
    W.w0 = (sb0 + zb0) / (2 * gamma);
    W.w1 = (sb1 - zb1) / (2 * gamma);
    W.w2 = (sb2 - zb2) / (2 * gamma);
    W.eta = sqrt(a / b);
    return W;
}

// out = W u (inverse=false) or W^-1 u (inverse=true)
__device__ __forceinline__ void soc3_apply(const Soc3& W, const double u[3], double out[3], bool inverse) {
    double dot = W.w1 * u[1] + W.w2 * u[2];
    if (inverse) {
        double f = -u[0] + dot / (1 + W.w0);
        out[0] = (W.w0 * u[0] - dot) / W.eta;
        out[1] = (u[1] + f * W.w1) / W.eta;
        out[2] = (u[2] + f * W.w2) / W.eta;
    } else {
        double f = u[0] + dot / (1 + W.w0);
        out[0] = (W.w0 * u[0] + dot) * W.eta;
        out[1] = (u[1] + f * W.w1) * W.eta;
        out[2] = (u[2] + f * W.w2) * W.eta;
    }
}

// M = W^-2 (symmetric 3x3): m[0..5] = M00 M01 M02 M11 M12 M22
__device__ __forceinline__ void soc3_inv2(const Soc3& W, double m[6]) {
    double u0 = W.w0, u1 = -W.w1, u2 = -W.w2, e2 = 1.0 / (W.eta * W.eta);
    m[0] = (2 * u0 * u0 - 1) * e2;
    m[1] = 2 * u0 * u1 * e2;
    m[2] = 2 * u0 * u2 * e2;
    m[3] = (2 * u1 * u1 + 1) * e2;
    m[4] = 2 * u1 * u2 * e2;
    m[5] = (2 * u2 * u2 + 1) * e2;
}
__device__ __forceinline__ double sym3(const double m[6], int a, int b) {
    const int idx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    return m[idx[a][b]];
}
__device__ __forceinline__ void soc3_inv2_apply(const Soc3& W, const double v[3], double out[3]) {
    double u0 = W.w0, u1 = -W.w1, u2 = -W.w2, e2 = 1.0 / (W.eta * W.eta);
    double uv = u0 * v[0] + u1 * v[1] + u2 * v[2];
    out[0] = (2 * u0 * uv - v[0]) * e2;
    out[1] = (2 * u1 * uv + v[1]) * e2;
    out[2] = (2 * u2 * uv + v[2]) * e2;
}

__device__ __forceinline__ void soc3_prod(const double u[3], const double v[3], double out[3]) {
    out[0] = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
    out[1] = u[0] * v[1] + v[0] * u[1];
    out[2] = u[0] * v[2] + v[0] * u[2];
}
// x with lam o x = d
__device__ __forceinline__ void soc3_div(const double lam[3], const double d[3], double out[3]) {
    double a = jres(lam[0], sqrt(lam[1] * lam[1] + lam[2] * lam[2]));
    double ld = lam[1] * d[1] + lam[2] * d[2];
    out[0] = (lam[0] * d[0] - ld) / a;
    out[1] = (d[1] - out[0] * lam[1]) / lam[0];
    out[2] = (d[2] - out[0] * lam[2]) / lam[0];
}
// ||rho_1|| - rho_0 with rho = T(lam) d, T lam = e   (step bound: alpha <= 1/that)
__device__ __forceinline__ double soc3_step(const double lam[3], const double d[3]) {
    double a = sqrt(jres(lam[0], sqrt(lam[1] * lam[1] + lam[2] * lam[2])));
    double lb0 = lam[0] / a, lb1 = lam[1] / a, lb2 = lam[2] / a;
    double dot = lb1 * d[1] + lb2 * d[2];
    double rho0 = (lb0 * d[0] - dot) / a;
    double f = -d[0] + dot / (1 + lb0);
    double r1 = (d[1] + f * lb1) / a, r2 = (d[2] + f * lb2) / a;
    return sqrt(r1 * r1 + r2 * r2) - rho0;
}

}  // namespace mbfir
