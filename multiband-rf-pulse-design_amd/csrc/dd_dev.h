// Double-double ("dd", an unevaluated sum hi + lo of two doubles, ~32 significant digits) device
// arithmetic for the extended-precision KKT solve (ddkkt.hip).  Error-free transformations after
// Dekker (1971) / Knuth TAOCP vol. 2 4.2.2 and the dd algorithms of Hida, Li & Bailey (ARITH-15,
// 2001).  gfx950 has a full-rate v_fma_f64, so TwoProd is two instructions.
//
// The translation unit that includes this must not let the compiler re-associate or contract the
// sums (hipcc defaults to -ffp-contract=fast, which only fuses a*b+c -- harmless for TwoSum, which
// has no products; the product terms below ask for fma explicitly).
#pragma once
#include <hip/hip_runtime.h>

namespace mbfir {

struct dd {
    double h, l;
};

__device__ __forceinline__ dd dd_make(double h, double l = 0.0) { dd r; r.h = h; r.l = l; return r; }
// The sums of the error-free transformations must stay separate roundings: contraction is switched off inside
// them (an add without the `contract` flag is never fused with a product of the caller, inlined or not).
__device__ __forceinline__ dd two_sum(double a, double b) {
#pragma clang fp contract(off)
    const double s = a + b, bb = s - a;
    return dd_make(s, (a - (s - bb)) + (b - bb));
}
__device__ __forceinline__ dd quick_two_sum(double a, double b) {
#pragma clang fp contract(off)
    const double s = a + b;
    return dd_make(s, b - (s - a));
}
__device__ __forceinline__ dd two_prod(double a, double b) {
#pragma clang fp contract(off)
    const double p = a * b;
    return dd_make(p, __builtin_fma(a, b, -p));
}
__device__ __forceinline__ dd dd_neg(dd a) { return dd_make(-a.h, -a.l); }
__device__ __forceinline__ dd dd_add(dd a, dd b) {
#pragma clang fp contract(off)
    dd s = two_sum(a.h, b.h);
    const dd t = two_sum(a.l, b.l);
    s.l += t.h;
    s = quick_two_sum(s.h, s.l);
    s.l += t.l;
    return quick_two_sum(s.h, s.l);
}
__device__ __forceinline__ dd dd_sub(dd a, dd b) { return dd_add(a, dd_neg(b)); }
__device__ __forceinline__ dd dd_add_d(dd a, double b) {
#pragma clang fp contract(off)
    dd s = two_sum(a.h, b);
    s.l += a.l;
    return quick_two_sum(s.h, s.l);
}
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
    dd p = two_prod(a.h, b.h);
    p.l = __builtin_fma(a.h, b.l, __builtin_fma(a.l, b.h, p.l));
    return quick_two_sum(p.h, p.l);
}
__device__ __forceinline__ dd dd_mul_d(dd a, double b) {
    dd p = two_prod(a.h, b);
    p.l = __builtin_fma(a.l, b, p.l);
    return quick_two_sum(p.h, p.l);
}
// acc - a*b, a and b dd
__device__ __forceinline__ dd dd_fnma(dd acc, dd a, dd b) { return dd_sub(acc, dd_mul(a, b)); }
__device__ __forceinline__ dd dd_div(dd a, dd b) {
    const double q1 = a.h / b.h;
    dd r = dd_sub(a, dd_mul_d(b, q1));
    const double q2 = r.h / b.h;
    r = dd_sub(r, dd_mul_d(b, q2));
    const double q3 = r.h / b.h;
    return dd_add_d(quick_two_sum(q1, q2), q3);
}
__device__ __forceinline__ dd dd_sqrt(dd a) {
    // Karp & Markstein: sqrt(a) = a x + [a - (a x)^2] x / 2 with x = 1/sqrt(a.h)
    // The error of the result is 1.5 delta^2 for a relative error delta of x; the compiler may turn 1/sqrt into
    // the hardware's reciprocal-square-root approximation (delta ~ 3e-9 measured on gfx950), so x gets one
    // Newton step of its own first.
    if (!(a.h > 0.0)) return dd_make(0.0, 0.0);
    double x = 1.0 / sqrt(a.h);
    x = x * (1.5 - 0.5 * a.h * x * x);
    const double ax = a.h * x;
    const dd e = dd_sub(a, two_prod(ax, ax));
    return two_sum(ax, e.h * (x * 0.5));
}
__device__ __forceinline__ dd dd_shfl(dd a, int src) {
    return dd_make(__shfl(a.h, src, 64), __shfl(a.l, src, 64));
}
__device__ __forceinline__ dd dd_shfl_down(dd a, int off) {
    return dd_make(__shfl_down(a.h, off, 64), __shfl_down(a.l, off, 64));
}
__device__ __forceinline__ dd dd_wave_sum(dd v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = dd_add(v, dd_shfl_down(v, o));
    return v;
}

}  // namespace mbfir
