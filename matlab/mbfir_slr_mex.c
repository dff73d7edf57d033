/*
 * mbfir_slr_mex.c -- MEX gateway for the inverse SLR step and the forward simulation of include/mbfir.h.
 *
 *   [o1_re, o1_im, o2_re, o2_im] = mbfir_slr_mex(op, ...)
 *     op 0: (b)              a  = b2a(b)            -> o1 = a                       (mbfir_b2a)
 *     op 1: (a, b)           rf = ab2rf(a, b)       -> o1 = rf                      (mbfir_ab2rf)
 *     op 2: (b)              rf = b2rf(b)           -> o1 = rf                      (mbfir_b2rf)
 *     op 3: (rf, g, x, mode) [a b] = abrm(rf, g, x) -> o1 = a, o2 = b; g may be []  (mbfir_abr)
 *
 * It is the reference's own gateways (rf_tools/mex5/b2a.c:31-68, cabc2rf.c, abrx.c:35-62) with the compute call
 * swapped for the C ABI: same plain double planes, no static scratch, no MAXN cap.  Shares nothing with
 * mbfir_mex.c but the context idiom.  Build:
 *   mex -R2017b matlab/mbfir_slr_mex.c -Iinclude -Lmultiband-rf-pulse-design_amd -lmbfir
 */
#include <string.h>
#include "mex.h"
#include "mbfir.h"

static mbfir_ctx* g_ctx = NULL;

static void release_ctx(void) {
    if (g_ctx) { mbfir_destroy(g_ctx); g_ctx = NULL; }
}

static size_t veclen(const mxArray* v) {
    size_t m = mxGetM(v), n = mxGetN(v);
    return m > n ? m : n;
}

static void planes(const mxArray* v, size_t len, double* re, double* im) {
    size_t i;
#if MX_HAS_INTERLEAVED_COMPLEX
    if (mxIsComplex(v)) {
        const mxComplexDouble* z = mxGetComplexDoubles(v);
        for (i = 0; i < len; ++i) { re[i] = z[i].real; im[i] = z[i].imag; }
    } else {
        const double* r = mxGetDoubles(v);
        for (i = 0; i < len; ++i) { re[i] = r[i]; im[i] = 0.0; }
    }
#else
    const double* r = mxGetPr(v);
    const double* q = mxGetPi(v);
    for (i = 0; i < len; ++i) { re[i] = r[i]; im[i] = q ? q[i] : 0.0; }
#endif
}

static double* out_plane(mxArray** slot, size_t len) {
    *slot = mxCreateDoubleMatrix(1, len, mxREAL);
#if MX_HAS_INTERLEAVED_COMPLEX
    return mxGetDoubles(*slot);
#else
    return mxGetPr(*slot);
#endif
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    int op, rc = MBFIR_E_ARG;
    size_t n, nx = 0;
    double *i1r, *i1i, *i2r = NULL, *i2i = NULL, *g = NULL, *x = NULL, *tmp = NULL;
    double *o1r, *o1i, *o2r = NULL, *o2i = NULL;

    if (nrhs < 2 || nlhs > 4) mexErrMsgTxt("Usage: [o1_re,o1_im,o2_re,o2_im] = mbfir_slr_mex(op, ...)");
    op = (int)mxGetScalar(prhs[0]);
    n = veclen(prhs[1]);
    if (n < 1) mexErrMsgTxt("empty input");
    if (!g_ctx) {
        g_ctx = mbfir_create(0);
        if (!g_ctx) mexErrMsgTxt(mbfir_last_error(NULL));
        mexAtExit(release_ctx);
    }
    i1r = (double*)mxCalloc(n, sizeof(double)); i1i = (double*)mxCalloc(n, sizeof(double));
    planes(prhs[1], n, i1r, i1i);
    if (op == 1) {                                       /* ab2rf(a, b) */
        if (nrhs < 3 || veclen(prhs[2]) != n) mexErrMsgTxt("ab2rf: a and b must have the same length");
        i2r = (double*)mxCalloc(n, sizeof(double)); i2i = (double*)mxCalloc(n, sizeof(double));
        planes(prhs[2], n, i2r, i2i);
    }
    if (op == 3) {                                       /* abrm(rf, g, x, mode) */
        if (nrhs < 4) mexErrMsgTxt("abrm: rf, g, x expected");
        nx = veclen(prhs[3]);
        if (nx < 1) mexErrMsgTxt("abrm: empty x");
        x = (double*)mxCalloc(nx, sizeof(double)); tmp = (double*)mxCalloc(nx > n ? nx : n, sizeof(double));
        planes(prhs[3], nx, x, tmp);
        if (!mxIsEmpty(prhs[2])) {
            if (veclen(prhs[2]) != n) mexErrMsgTxt("abrm: g must have one entry per rf sample");
            g = (double*)mxCalloc(n, sizeof(double));
            planes(prhs[2], n, g, tmp);
        }
        o1r = out_plane(&plhs[0], nx); o1i = out_plane(&plhs[1], nx);
        o2r = out_plane(&plhs[2], nx); o2i = out_plane(&plhs[3], nx);
        rc = mbfir_abr(g_ctx, (int)n, i1r, i1i, g, (int)nx, x, nrhs > 4 ? (int)mxGetScalar(prhs[4]) : 0, o1r, o1i, o2r, o2i);
    } else {
        o1r = out_plane(&plhs[0], n); o1i = out_plane(&plhs[1], n);
        if (op == 0) rc = mbfir_b2a(g_ctx, (int)n, i1r, i1i, o1r, o1i);
        else if (op == 1) rc = mbfir_ab2rf(g_ctx, (int)n, i1r, i1i, i2r, i2i, o1r, o1i);
        else if (op == 2) rc = mbfir_b2rf(g_ctx, (int)n, i1r, i1i, o1r, o1i);
        else mexErrMsgTxt("unknown op");
    }
    if (rc != 0) mexErrMsgTxt(mbfir_last_error(g_ctx));
    mxFree(i1r); mxFree(i1i);
    if (i2r) { mxFree(i2r); mxFree(i2i); }
    if (x) { mxFree(x); mxFree(tmp); }
    if (g) mxFree(g);
}
