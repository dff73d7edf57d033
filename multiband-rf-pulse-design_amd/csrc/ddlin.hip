// Double-double dense kernels of the extended-precision KKT solve (see ddkkt in solver.hip):
//   k_dd_syrk        H (dd) = H_w (double) + sum_r X_r u_r u_r'      -- the strongly weighted rows
//   k_ddchol_*       blocked right-looking Cholesky of H in dd (32-wide panels), L and L' kept
//   k_dd_trsv        (L L')^-1 b for 1 or 2 right-hand sides in dd
// The strongly active second-order cones of fir_qp_cvx (fir_qp_cvx.m:145-166) carry NT weights up to 1e16
// times the median weight; in double precision the weakly weighted directions of G' W^-2 G drown in the
// rounding of the strong ones.  With the strong rank-one terms accumulated, and the
// sum factorised, in dd they survive.  Everything here is fp64 VALU work with error-free transformations:
// there is no matrix-core path for dd.
#include "dev_common.h"
#include <cstdlib>
#include <mutex>
#include "dd_dev.h"
#include "solver.h"

#pragma clang fp contract(off)

namespace mbfir {

constexpr int DNB = 32;          // panel width of the dd Cholesky
constexpr int DT = 64;           // output tile of the dd rank-k kernels

// ------------------------------------------------------------------------------------------------
// H(dd, lower-triangle tiles) = Hh (as given, double) + sum_{r < *kcount} X[r] U[r][i] U[r][j]
// Tiles of 16 PT x 16 PT, a PT x PT patch per thread.  PT = 4 (64 x 64) gives 153 workgroups at np = 1088 -- fewer than the
// chip has CUs, each walking all k <= 588 rows with one wave per SIMD: 0.7 ms; PT = 2 (32 x 32) gives 561 of a quarter
// the length.  An entry's sum runs over r in the same order either way: the results are the same bit for bit.
template <int PT>
__global__ __launch_bounds__(256) void k_dd_syrk(const double* __restrict__ U, int ldu, const double* __restrict__ X,
                                                 const int* __restrict__ kcount, int np, double* __restrict__ Hh,
                                                 double* __restrict__ Hl) {
    constexpr int TS = 16 * PT;
    __shared__ double ui[8][TS], uj[8][TS], xs[8];
    // linear block index -> (bi >= bj)
    int bi = int((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((long)(bi + 1) * (bi + 2) / 2 <= (long)blockIdx.x) ++bi;
    while ((long)bi * (bi + 1) / 2 > (long)blockIdx.x) --bi;
    const int bj = blockIdx.x - bi * (bi + 1) / 2;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = bi * TS + ty * PT, j0 = bj * TS + tx * PT;
    dd acc[PT][PT];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int s = 0; s < PT; ++s) acc[p][s] = dd_make(Hh[(long)(i0 + p) * np + j0 + s], 0.0);
    const int k = *kcount;
    for (int r0 = 0; r0 < k; r0 += 8) {
        __syncthreads();
        for (int e = threadIdx.x; e < 8 * TS; e += 256) {
            const int q = e / TS, c = e - q * TS, r = r0 + q;
            ui[q][c] = r < k ? U[(long)r * ldu + bi * TS + c] : 0.0;
            uj[q][c] = r < k ? U[(long)r * ldu + bj * TS + c] : 0.0;
        }
        if (threadIdx.x < 8) xs[threadIdx.x] = r0 + threadIdx.x < k ? X[r0 + threadIdx.x] : 0.0;
        __syncthreads();
#pragma unroll 2
        for (int q = 0; q < 8; ++q) {
            const double x = xs[q];
            dd ax[PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) ax[p] = two_prod(ui[q][ty * PT + p], x);
#pragma unroll
            for (int s = 0; s < PT; ++s) {
                const double b = uj[q][tx * PT + s];
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[p][s] = dd_add(acc[p][s], dd_mul_d(ax[p], b));
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int s = 0; s < PT; ++s) {
            Hh[(long)(i0 + p) * np + j0 + s] = acc[p][s].h;
            Hl[(long)(i0 + p) * np + j0 + s] = acc[p][s].l;
        }
}

__global__ void k_dd_diag_copy(const double* __restrict__ Hh, int np, double* __restrict__ d0) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) d0[i] = Hh[(long)i * np + i];
}

// ------------------------------------------------------------------------------------------------
// Cholesky, panel step k0: (1) factorise the 32 x 32 diagonal block in one workgroup
// pivot rule of oracle/ddlin.c dd_chol: a pivot not above pivtol * d0 is replaced by d0 (counted in flag[0])
__global__ __launch_bounds__(1024) void k_ddchol_diag(double* __restrict__ Hh, double* __restrict__ Hl,
                                                     double* __restrict__ Lth, double* __restrict__ Ltl, int np, int k0,
                                                     const double* __restrict__ d0, double pivtol, int* __restrict__ flag,
                                                     double* __restrict__ rih, double* __restrict__ ril) {
    // ONE barrier per pivot: D keeps the running Schur complement UNSCALED (column j is not touched again after pivot j),
    // every thread forms 1 / sqrt(pivot) itself and scales the two column entries its updates need on the fly -- the same
    // products from the same operands in every thread, so the factor is what the three-barrier form (scale the column in
    // place, barrier, update) produced, bit for bit -- and the scaled entries go to a second array nobody reads in the loop.
    __shared__ double Dh[DNB][DNB + 1], Dl[DNB][DNB + 1];
    __shared__ double Fh[DNB][DNB + 1], Fl[DNB][DNB + 1];
    __shared__ double rh[DNB], rl[DNB], d0s[DNB];
    // one thread per entry (1024 = 32 x 32; the strict upper triangle idles): with four entries per thread their four
    // chains of dd operations ran one after the other -- the stores to D could alias the next entry's loads -- 1.1 us a pivot
    const int tid = threadIdx.x, i = tid / DNB, c = tid - i * DNB;
    {
        const bool lo = c <= i;
        Dh[i][c] = lo ? Hh[(long)(k0 + i) * np + k0 + c] : 0.0;
        Dl[i][c] = lo ? Hl[(long)(k0 + i) * np + k0 + c] : 0.0;
    }
    if (tid < DNB) d0s[tid] = d0[k0 + tid];
    for (int j = 0; j < DNB; ++j) {
        __syncthreads();                                      // the updates of pivot j - 1 are in place
        // x = 1 / sqrt(p.h) to double accuracy (the hardware's reciprocal square root, ~1e-9, and two Newton steps), then
        // one Newton step towards dd,  ri = x + x (1 - p x^2) / 2  (error 3/8 e^2, e ~ 1e-16), and r = p ri.  The residual
        // 1 - p x^2 is ~1e-16 and only its leading digits count: x^2 exactly as a two-term product, then three fused
        // multiply-adds (the first, 1 - p.h t.h, is exact up to one rounding of a number that small) -- 9 dependent
        // operations where the generic dd routines (and a correctly rounded 1 / sqrt) took 60: 15 of the block's 37 us
        // (tools/exp/ddiag_exp.hip; the factor moves by 1e-31).
        dd p = dd_make(Dh[j][j], Dl[j][j]);
        const double dj = d0s[j];
        const bool bad = !(p.h > pivtol * dj);
        if (bad) p = dd_make(dj > 1e-300 ? dj : 1e-300, 0.0);
        double x = __builtin_amdgcn_rsq(p.h);
        x = x * (1.5 - 0.5 * p.h * x * x);
        x = x * (1.5 - 0.5 * p.h * x * x);
        const dd t2 = two_prod(x, x);
        double e1 = __builtin_fma(-p.h, t2.h, 1.0);
        e1 = __builtin_fma(-p.h, t2.l, e1);
        e1 = __builtin_fma(-p.l, t2.h, e1);
        const dd ri = quick_two_sum(x, 0.5 * x * e1);
        if (i == j && c == j) {
            const dd r = dd_mul(p, ri);
            Fh[j][j] = r.h; Fl[j][j] = r.l;
            rh[j] = ri.h; rl[j] = ri.l;
            if (bad) atomicAdd(flag, 1);
        } else if (c == j && i > j) {
            const dd v = dd_mul(dd_make(Dh[i][j], Dl[i][j]), ri);
            Fh[i][j] = v.h; Fl[i][j] = v.l;
        } else if (c > j && i >= c) {
            const dd li = dd_mul(dd_make(Dh[i][j], Dl[i][j]), ri), lc = dd_mul(dd_make(Dh[c][j], Dl[c][j]), ri);
            const dd v = dd_fnma(dd_make(Dh[i][c], Dl[i][c]), li, lc);
            Dh[i][c] = v.h; Dl[i][c] = v.l;
        }
    }
    __syncthreads();
    if (c <= i) {
        Hh[(long)(k0 + i) * np + k0 + c] = Fh[i][c]; Hl[(long)(k0 + i) * np + k0 + c] = Fl[i][c];
        Lth[(long)(k0 + c) * np + k0 + i] = Fh[i][c]; Ltl[(long)(k0 + c) * np + k0 + i] = Fl[i][c];
    }
    if (tid < DNB) { rih[k0 + tid] = rh[tid]; ril[k0 + tid] = rl[tid]; }
}

// (2) panel rows below the diagonal block: X D' = A by forward substitution.  Eight threads per row; thread q OWNS the
// columns c = q (mod 8) of its row and keeps their running values  a_c - sum_{k < c, k solved} x_k D[c][k]  in registers.
// Step c: the owner of column c finishes it (one dd product with 1 / D[c][c]) and shuffles it to the row's other seven
// threads, and every thread takes x_c out of the columns it still owns.  On the chain from x_c to x_c+1 are one shuffle,
// one dd multiply-add and one dd product -- the other (up to three) multiply-adds of a thread wait for nobody.  (Round 2
// summed the products per thread and folded eight partial sums by three shuffles for EVERY column: 29 us a panel; one thread
// per row walked 496 dependent products: 55 us.)  The sums run over k ascending, as in the oracle's dd_chol.
__global__ __launch_bounds__(256) void k_ddchol_trsm(double* __restrict__ Hh, double* __restrict__ Hl,
                                                     double* __restrict__ Lth, double* __restrict__ Ltl, int np, int k0,
                                                     const double* __restrict__ rih, const double* __restrict__ ril) {
    __shared__ double Dh[DNB][DNB + 1], Dl[DNB][DNB + 1];
    __shared__ double rh[DNB], rl[DNB];
    const int tid = threadIdx.x, rr = tid >> 3, q8 = tid & 7, lane = tid & 63;
    const int i0 = k0 + DNB + blockIdx.x * 32, i = i0 + rr;
    for (int e = tid; e < DNB * DNB; e += 256) {
        const int r = e / DNB, c = e - r * DNB;
        Dh[r][c] = Hh[(long)(k0 + r) * np + k0 + c];
        Dl[r][c] = Hl[(long)(k0 + r) * np + k0 + c];
    }
    if (tid < DNB) { rh[tid] = rih[k0 + tid]; rl[tid] = ril[k0 + tid]; }
    dd x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 8 * k + q8;
        x[k] = i < np ? dd_make(Hh[(long)i * np + k0 + c], Hl[(long)i * np + k0 + c]) : dd_make(0.0, 0.0);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < DNB; ++c) {
        const int kc = c >> 3, owner = c & 7;                 // (static after unrolling)
        if (q8 == owner) x[kc] = dd_mul(x[kc], dd_make(rh[c], rl[c]));
        const dd xc = dd_shfl(x[kc], (lane & ~7) | owner);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cc = 8 * k + q8;                        // a column this thread owns
            if (8 * k + 7 > c) {                              // (static: some column of block k lies behind c)
                const dd v = dd_fnma(x[k], xc, dd_make(Dh[cc][c], Dl[cc][c]));
                if (cc > c) x[k] = v;
            }
        }
    }
    if (i < np) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 8 * k + q8;
            Hh[(long)i * np + k0 + c] = x[k].h; Hl[(long)i * np + k0 + c] = x[k].l;
            Lth[(long)(k0 + c) * np + i] = x[k].h; Ltl[(long)(k0 + c) * np + i] = x[k].l;
        }
    }
}

// (3) trailing update A[i][j] -= sum_c L[i][k0+c] L[j][k0+c] for i >= j >= k0 + 32, lower-triangle tiles of 16 PT x 16 PT
// on the global grid of that size (entries of a tile above k1 = k0 + 32 are skipped), a PT x PT patch per thread.
// With 64 x 64 tiles (PT = 4) the trailing matrix of np = 1088 is 153 tiles at the first step and 51 on average -- a
// fifth of the chip, 512 dependent dd multiply-adds per thread, 48 us a step on the chain of the factorisation; 32 x 32
// tiles (PT = 2) are four times as many of a quarter the length.  Same sums in the same order: bit-identical.
template <int PT>
__global__ __launch_bounds__(256) void k_ddchol_update(double* __restrict__ Hh, double* __restrict__ Hl, int np, int k0) {
    constexpr int TS = 16 * PT, LD = TS + 1;                  // panel staged as [column][row], conflict-free both ways
    extern __shared__ double upd_sm[];
    double* Lih = upd_sm;
    double* Lil = Lih + DNB * LD;
    double* Ljh = Lil + DNB * LD;
    double* Ljl = Ljh + DNB * LD;
    const int k1 = k0 + DNB, b0 = k1 / TS;
    int bi = int((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((long)(bi + 1) * (bi + 2) / 2 <= (long)blockIdx.x) ++bi;
    while ((long)bi * (bi + 1) / 2 > (long)blockIdx.x) --bi;
    const int bj = blockIdx.x - bi * (bi + 1) / 2;
    const int ti = (b0 + bi) * TS, tj = (b0 + bj) * TS;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    for (int e = threadIdx.x; e < TS * DNB; e += 256) {
        const int r = e / DNB, c = e - r * DNB;
        const bool vi = ti + r >= k1, vj = tj + r >= k1;
        Lih[c * LD + r] = vi ? Hh[(long)(ti + r) * np + k0 + c] : 0.0;
        Lil[c * LD + r] = vi ? Hl[(long)(ti + r) * np + k0 + c] : 0.0;
        Ljh[c * LD + r] = vj ? Hh[(long)(tj + r) * np + k0 + c] : 0.0;
        Ljl[c * LD + r] = vj ? Hl[(long)(tj + r) * np + k0 + c] : 0.0;
    }
    __syncthreads();
    const int i0 = ti + ty * PT, j0 = tj + tx * PT;
    if (i0 + PT - 1 < j0) return;                             // PT x PT patch entirely above the diagonal
    dd acc[PT][PT];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int s = 0; s < PT; ++s) acc[p][s] = dd_make(Hh[(long)(i0 + p) * np + j0 + s], Hl[(long)(i0 + p) * np + j0 + s]);
    for (int c = 0; c < DNB; ++c) {
        dd a[PT], b[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) a[p] = dd_make(Lih[c * LD + ty * PT + p], Lil[c * LD + ty * PT + p]);
#pragma unroll
        for (int s = 0; s < PT; ++s) b[s] = dd_make(Ljh[c * LD + tx * PT + s], Ljl[c * LD + tx * PT + s]);
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int s = 0; s < PT; ++s) acc[p][s] = dd_fnma(acc[p][s], a[p], b[s]);
    }
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int s = 0; s < PT; ++s)
            if (i0 + p >= k1 && j0 + s >= k1 && i0 + p >= j0 + s) {
                Hh[(long)(i0 + p) * np + j0 + s] = acc[p][s].h;
                Hl[(long)(i0 + p) * np + j0 + s] = acc[p][s].l;
            }
}
template <int PT>
constexpr size_t upd_lds() { return 4 * (size_t)DNB * (16 * PT + 1) * sizeof(double); }

// tile edge of the dd rank-k kernels: 32 while that still gives at most a few thousand workgroups, 64 beyond (MBFIR_DD_TILE=64|32)
static int dd_patch(int np) {
    int pt = np <= 2048 ? 2 : 4;
    if (const char* ev = std::getenv("MBFIR_DD_TILE")) pt = std::atoi(ev) == 64 ? 4 : (std::atoi(ev) == 32 ? 2 : pt);
    return pt;
}

void dd_syrk_launch(const double* U, int ldu, const double* X, const int* kcount, int np, double* Hh, double* Hl,
                    hipStream_t st) {
    if (dd_patch(np) == 2) {
        const int nt = np / 32;
        hipLaunchKernelGGL(k_dd_syrk<2>, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, U, ldu, X, kcount, np, Hh, Hl);
    } else {
        const int nt = np / 64;
        hipLaunchKernelGGL(k_dd_syrk<4>, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, U, ldu, X, kcount, np, Hh, Hl);
    }
}

// Inverses of the 64 x 64 diagonal blocks of L in double-double (one workgroup per block; (Xh, Xl): np/64 blocks of 64 x 64,
// row-major, the strict upper triangles zero).  The triangular solves multiply by them instead of substituting: a
// substitution is 64 dependent steps per block ON the chain of the solve, a product is one (k_dd_trsv_bi).  With the
// conditioning of a diagonal block of L (<= sqrt(cond H) ~ 1e10) and a unit round-off of 1e-32 the product is as good
// as the substitution to ~1e-22.  Column j of the inverse is a forward substitution of e_j: four adjacent lanes share a
// column (the sum over k split by k mod 4, folded with two DPP-free shuffles), a thread only ever reads back the
// entries it stored itself (k mod 4 == its part), so the 64 steps need no barrier.
constexpr int DIB = 64;
constexpr size_t BLOCKINV_LDS = 4 * DIB * DIB * sizeof(double);
__global__ __launch_bounds__(256) void k_dd_blockinv(const double* __restrict__ Lh, const double* __restrict__ Ll,
                                                     const double* __restrict__ rih, const double* __restrict__ ril, int np,
                                                     double* __restrict__ Xh, double* __restrict__ Xl) {
    extern __shared__ double smem[];                                // 4 x 64 x 64 doubles (BLOCKINV_LDS)
    double (*lh)[DIB] = reinterpret_cast<double (*)[DIB]>(smem);                       // the block of L
    double (*ll)[DIB] = reinterpret_cast<double (*)[DIB]>(smem + DIB * DIB);
    double (*xh)[DIB] = reinterpret_cast<double (*)[DIB]>(smem + 2 * DIB * DIB);       // xh[k][j] = (L_bb^-1)_kj, written by the thread (j, k mod 4)
    double (*xl)[DIB] = reinterpret_cast<double (*)[DIB]>(smem + 3 * DIB * DIB);
    const int b0 = blockIdx.x * DIB, tid = threadIdx.x;
    for (int e = tid; e < DIB * DIB; e += 256) {
        const int r = e / DIB, c = e - r * DIB;
        lh[r][c] = c <= r ? Lh[(long)(b0 + r) * np + b0 + c] : 0.0;
        ll[r][c] = c <= r ? Ll[(long)(b0 + r) * np + b0 + c] : 0.0;
    }
    __syncthreads();
    const int j = tid >> 2, part = tid & 3;
    double* oh = Xh + (size_t)blockIdx.x * DIB * DIB;
    double* ol = Xl + (size_t)blockIdx.x * DIB * DIB;
    for (int i = 0; i < DIB; ++i) {
        dd x = dd_make(0.0, 0.0);
        if (i == j) x = dd_make(rih[b0 + i], ril[b0 + i]);
        else if (i > j) {
            dd acc = dd_make(0.0, 0.0);
            for (int k = j + ((part - j) & 3); k < i; k += 4) acc = dd_fnma(acc, dd_make(lh[i][k], ll[i][k]), dd_make(xh[k][j], xl[k][j]));
            acc = dd_add(acc, dd_make(__shfl_xor(acc.h, 1, 64), __shfl_xor(acc.l, 1, 64)));
            acc = dd_add(acc, dd_make(__shfl_xor(acc.h, 2, 64), __shfl_xor(acc.l, 2, 64)));
            x = dd_mul(acc, dd_make(rih[b0 + i], ril[b0 + i]));
        }
        if ((i & 3) == part) {
            xh[i][j] = x.h; xl[i][j] = x.l;                         // (read back by this thread only)
            oh[i * DIB + j] = x.h; ol[i * DIB + j] = x.l;
        }
    }
}

// The launch sites of the dd Cholesky, one non-inlined function each: a stack trace through hipLaunchKernel then names the
// site (round 3's crash log showed one frame, dd_chol_launch, for all of them).
#define DD_SITE __attribute__((noinline)) static
DD_SITE void dd_site_diag(double* Hh, double* Hl, double* Lth, double* Ltl, int np, int k0, double* d0, double pivtol, int* flag,
                          double* rih, double* ril, hipStream_t st) {
    hipLaunchKernelGGL(k_ddchol_diag, dim3(1), dim3(1024), 0, st, Hh, Hl, Lth, Ltl, np, k0, d0, pivtol, flag, rih, ril);
}
DD_SITE void dd_site_trsm(double* Hh, double* Hl, double* Lth, double* Ltl, int np, int k0, int rows, double* rih, double* ril, hipStream_t st) {
    hipLaunchKernelGGL(k_ddchol_trsm, dim3(cdiv(rows, 32)), dim3(256), 0, st, Hh, Hl, Lth, Ltl, np, k0, rih, ril);
}
DD_SITE void dd_site_update(double* Hh, double* Hl, int np, int k0, int pt, hipStream_t st) {
    if (pt == 2) {
        const int nt = (np - k0 - DNB) / 32;
        hipLaunchKernelGGL(k_ddchol_update<2>, dim3(nt * (nt + 1) / 2), dim3(256), upd_lds<2>(), st, Hh, Hl, np, k0);
    } else {
        const int b0 = (k0 + DNB) / DT, nt = np / DT - b0;
        hipLaunchKernelGGL(k_ddchol_update<4>, dim3(nt * (nt + 1) / 2), dim3(256), upd_lds<4>(), st, Hh, Hl, np, k0);
    }
}
DD_SITE void dd_site_blockinv(double* Hh, double* Hl, double* rih, double* ril, int np, double* dinv, hipStream_t st) {
    hipLaunchKernelGGL(k_dd_blockinv, dim3(np / DIB), dim3(256), BLOCKINV_LDS, st, Hh, Hl, rih, ril, np, dinv, dinv + (size_t)np * DIB);
}
#undef DD_SITE

// In-place lower Cholesky of the dd matrix (Hh, Hl) (np x np row-major, np a multiple of 64); on exit the lower
// triangle holds L, (Lth, Ltl) hold L' (upper triangle, row-major), (rih, ril) the reciprocals of diag(L);
// flag[0] counts replaced pivots; d0 is a work vector of np doubles.
void dd_chol_launch(double* Hh, double* Hl, double* Lth, double* Ltl, double* rih, double* ril, double* d0, int np,
                    double pivtol, int* flag, hipStream_t st, double* dinv) {
    const int pt = dd_patch(np);
    hipMemsetAsync(flag, 0, sizeof(int), st);
    hipLaunchKernelGGL(k_dd_diag_copy, dim3(cdiv(np, 256)), dim3(256), 0, st, Hh, np, d0);
    for (int k0 = 0; k0 < np; k0 += DNB) {
        dd_site_diag(Hh, Hl, Lth, Ltl, np, k0, d0, pivtol, flag, rih, ril, st);
        const int rows = np - k0 - DNB;
        if (rows <= 0) break;
        dd_site_trsm(Hh, Hl, Lth, Ltl, np, k0, rows, rih, ril, st);
        dd_site_update(Hh, Hl, np, k0, pt, st);
    }
    if (dinv) dd_site_blockinv(Hh, Hl, rih, ril, np, dinv, st);
    // the update and block-inverse kernels need more LDS than a launch gets by default: a device whose attribute was not set
    // rejects them, and the factor would stay whatever the buffers held
    MBFIR_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// B (NV right-hand sides, dd, stride ldv) <- (L L')^-1 B.  One workgroup; the right-hand sides live in LDS.
// Forward sweep over 32-row blocks with L (rows), backward sweep with L' (rows of Lt): the diagonal block is
// solved by one wave per right-hand side (lane = row, the solved unknown broadcast by shuffles), the rest of
// the block column is applied by all threads.
constexpr int DD_NP_MAX = 4608;
template <int NV>
__global__ __launch_bounds__(1024) void k_dd_trsv(const double* __restrict__ Lh, const double* __restrict__ Ll,
                                                  const double* __restrict__ Lth, const double* __restrict__ Ltl,
                                                  const double* __restrict__ rih, const double* __restrict__ ril, int np,
                                                  double* __restrict__ Bh, double* __restrict__ Bl, int ldv) {
    extern __shared__ double smem[];
    double* yh = smem;                        // NV * np
    double* yl = yh + (size_t)NV * np;        // NV * np
    double* Dh = yl + (size_t)NV * np;        // 32 * 33
    double* Dl = Dh + DNB * (DNB + 1);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < NV * np; e += 1024) {
        const int v = e / np, i = e - v * np;
        yh[e] = Bh[(long)v * ldv + i]; yl[e] = Bl[(long)v * ldv + i];
    }
    for (int pass = 0; pass < 2; ++pass) {
        const double* Th = pass == 0 ? Lh : Lth;
        const double* Tl = pass == 0 ? Ll : Ltl;
        for (int bb = 0; bb < np / DNB; ++bb) {
            const int b0 = pass == 0 ? bb * DNB : np - DNB - bb * DNB;
            __syncthreads();
            for (int e = tid; e < DNB * DNB; e += 1024) {
                const int r = e / DNB, c = e - r * DNB;
                Dh[r * (DNB + 1) + c] = Th[(long)(b0 + r) * np + b0 + c];
                Dl[r * (DNB + 1) + c] = Tl[(long)(b0 + r) * np + b0 + c];
            }
            __syncthreads();
            if (wv < NV && lane < DNB) {
                dd y = dd_make(yh[wv * np + b0 + lane], yl[wv * np + b0 + lane]);
                for (int s = 0; s < DNB; ++s) {
                    const int q = pass == 0 ? s : DNB - 1 - s;           // forward: ascending, backward: descending
                    const dd t = dd_mul(dd_shfl(y, q), dd_make(rih[b0 + q], ril[b0 + q]));
                    if (lane == q) y = t;
                    const bool pending = pass == 0 ? lane > q : lane < q;
                    if (pending) y = dd_fnma(y, dd_make(Dh[lane * (DNB + 1) + q], Dl[lane * (DNB + 1) + q]), t);
                }
                yh[wv * np + b0 + lane] = y.h; yl[wv * np + b0 + lane] = y.l;
            }
            __syncthreads();
            // rest of the block column: forward rows i >= b0 + 32 (row i of L, columns b0..b0+31),
            // backward rows i < b0 (row i of L', columns b0..b0+31)
            const int lo = pass == 0 ? b0 + DNB : 0, hi = pass == 0 ? np : b0;
            for (int i = lo + tid; i < hi; i += 1024) {
                dd acc[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] = dd_make(yh[v * np + i], yl[v * np + i]);
                const double* th = Th + (long)i * np + b0;
                const double* tl = Tl + (long)i * np + b0;
                for (int c = 0; c < DNB; ++c) {
                    const dd lv = dd_make(th[c], tl[c]);
#pragma unroll
                    for (int v = 0; v < NV; ++v) acc[v] = dd_fnma(acc[v], lv, dd_make(yh[v * np + b0 + c], yl[v * np + b0 + c]));
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) { yh[v * np + i] = acc[v].h; yl[v * np + i] = acc[v].l; }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < NV * np; e += 1024) {
        const int v = e / np, i = e - v * np;
        Bh[(long)v * ldv + i] = yh[e]; Bl[(long)v * ldv + i] = yl[e];
    }
}

// The same solve spread over one workgroup per 32-row block and pass (2 np / 32 workgroups in ONE launch): block b
// accumulates  - L_bc y_c  as the blocks c it depends on are published (a flag per block in global memory, value =
// `epoch` of this call, polled with a bound), solves its diagonal block and publishes y_b.
// Forward progress does not rest on the order in which the hardware dispatches workgroups: a workgroup takes its block
// from a TICKET counter (flags[2 nb], never reset: launch number e of one solve hands out (e-1) grid .. e grid - 1) in
// dependency order -- forward blocks first, each block only waits for lower tickets, and a ticket is drawn by a
// workgroup that is already running -- so nobody waits for a workgroup that has not started.
// Hand-off (cdna_hip_programming.md guideline 16, R1): the right-hand sides are overwritten in place with sc1
// (write-through) stores, every storing wave drains them (s_waitcnt vmcnt(0)) before the barrier behind which ONE lane
// stores the flag; the consumer polls with ONE lane, passes a barrier, and reads the published entries with sc1 loads
// (another CU, maybe another XCD, produced them); the factor is read normally (nobody writes it).  A poll that expires
// adds DD_SYNC_LOST to the pivot counter, which the host turns into an error.
// One workgroup walked the whole factor in 1.4 ms at np = 1088; the chain here is 34 x (diagonal solve + hand-over).
__device__ __forceinline__ double ld_dev(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wait_flag(const int* f, int epoch, int* lost) {
    int spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        if (++spins >= (1 << 22)) { atomicAdd(lost, DD_SYNC_LOST); return; }
        __builtin_amdgcn_s_sleep(2);
    }
}
template <int NV>
__global__ __launch_bounds__(256) void k_dd_trsv_mw(const double* __restrict__ Lh, const double* __restrict__ Ll,
                                                    const double* __restrict__ Lth, const double* __restrict__ Ltl,
                                                    const double* __restrict__ rih, const double* __restrict__ ril, int np,
                                                    double* Bh, double* Bl, int ldv, int* flags, int epoch, int* lost) {
    __shared__ double Dh[DNB * (DNB + 1)], Dl[DNB * (DNB + 1)];
    __shared__ double ysh[2][NV][DNB], yrow[2][NV][DNB];
    __shared__ int ticket;
    const int nb = np / DNB;
    if (threadIdx.x == 0) ticket = atomicAdd(flags + 2 * nb, 1) - (epoch - 1) * int(gridDim.x);
    __syncthreads();
    const int g = ticket, pass = g >= nb ? 1 : 0, bb = pass ? g - nb : g;
    if (g < 0 || g >= 2 * nb) { if (threadIdx.x == 0) atomicAdd(lost, DD_SYNC_LOST); return; }     // (a ticket word somebody else touched)
    const int b0 = pass == 0 ? bb * DNB : np - DNB - bb * DNB;
    const double* Th = pass == 0 ? Lh : Lth;
    const double* Tl = pass == 0 ? Ll : Ltl;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = tid >> 3, q8 = tid & 7;
    for (int e = tid; e < DNB * DNB; e += 256) {             // the diagonal block (read-only data)
        const int rr = e / DNB, c = e - rr * DNB;
        Dh[rr * (DNB + 1) + c] = Th[(long)(b0 + rr) * np + b0 + c];
        Dl[rr * (DNB + 1) + c] = Tl[(long)(b0 + rr) * np + b0 + c];
    }
    // the rows' own right-hand side first (off the chain): given (forward), or the forward result of the same rows
    // (backward; that flag is raised long before the backward blocks this one waits for below)
    if (pass == 1) {
        if (tid == 0) wait_flag(flags + b0 / DNB, epoch, lost);
        __syncthreads();
    }
    dd acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
        acc[v] = q8 == 0 ? dd_make(ld_dev(Bh + (long)v * ldv + b0 + r), ld_dev(Bl + (long)v * ldv + b0 + r)) : dd_make(0.0, 0.0);
    const double* th = Th + (long)(b0 + r) * np + q8 * 4;
    const double* tl = Tl + (long)(b0 + r) * np + q8 * 4;
    for (int c = 0; c < bb; ++c) {
        const int b0c = pass == 0 ? c * DNB : np - DNB - c * DNB;
        if (tid == 0) wait_flag(flags + pass * nb + c, epoch, lost);
        __syncthreads();
        if (tid < DNB * NV) {
            const int v = tid / DNB, i = tid - v * DNB;
            ysh[0][v][i] = ld_dev(Bh + (long)v * ldv + b0c + i);
            ysh[1][v][i] = ld_dev(Bl + (long)v * ldv + b0c + i);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const dd lv = dd_make(th[b0c + j], tl[b0c + j]);
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] = dd_fnma(acc[v], lv, dd_make(ysh[0][v][q8 * 4 + j], ysh[1][v][q8 * 4 + j]));
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) acc[v] = dd_add(acc[v], dd_make(__shfl_xor(acc[v].h, o, 64), __shfl_xor(acc[v].l, o, 64)));
        if (q8 == 0) { yrow[0][v][r] = acc[v].h; yrow[1][v][r] = acc[v].l; }
    }
    __syncthreads();
    if (wv < NV && lane < DNB) {                              // diagonal block: one wave per right-hand side, lane = row
        dd y = dd_make(yrow[0][wv][lane], yrow[1][wv][lane]);
        for (int s = 0; s < DNB; ++s) {
            const int q = pass == 0 ? s : DNB - 1 - s;
            const dd t = dd_mul(dd_shfl(y, q), dd_make(rih[b0 + q], ril[b0 + q]));
            if (lane == q) y = t;
            const bool pending = pass == 0 ? lane > q : lane < q;
            if (pending) y = dd_fnma(y, dd_make(Dh[lane * (DNB + 1) + q], Dl[lane * (DNB + 1) + q]), t);
        }
        st_dev(Bh + (long)wv * ldv + b0 + lane, y.h);
        st_dev(Bl + (long)wv * ldv + b0 + lane, y.l);
    }
    drain_stores();                                                           // every storing wave: its sc1 stores have left the CU ...
    __syncthreads();                                              // ... before the one lane that signals for all of them does
    if (tid == 0) __hip_atomic_store(flags + pass * nb + bb, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same solve on 64-row blocks with the inverses of the diagonal blocks (k_dd_blockinv): 2 np / 64 workgroups of 512
// threads (thread = row x eighth of a block's columns).  On the chain of block b there is, after the last block it waits
// for has been published: 8 multiply-adds per thread and right-hand side, a fold over the 8 threads of a row, the same
// again with the inverse block (held in registers since before the wait), the stores.  34 chain steps instead of 68 at
// np = 1088, none with a substitution in it.  Tickets, flags, hand-offs: as in k_dd_trsv_mw.
template <int NV>
__global__ __launch_bounds__(512) void k_dd_trsv_bi(const double* __restrict__ Lh, const double* __restrict__ Ll,
                                                    const double* __restrict__ Lth, const double* __restrict__ Ltl,
                                                    const double* __restrict__ Xh, const double* __restrict__ Xl, int np,
                                                    double* Bh, double* Bl, int ldv, int* flags, int ticket_at, int epoch, int* lost) {
    __shared__ double ysh[2][NV][DIB];
    __shared__ int ticket;
    const int nb = np / DIB;
    if (threadIdx.x == 0) ticket = atomicAdd(flags + ticket_at, 1) - (epoch - 1) * int(gridDim.x);
    __syncthreads();
    const int g = ticket, pass = g >= nb ? 1 : 0, bb = pass ? g - nb : g;
    if (g < 0 || g >= 2 * nb) { if (threadIdx.x == 0) atomicAdd(lost, DD_SYNC_LOST); return; }     // (a ticket word somebody else touched)
    const int blk = pass == 0 ? bb : nb - 1 - bb, b0 = blk * DIB;
    const double* Th = pass == 0 ? Lh : Lth;
    const double* Tl = pass == 0 ? Ll : Ltl;
    const int tid = threadIdx.x, r = tid >> 3, q8 = tid & 7;
    // this thread's eight entries of the inverse block (forward: row r of it, backward: row r of its transpose)
    dd xi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = q8 * 8 + j;
        const size_t o = (size_t)blk * DIB * DIB + (pass == 0 ? (size_t)r * DIB + c : (size_t)c * DIB + r);
        xi[j] = dd_make(Xh[o], Xl[o]);
    }
    if (pass == 1) {                                           // the rows' own right-hand side: the forward result of the same rows
        if (tid == 0) wait_flag(flags + blk, epoch, lost);
        __syncthreads();
    }
    dd acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
        acc[v] = q8 == 0 ? dd_make(ld_dev(Bh + (long)v * ldv + b0 + r), ld_dev(Bl + (long)v * ldv + b0 + r)) : dd_make(0.0, 0.0);
    const double* th = Th + (long)(b0 + r) * np + q8 * 8;
    const double* tl = Tl + (long)(b0 + r) * np + q8 * 8;
    for (int c = 0; c < bb; ++c) {
        const int b0c = (pass == 0 ? c : nb - 1 - c) * DIB;
        dd lv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) lv[j] = dd_make(th[b0c + j], tl[b0c + j]);           // (the factor: read-only, in flight across the wait)
        if (tid == 0) wait_flag(flags + pass * nb + c, epoch, lost);
        __syncthreads();
        if (tid < DIB * NV) {
            const int v = tid / DIB, i = tid - v * DIB;
            ysh[0][v][i] = ld_dev(Bh + (long)v * ldv + b0c + i);
            ysh[1][v][i] = ld_dev(Bl + (long)v * ldv + b0c + i);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] = dd_fnma(acc[v], lv[j], dd_make(ysh[0][v][q8 * 8 + j], ysh[1][v][q8 * 8 + j]));
    }
    __syncthreads();                                            // (ysh is reused below)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) acc[v] = dd_add(acc[v], dd_make(__shfl_xor(acc[v].h, o, 64), __shfl_xor(acc[v].l, o, 64)));
        if (q8 == 0) { ysh[0][v][r] = acc[v].h; ysh[1][v][r] = acc[v].l; }
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        dd y = dd_make(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < 8; ++j) y = dd_add(y, dd_mul(xi[j], dd_make(ysh[0][v][q8 * 8 + j], ysh[1][v][q8 * 8 + j])));
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) y = dd_add(y, dd_make(__shfl_xor(y.h, o, 64), __shfl_xor(y.l, o, 64)));
        if (q8 == 0) {
            st_dev(Bh + (long)v * ldv + b0 + r, y.h);
            st_dev(Bl + (long)v * ldv + b0 + r, y.l);
        }
    }
    drain_stores();                                                           // every storing wave: its sc1 stores have left the CU ...
    __syncthreads();                                              // ... before the one lane that signals for all of them does
    if (tid == 0) __hip_atomic_store(flags + pass * nb + bb, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// dinv: the inverses of the 64 x 64 diagonal blocks (dd_chol_launch; 2 np x 64 doubles) -> k_dd_trsv_bi; nullptr: substitution
// on 32-row blocks (k_dd_trsv_mw).
// flags: 2 np / 32 + 1 ints owned by the caller (zeroed once; the last one is the ticket counter), epoch: 1, 2, 3, ...
// over the calls that use these flags (every call with the same np); lost: the counter a lost hand-off is reported in
// (DD_SYNC_LOST, next to the replaced pivots); flags == nullptr: the single-workgroup kernel
void dd_trsv_launch(const double* Lh, const double* Ll, const double* Lth, const double* Ltl, const double* rih,
                    const double* ril, int np, double* Bh, double* Bl, int nv, int ldv, hipStream_t st, int* flags, int epoch, int* lost,
                    const double* dinv) {
    if (dinv && flags && lost && (nv == 1 || nv == 2) && np % DIB == 0) {
        // (one solve uses one form throughout: the ticket word counts the launches of ONE grid size)
        const dim3 grid(2 * np / DIB);
        const double* xl = dinv + (size_t)np * DIB;
        if (nv == 1) hipLaunchKernelGGL(k_dd_trsv_bi<1>, grid, dim3(512), 0, st, Lh, Ll, Lth, Ltl, dinv, xl, np, Bh, Bl, ldv, flags, 2 * np / DNB, epoch, lost);
        else hipLaunchKernelGGL(k_dd_trsv_bi<2>, grid, dim3(512), 0, st, Lh, Ll, Lth, Ltl, dinv, xl, np, Bh, Bl, ldv, flags, 2 * np / DNB, epoch, lost);
        return;
    }
    if (flags && lost && (nv == 1 || nv == 2) && np % DNB == 0) {
        const dim3 grid(2 * np / DNB);
        if (nv == 1) hipLaunchKernelGGL(k_dd_trsv_mw<1>, grid, dim3(256), 0, st, Lh, Ll, Lth, Ltl, rih, ril, np, Bh, Bl, ldv, flags, epoch, lost);
        else hipLaunchKernelGGL(k_dd_trsv_mw<2>, grid, dim3(256), 0, st, Lh, Ll, Lth, Ltl, rih, ril, np, Bh, Bl, ldv, flags, epoch, lost);
        return;
    }
    if (np > DD_NP_MAX) throw HipError("dd solve: matrix too large for the LDS-resident right-hand sides");
    const size_t sh = (2 * (size_t)nv * np + 2 * DNB * (DNB + 1)) * sizeof(double);
    if (nv == 1) hipLaunchKernelGGL(k_dd_trsv<1>, dim3(1), dim3(1024), sh, st, Lh, Ll, Lth, Ltl, rih, ril, np, Bh, Bl, ldv);
    else if (nv == 2) hipLaunchKernelGGL(k_dd_trsv<2>, dim3(1), dim3(1024), sh, st, Lh, Ll, Lth, Ltl, rih, ril, np, Bh, Bl, ldv);
    else throw HipError("dd solve: unsupported number of right-hand sides");
    MBFIR_HIP(hipGetLastError());                          // (up to 160 KB of LDS: rejected where the attribute is missing)
}

// Per-device preparation of this file's kernels (called from the Solver's constructor, once per device id under its
// mutex): the code objects are resolved by ONE thread before any context launches them, and the dynamic-LDS limits above
// 64 KB -- an attribute of the device function of the CURRENT device -- are set on every device that gets a context
// (round 3 set them behind process-wide flags: only the first context's device ever got them).
void dd_warm_kernels() {
    hipFuncAttributes fa;
    const void* ks[] = {reinterpret_cast<const void*>(&k_dd_syrk<2>), reinterpret_cast<const void*>(&k_dd_syrk<4>),
                        reinterpret_cast<const void*>(&k_dd_diag_copy), reinterpret_cast<const void*>(&k_ddchol_diag),
                        reinterpret_cast<const void*>(&k_ddchol_trsm), reinterpret_cast<const void*>(&k_ddchol_update<2>),
                        reinterpret_cast<const void*>(&k_ddchol_update<4>), reinterpret_cast<const void*>(&k_dd_blockinv)};
    for (const void* k : ks) (void)hipFuncGetAttributes(&fa, k);
    MBFIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ddchol_update<4>), hipFuncAttributeMaxDynamicSharedMemorySize, int(upd_lds<4>())));
    MBFIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ddchol_update<2>), hipFuncAttributeMaxDynamicSharedMemorySize, int(upd_lds<2>())));
    MBFIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dd_blockinv), hipFuncAttributeMaxDynamicSharedMemorySize, int(BLOCKINV_LDS)));
    MBFIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dd_trsv<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    MBFIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dd_trsv<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
}

}  // namespace mbfir
