// Capacitance (saddle-point) form of the extended-precision KKT solve, plain double precision on the fp64 matrix cores
// (round 4; oracle/conic_ipm.py factor_dd with ddkkt["form"] == "cap" is the same algorithm, step for step).
//
// The normal matrix of an iterate with nearly active cones is  H = H_w + U' X U : H_w from the CAPPED scaling (every eigenvalue
// of W^-2 at most cap = 1e6 x the typical weight), U one row G'e per strong eigen-direction (k rows), X = diag(excess weights,
// up to 1e16 x the typical one).  Round 2 / 3 accumulated and factorised H in double-double (ddlin.hip: fp64 VALU with
// error-free transformations, 34 panel steps x 3 launches, 2.7 ms per factorisation at np = 1088).  Here the strong directions
// stay what they are -- nearly-equality constraints --
//        [ H_w   U'   ] [ dx   ]   [ rhs_w ]
//        [ U   -X^-1  ] [ zeta ] = [   t   ]        zeta = X (U dx - t)
// and the system is solved by block elimination on H_w: no number of the size of X ever meets one of the size of H_w, so double
// precision is enough (measured in the oracle: BASELINE config 3's family, n = 384, 60 iterations -- the same count, the same
// objective to 13 digits and the taps to 4e-8 of the double-double form):
//        H_w = L L'  (the ordinary double-precision factorisation, chol.hip; M = L^-1)
//        Yt  = U M'              (k x np)        Zt = Yt M = U H_w^-1     (k x np)
//        S   = X^-1 + Yt Yt'     (k x k, factorised by the same chol.hip routine: k <= 1024 instead of a dd one of np)
//        y = H_w^-1 rhs_w ;  zeta = S^-1 (U y - t) ;  dx = y - Zt' zeta
// Kernels: one 64 x 64-tile fp64 MFMA product kernel in three modes (the two products with the triangular M touch only the tiles
// of its lower triangle and mask the diagonal tile -- the storage above the diagonal of M is never written), and three small
// vector kernels.
#include "dev_common.h"

namespace mbfir {

namespace {
// Lock-step units (round 5): blockIdx.z / gridDim.z carry (split, lane) or the lane; every matrix of lane b sits lane_bytes behind
// lane b - 1's; kcnt (lane's arena) holds the lane's own number of strong directions -- the launches carry the unit's largest,
// the padding rows behave as in a single solve padded to that size (zero rows of U, unit diagonal of S).
struct CapLanes { size_t lane_bytes; const int* mask; const int* kcnt; };
template <class T>
__device__ __forceinline__ T* cap_at(T* p, size_t off) { return p ? reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(p)) + off) : p; }
__device__ __forceinline__ int cap_k(const CapLanes& L, int lane, int k) {
    // (clamped to the size the launch carries -- `k` is the unit's largest count, what Yt / Zt / S are laid out for: a lane's count above it
    //  could index past them; the host's retry loop keeps every count <= CAP_KMAX, this keeps the kernel safe by itself -- ADVICE r5)
    return L.kcnt ? min(*cap_at(L.kcnt, (size_t)lane * L.lane_bytes), k) : k;
}
constexpr int TB = 64, TLD = 66;
constexpr int CAP_SPLIT = 4;      // parts of a tile's K range (see k_cap_gemm)

// acc[a][b] += A(64 x 64 tile in As) * op(B tile in Bs): each of the 4 waves owns a 32 x 32 quadrant (2 x 2 MFMA blocks);
// v_mfma_f64_16x16x4_f64 operand layout: a = A[i = lane % 16][k = lane / 16], b = B[k = lane / 16][j = lane % 16],
// c[q] = C[lane / 16 + 4 q][lane % 16]
template <bool TRANSB>
__device__ __forceinline__ void mma_tile(const double (*As)[TLD], const double (*Bs)[TLD], v4d acc[2][2]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wi = wv >> 1, wj = wv & 1;
#pragma unroll 4
    for (int k0 = 0; k0 < TB; k0 += 4) {
        const int k = k0 + (lane >> 4);
        double af[2], bf[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) af[a] = As[wi * 32 + a * 16 + (lane & 15)][k];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = wj * 32 + b * 16 + (lane & 15);
            bf[b] = TRANSB ? Bs[j][k] : Bs[k][j];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
}
// 64 x 64 tile: global -> registers (256 threads, eight 16-byte loads each, all in flight), registers -> LDS.  lower: keep the
// lower triangle only (col <= row) -- both products with the triangular M keep exactly those entries of its diagonal tile
struct TileRegs { double2 t[8]; };
__device__ __forceinline__ void fetch_tile(TileRegs& R, const double* __restrict__ src, long ld) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = threadIdx.x + 256 * u;
        R.t[u] = *reinterpret_cast<const double2*>(src + (long)(e >> 5) * ld + 2 * (e & 31));
    }
}
__device__ __forceinline__ void store_tile(double (*S)[TLD], const TileRegs& R, bool lower) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = threadIdx.x + 256 * u, r = e >> 5, c = 2 * (e & 31);
        double2 t = R.t[u];
        if (lower) { if (c > r) t.x = 0.0; if (c + 1 > r) t.y = 0.0; }
        *reinterpret_cast<double2*>(&S[r][c]) = t;
    }
}

// MODE 0:  C = A B',  B = M lower-triangular (np x np):  C[r][i] = sum_{j <= i} A[r][j] M[i][j]        (Yt = U M')
// MODE 1:  C = A B ,  B = M lower-triangular:            C[r][i] = sum_{j >= i} A[r][j] M[j][i]        (Zt = Yt M)
// MODE 2:  C = A A' + diag (lower tiles only):           C[r][q] = sum_j A[r][j] A[q][j]               (S = Yt Yt' + X^-1)
//          diag: 1 / X[r] for r < k, 1 for the padding rows r >= k (whose rows of A are zero)
// Split over K (round 5): blockIdx.z takes the z-th of gridDim.z contiguous parts of the tile's K range and writes its partial
// product to the slab C + z * slab_stride; k_cap_fold adds the parts in a fixed order (and the diagonal term of MODE 2).  One
// block per 64 x 64 tile left 170 blocks for 256 CUs, each a serial chain of up to 17 (load, two barriers, 64 MFMAs per wave)
// steps with nothing to overlap them: 0.087 of the fp64 matrix peak.
template <int MODE>
__global__ __launch_bounds__(256) void k_cap_gemm(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int np, int ldc,
                                                  long slab_stride, int nsplit, CapLanes L) {
    __shared__ __attribute__((aligned(16))) double As[TB][TLD];
    __shared__ __attribute__((aligned(16))) double Bs[TB][TLD];
    const int lane_id = blockIdx.z / nsplit;
    if (L.mask && !L.mask[lane_id]) return;
    if (lane_id) { const size_t off = (size_t)lane_id * L.lane_bytes; A = cap_at(A, off); B = cap_at(B, off); C = cap_at(C, off); }
    int rb, cb;
    if (MODE == 2) {                                          // lower tiles of the k x k result, one per block
        int t = blockIdx.x;
        rb = int((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((rb + 1) * (rb + 2) / 2 <= t) ++rb;
        while (rb * (rb + 1) / 2 > t) --rb;
        cb = t - rb * (rb + 1) / 2;
    } else {
        rb = blockIdx.y; cb = blockIdx.x;
    }
    const int nblk = np / TB;
    int j0 = MODE == 1 ? cb : 0, j1 = MODE == 0 ? cb + 1 : nblk;
    {
        const int span = j1 - j0, z = blockIdx.z % nsplit, nz = nsplit;
        const int lo = j0 + int((long)span * z / nz), hi = j0 + int((long)span * (z + 1) / nz);
        j0 = lo; j1 = hi;
        C += (long)z * slab_stride;
    }
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
    // the next tile pair is fetched into registers while the matrix cores work on the current one (one LDS buffer, two barriers a step)
    auto a_src = [&](int jb) { return A + (long)rb * TB * np + (long)jb * TB; };
    auto b_src = [&](int jb) {
        return MODE == 0 ? B + (long)cb * TB * np + (long)jb * TB             // tile (cb, jb) of M: rows = i, columns = j
             : MODE == 1 ? B + (long)jb * TB * np + (long)cb * TB             // tile (jb, cb) of M: rows = j, columns = i
                         : A + (long)cb * TB * np + (long)jb * TB;
    };
    TileRegs ra, rbb;
    if (j0 < j1) { fetch_tile(ra, a_src(j0), np); fetch_tile(rbb, b_src(j0), np); }
    for (int jb = j0; jb < j1; ++jb) {
        __syncthreads();
        store_tile(As, ra, false);
        store_tile(Bs, rbb, MODE != 2 && jb == cb);
        __syncthreads();
        if (jb + 1 < j1) { fetch_tile(ra, a_src(jb + 1), np); fetch_tile(rbb, b_src(jb + 1), np); }
        if (MODE == 1) mma_tile<false>(As, Bs, acc);
        else mma_tile<true>(As, Bs, acc);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wi = wv >> 1, wj = wv & 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rb * TB + wi * 32 + a * 16 + (lane >> 4) + 4 * q, c = cb * TB + wj * 32 + b * 16 + (lane & 15);
                C[(long)r * ldc + c] = acc[a][b][q];
            }
}
// C = sum of the nsplit partial products of the slab (fixed order); MODE 2: the lower tiles only, plus the diagonal term
template <int MODE>
__global__ __launch_bounds__(256) void k_cap_fold(const double* __restrict__ part, int nsplit, long slab_stride, double* __restrict__ C, int rows, int ldc,
                                                  const double* __restrict__ X, int k, CapLanes L) {
    if (L.mask && !L.mask[blockIdx.z]) return;
    if (blockIdx.z) { const size_t off = (size_t)blockIdx.z * L.lane_bytes; part = cap_at(part, off); C = cap_at(C, off); X = cap_at(X, off); }
    if (MODE == 2) k = cap_k(L, blockIdx.z, k);
    const int r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows || c >= ldc) return;
    if (MODE == 2 && (c / TB) > (r / TB)) return;
    double v = 0.0;
    for (int z = 0; z < nsplit; ++z) v += part[(long)z * slab_stride + (long)r * ldc + c];
    if (MODE == 2 && r == c) v += r < k ? 1.0 / X[r] : 1.0;
    C[(long)r * ldc + c] = v;
}

// rw = a + b  (NV vectors of np entries, stride ldv; entries past n are zeroed)
template <int NV>
__global__ void k_cap_add(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, int n, int np, int ldv, CapLanes L) {
    if (L.mask && !L.mask[blockIdx.z]) return;
    if (blockIdx.z) { const size_t off = (size_t)blockIdx.z * L.lane_bytes; a = cap_at(a, off); b = cap_at(b, off); out = cap_at(out, off); }
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= np) return;
#pragma unroll
    for (int v = 0; v < NV; ++v) out[(long)v * ldv + j] = j < n ? a[(long)v * ldv + j] + b[(long)v * ldv + j] : 0.0;
}
// w[v][r] = U[r] . y[v] - t[v][r]  for r < k (one wave per strong direction), 0 for the padding r in [k, kp)
template <int NV>
__global__ __launch_bounds__(256) void k_cap_uy(const double* __restrict__ U, int k, int kp, int n, int np, const double* __restrict__ y, int ldv,
                                                const double* __restrict__ t, double* __restrict__ w, int ldk, CapLanes L) {
    if (L.mask && !L.mask[blockIdx.z]) return;
    if (blockIdx.z) { const size_t off = (size_t)blockIdx.z * L.lane_bytes; U = cap_at(U, off); y = cap_at(y, off); t = cap_at(t, off); w = cap_at(w, off); }
    k = cap_k(L, blockIdx.z, k);
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= kp) return;
    if (r >= k) {
        if (lane == 0)
#pragma unroll
            for (int v = 0; v < NV; ++v) w[(long)v * ldk + r] = 0.0;
        return;
    }
    const double* u = U + (long)r * np;
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    for (int j = lane; j < n; j += 64) {
        const double uj = u[j];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] += uj * y[(long)v * ldv + j];
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const double s = wave_sum(acc[v]);
        if (lane == 0) w[(long)v * ldk + r] = s - t[(long)v * ldk + r];
    }
}
// dx[v][i] = y[v][i] - sum_{r < k} Zt[r][i] zeta[v][r] : 32 columns x 32 row groups per block of 1024 threads (a wave reads two
// 256-byte row segments), the groups' sums folded in a fixed order.  (The first version -- 64 columns x 4 groups, 17 blocks, 147
// dependent multiply-adds per thread at k = 588 -- took 75-80 us a call, four calls per iteration: 17 % of an iteration.)
constexpr int DXC = 32, DXG = 32;
template <int NV>
__global__ __launch_bounds__(1024) void k_cap_dx(const double* __restrict__ Zt, int k, int n, int np, const double* __restrict__ zeta, int ldk,
                                                 const double* __restrict__ y, double* __restrict__ dx, int ldv, CapLanes L) {
    __shared__ double part[NV][DXG][DXC + 1];
    if (L.mask && !L.mask[blockIdx.z]) return;
    if (blockIdx.z) { const size_t off = (size_t)blockIdx.z * L.lane_bytes; Zt = cap_at(Zt, off); zeta = cap_at(zeta, off); y = cap_at(y, off); dx = cap_at(dx, off); }
    k = cap_k(L, blockIdx.z, k);
    const int c = threadIdx.x & (DXC - 1), g = threadIdx.x / DXC, i = blockIdx.x * DXC + c;
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    if (i < n)
        for (int r = g; r < k; r += DXG) {
            const double z = Zt[(long)r * np + i];
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] += z * zeta[(long)v * ldk + r];
        }
#pragma unroll
    for (int v = 0; v < NV; ++v) part[v][g][c] = acc[v];
    __syncthreads();
    if (g == 0 && i < np) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            double t = 0.0;
            for (int q = 0; q < DXG; ++q) t += part[v][q][c];
            dx[(long)v * ldv + i] = i < n ? y[(long)v * ldv + i] - t : 0.0;
        }
    }
}
__global__ void k_cap_flag_add(int* __restrict__ flag, const int* __restrict__ more, CapLanes L) {
    if (L.mask && !L.mask[blockIdx.z]) return;
    if (blockIdx.z) { const size_t off = (size_t)blockIdx.z * L.lane_bytes; flag = cap_at(flag, off); more = cap_at(more, off); }
    flag[0] += more[0];
}
}  // namespace

// Yt = U M', Zt = Yt M, S = Yt Yt' + X^-1 (lower tiles; padding rows k .. kp-1 get a unit diagonal).  U: kp x np with zero rows from k on;
// M: the inverse Cholesky factor of H_w (lower triangle valid); S: kp x kp.
size_t cap_part_doubles(int kmax, int np) { return (size_t)CAP_SPLIT * kmax * (size_t)std::max(np, kmax); }
void cap_build_launch(const double* U, int k, int kp, int np, const double* M, const double* X, double* Yt, double* Zt, double* S, double* part,
                      hipStream_t st, int nlanes, size_t lane_bytes, const int* mask, const int* kcnt) {
    const int kb = kp / TB;
    const long sA = (long)kp * np, sS = (long)kp * kp;
    const CapLanes L{lane_bytes, mask, kcnt};
    const dim3 g(np / TB, kb, CAP_SPLIT * nlanes), gf(cdiv(np, 256), kp, nlanes), gs(cdiv(kp, 256), kp, nlanes);
    hipLaunchKernelGGL(k_cap_gemm<0>, g, dim3(256), 0, st, U, M, part, np, np, sA, CAP_SPLIT, L);
    hipLaunchKernelGGL(k_cap_fold<0>, gf, dim3(256), 0, st, part, CAP_SPLIT, sA, Yt, kp, np, (const double*)nullptr, 0, L);
    hipLaunchKernelGGL(k_cap_gemm<1>, g, dim3(256), 0, st, Yt, M, part, np, np, sA, CAP_SPLIT, L);
    hipLaunchKernelGGL(k_cap_fold<1>, gf, dim3(256), 0, st, part, CAP_SPLIT, sA, Zt, kp, np, (const double*)nullptr, 0, L);
    hipLaunchKernelGGL(k_cap_gemm<2>, dim3(kb * (kb + 1) / 2, 1, CAP_SPLIT * nlanes), dim3(256), 0, st, Yt, (const double*)nullptr, part, np, kp, sS, CAP_SPLIT, L);
    hipLaunchKernelGGL(k_cap_fold<2>, gs, dim3(256), 0, st, part, CAP_SPLIT, sS, S, kp, kp, X, k, L);
    MBFIR_HIP(hipGetLastError());                         // (a refused launch -- grid, LDS -- would otherwise surface iterations later as a wrong step)
}
void cap_add_launch(const double* a, const double* b, double* out, int n, int np, int ldv, int nv, hipStream_t st, int nlanes, size_t lane_bytes, const int* mask) {
    const CapLanes L{lane_bytes, mask, nullptr};
    if (nv == 1) hipLaunchKernelGGL(k_cap_add<1>, dim3(cdiv(np, 256), 1, nlanes), dim3(256), 0, st, a, b, out, n, np, ldv, L);
    else hipLaunchKernelGGL(k_cap_add<2>, dim3(cdiv(np, 256), 1, nlanes), dim3(256), 0, st, a, b, out, n, np, ldv, L);
}
void cap_uy_launch(const double* U, int k, int kp, int n, int np, const double* y, int ldv, const double* t, double* w, int ldk, int nv, hipStream_t st,
                   int nlanes, size_t lane_bytes, const int* mask, const int* kcnt) {
    const CapLanes L{lane_bytes, mask, kcnt};
    if (nv == 1) hipLaunchKernelGGL(k_cap_uy<1>, dim3(cdiv(kp, 4), 1, nlanes), dim3(256), 0, st, U, k, kp, n, np, y, ldv, t, w, ldk, L);
    else hipLaunchKernelGGL(k_cap_uy<2>, dim3(cdiv(kp, 4), 1, nlanes), dim3(256), 0, st, U, k, kp, n, np, y, ldv, t, w, ldk, L);
}
void cap_dx_launch(const double* Zt, int k, int n, int np, const double* zeta, int ldk, const double* y, double* dx, int ldv, int nv, hipStream_t st,
                   int nlanes, size_t lane_bytes, const int* mask, const int* kcnt) {
    const CapLanes L{lane_bytes, mask, kcnt};
    if (nv == 1) hipLaunchKernelGGL(k_cap_dx<1>, dim3(np / DXC, 1, nlanes), dim3(1024), 0, st, Zt, k, n, np, zeta, ldk, y, dx, ldv, L);
    else hipLaunchKernelGGL(k_cap_dx<2>, dim3(np / DXC, 1, nlanes), dim3(1024), 0, st, Zt, k, n, np, zeta, ldk, y, dx, ldv, L);
}
void cap_flag_add_launch(int* flag, const int* more, hipStream_t st, int nlanes, size_t lane_bytes, const int* mask) {
    hipLaunchKernelGGL(k_cap_flag_add, dim3(1, 1, nlanes), dim3(1), 0, st, flag, more, CapLanes{lane_bytes, mask, nullptr});
}

}  // namespace mbfir
