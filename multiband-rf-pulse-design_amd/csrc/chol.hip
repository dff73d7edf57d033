// K4: dense KKT factorisation.  H = L L' (blocked right-looking Cholesky, 64-wide panels) with
// M = L^-1 carried along by forward substitution on the identity, so that every later solve is
// two triangular GEMVs (x = M'(M b)) instead of two latency-bound substitutions.
// All fp64; tile products on v_mfma_f64_16x16x4_f64.
#include "dev_common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace mbfir {

constexpr int CB = 64;       // panel width
constexpr int CLD = 66;      // padded LDS leading dimension

// ---- 64x64 MFMA helper: each of the 4 waves owns a 32x32 quadrant (2x2 MFMA blocks) -----------
// acc[a][b] += sum_k  Aop[i][k] * Bop[k][j]   with  Aop[i][k] = As[i][k]  (As row-major [64][CLD])
// and Bop[k][j] = transB ? Bs[j][k] : Bs[k][j].
template <bool TRANSB>
__device__ __forceinline__ void mma64(const double (*As)[CLD], const double (*Bs)[CLD], int kbeg, int kend,
                                      v4d acc[2][2]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wi = wv >> 1, wj = wv & 1;
    for (int k0 = kbeg; k0 < kend; k0 += 4) {
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
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
}

// Visit the (i, j, value) triples of a wave's accumulator quadrant.
template <class F>
__device__ __forceinline__ void acc_foreach(v4d acc[2][2], F f) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wi = wv >> 1, wj = wv & 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                f(wi * 32 + a * 16 + (lane >> 4) + 4 * r, wj * 32 + b * 16 + (lane & 15), acc[a][b][r]);
}

// 64x64 tile -> LDS, all 8 16-byte loads of a thread in flight before the first LDS store
// (256-thread blocks)
__device__ __forceinline__ void load_block(double (*S)[CLD], const double* __restrict__ src, int ld) {
    double2 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = threadIdx.x + 256 * u;
        t[u] = *reinterpret_cast<const double2*>(src + (long)(e >> 5) * ld + 2 * (e & 31));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = threadIdx.x + 256 * u;
        *reinterpret_cast<double2*>(&S[e >> 5][2 * (e & 31)]) = t[u];
    }
}

// =================================================================================================
// Right-looking blocked Cholesky that carries M = L^-1 along, ONE launch per 64-wide panel step.
// Launch k (k = 0 .. nblk) runs four kinds of workgroups side by side:
//
//   P  (panel k)        every block first applies the one outstanding update (panel k-1) to A_kk and
//                       factorises it in LDS (redundantly, 87 kflop; block 0 publishes L_kk); blocks
//                       b > 0 then take 16 rows of one tile A_ik, apply the panel k-1 update to them and
//                       solve  X L_kk' = A_ik  by forward substitution (NOT by multiplying with
//                       inv(L_kk): that is not backward stable and breaks the factorisation on the
//                       near-singular late IPM iterates);
//   T  (trailing)       A_ij -= L_i,k-1 L_j,k-1'   for k < j <= i   (panel k-1; column k is done by P);
//   MS (inverse row)    row block r = k-1 of the inverse:  M_rj = L_rr^-1 (R_rj - L_r,r-1 M_r-1,j),
//                       16 columns of one tile per block (R lives in the M buffer, initialised to I);
//   RU (inverse update) R_ij -= L_i,k-2 M_k-2,j    for i >= k, j <= k-2.
//
// So every tile sees its updates in order, each launch only reads what earlier launches wrote, and
// the dependent chain per panel is  tile product -> potf2 -> substitution  (one launch) instead of
// two launches with two substitutions.
// Lock-step batches (>= 3 designs per launch) run the SPLIT form of the step: only block 0 of a lane factorises
// L_kk; the row blocks, dispatched last in the same launch, do their own preamble and then wait for the lane's
// diagonal block on a flag in global memory (CholStep::phase, panel_block<FROM_IMAGE>).  The only cross-workgroup
// dependency inside a launch is that one: row blocks on the diagonal block of their own lane, which has a lower
// workgroup index and is therefore dispatched first -- no workgroup ever waits for one that has not started.
// M computed this way has the accuracy of a substitution-based inverse (measured: solve residual
// 3e-5 at cond(H)=3e9, same as LAPACK trtri; multiplying explicit 64x64 inverses gives 2e-3).
// Pivot rule: a pivot not above pivtol * H_jj is rounding noise and is replaced by H_jj itself; by
// Cauchy-Schwarz the rest of that Schur-complement column is at noise level too, so the column is
// effectively decoupled and M'M stays a non-singular preconditioner (flag counts the replacements).
// =================================================================================================
typedef double v16d __attribute__((ext_vector_type(16)));   // register-resident 16-vector (an array would go to scratch)

// NOTE on code shape (all measured on MI355X with tools/exp/*.hip):
//  * one wave issues an independent v_fma_f64 every ~2.9 ns, a (uniform select + fma) pair costs
//    18 ns (v_cndmask pairs), a ds_read_b128 costs the CU 13.5 ns whatever the address pattern
//    (the 128 B/clk return path), a dependent mul -> DPP -> fma chain 22 ns, an LDS store ->
//    barrier -> load round trip ~65 ns; a single wave per SIMD issues in order, so whatever is
//    not hidden behind a latency adds up;
//  * so: no per-element selects (finished rows / columns are masked by ZEROS in the broadcast
//    images instead), operands in registers with static indices, as few LDS bytes and as few
//    instructions per pivot as possible, and cross-lane traffic on DPP where the layout allows it.

// ---- forward substitution, 16 lanes per right-hand side ----------------------------------------
// A DPP row (16 lanes) owns one right-hand side: lane lam holds a[t] for t = lam + 16 i in v[i].
// Right-looking: step j forms x_j = a_j / L_jj in lane j & 15, broadcasts it inside the row with
// one v_mov_b64_dpp row_newbcast, and every lane updates its (at most 4) later entries.  The
// factor comes from the zero-padded column image  Lz[j][zpos(t)] = t > j ? L[t][j] : 0, so entries
// that are already final see a zero and need no predicate; they are scaled by 1 / L_tt at the end.
// Fully unrolled (64 steps x ~7 instructions).
constexpr int ZLD = 66;
__device__ __forceinline__ int zpos(int t) { return ((t >> 5) << 5) + 2 * (t & 15) + ((t >> 4) & 1); }

template <int LANE>
__device__ __forceinline__ double row_bcast(double v) {
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + LANE, 0xF, 0xF, true);     // row_newbcast:LANE
}

template <int J>
__device__ __forceinline__ void subst16_steps(const double* __restrict__ lz, double dm0, double dm1, double dm2, double dm3,
                                              double& v0, double& v1, double& v2, double& v3) {
    if constexpr (J < 64) {
        constexpr int I = J >> 4;
        const double cur = I == 0 ? v0 * dm0 : I == 1 ? v1 * dm1 : I == 2 ? v2 * dm2 : v3 * dm3;
        const double x = row_bcast<(J & 15)>(cur);
        const double* lr = lz + J * ZLD;
        if constexpr (I < 2) {
            const double2 lo = *reinterpret_cast<const double2*>(lr);
            if constexpr (I == 0) v0 -= x * lo.x;
            v1 -= x * lo.y;
        }
        const double2 hi = *reinterpret_cast<const double2*>(lr + 32);
        if constexpr (I < 3) v2 -= x * hi.x;
        v3 -= x * hi.y;
        subst16_steps<J + 1>(lz, dm0, dm1, dm2, dm3, v0, v1, v2, v3);
    }
}

// Lz: image base; dinv[j] = 1 / L_jj.  On entry v[i] = a[lam + 16 i], on exit x[lam + 16 i].
__device__ __forceinline__ void subst16(const double* __restrict__ Lz, const double* __restrict__ dinv, double (&v)[4]) {
    const int lam = threadIdx.x & 15;
    const double dm0 = dinv[lam], dm1 = dinv[lam + 16], dm2 = dinv[lam + 32], dm3 = dinv[lam + 48];
    double v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    subst16_steps<0>(Lz + 2 * lam, dm0, dm1, dm2, dm3, v0, v1, v2, v3);
    v[0] = v0 * dm0; v[1] = v1 * dm1; v[2] = v2 * dm2; v[3] = v3 * dm3;
}

// ---- factorisation of the 64x64 diagonal block: four 16-column slabs -------------------------------
// The trailing matrix lives in MFMA accumulators (the ten lower 16x16 tiles, dealt to the 4 waves by
// the table below); per slab
//   (1) the owners of its tiles put the slab (64 x 16) into the LDS array LB,
//   (2) ONE wave eliminates it right-looking with lane = row and the 16 columns in registers: the
//       pivot-row values come from v_readlane into SGPRs (no LDS, no barrier inside a slab; 99 ns
//       per pivot measured, against 275 ns for an LDS broadcast + barrier per pivot),
//   (3) the other tiles get their rank-16 update on the matrix cores (operands: the finished slab,
//       unscaled from LB and scaled by -1/pivot from LS).
// Square-root free (S_rc -= S_rj S_cj / p_j, 1/p from v_rcp_f64 + one Newton step); LB ends up
// holding S_rc for the whole lower triangle and piv[] the pivots; the caller applies
// L_rc = S_rc / sqrt(p_c).  Pivot rule as described at the top of this file.
constexpr int LBLD = 65, LSLD = 17;

// tiles of wave w: (ti, tj) pairs, 4 bits each (ti | tj << 2), count in bits 12..
//   w0: (0,0) (2,1) (3,2)   w1: (1,0) (2,2) (3,3)   w2: (1,1) (3,0)   w3: (2,0) (3,1)
__device__ __forceinline__ unsigned wave_tiles(int wv) {
    return wv == 0 ? (3u << 12 | 0x0u | 0x6u << 4 | 0xBu << 8)
         : wv == 1 ? (3u << 12 | 0x1u | 0xAu << 4 | 0xFu << 8)
         : wv == 2 ? (2u << 12 | 0x5u | 0x3u << 4)
                   : (2u << 12 | 0x2u | 0x7u << 4);
}

__device__ __forceinline__ double rdlane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// the 16 pivots of one slab; t[c] = S[row = lane][16 b + c], base = 16 b; rp = 1 / pivot, pv = pivot
template <int J>
__device__ __forceinline__ void slab_steps(double (&t)[16], double (&rp)[16], double (&pv)[16], const double* dsh,
                                           int base, double pivtol, int& nbad) {
    if constexpr (J < 16) {
        double p = rdlane(t[J], base + J);
        const double dj = dsh[base + J];
        const bool bad = !(p > pivtol * dj);
        p = bad ? fmax(dj, 1e-300) : p;
        nbad += bad;
        double rcp = __builtin_amdgcn_rcp(p);             // ~26 bits
        rcp = rcp * fma(-p, rcp, 2.0);                    // 1/p to rounding
        const double l = t[J] * rcp;
#pragma unroll
        for (int c = J + 1; c < 16; ++c) t[c] -= l * rdlane(t[J], base + c);
        rp[J] = rcp; pv[J] = p;
        slab_steps<J + 1>(t, rp, pv, dsh, base, pivtol, nbad);
    }
}

// acc[q]: the wave's tiles of S (MFMA D layout).  LB: 64 x LBLD, LS: 64 x LSLD, piv: 64.
// Lz / dinv: on exit the zero-padded column image of L (see subst16) and 1 / diag(L); the image of a
// finished slab is written by the three idle waves while wave 0 eliminates the next one.
__device__ __forceinline__ void slab_image(const double* LB, const double* piv, double* Lz, double* dinv, int b, int e0,
                                           int estride) {
    for (int e = e0; e < 16 * CB; e += estride) {
        const int r = e & 63, c = 16 * b + (e >> 6);
        const double p = piv[c];
        double y = __builtin_amdgcn_rsq(p);               // 1 / sqrt(pivot): v_rsq_f64 + two Newton steps
        y = y * (1.5 - 0.5 * p * y * y);
        y = y * (1.5 - 0.5 * p * y * y);
        Lz[c * ZLD + zpos(r)] = r > c ? LB[r * LBLD + c] * y : 0.0;
        if (r == 0) dinv[c] = y;
    }
}

#if defined(CHOL_TRACE) && defined(CHOL_TRACE_D)
__device__ long long g_trace2[64];
__device__ int g_trace2_k;
#define TRACE2(slot) if (threadIdx.x == 0 && blockIdx.x == 0 && b == 1 && g_trace2_k) g_trace2[(slot)] = __builtin_amdgcn_s_memrealtime();
#else
#define TRACE2(slot)
#endif
__device__ __forceinline__ void potf2_slabs(v4d (&acc)[3], unsigned tiles, double* LB, double* LS, const double* dsh,
                                            double* piv, double* Lz, double* dinv, double pivtol, int* flag, bool count) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4, nt = int(tiles >> 12);
    int nbad = 0;
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
        TRACE2(0)
#if defined(CHOL_TRACE) && defined(CHOL_TRACE_D)
        if (threadIdx.x == 0 && blockIdx.x == 0 && b == 2 && g_trace2_k) g_trace2[6] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int q = 0; q < 3; ++q) {                     // (1) slab b -> LB
            const int ti = (tiles >> (4 * q)) & 3, tj = (tiles >> (4 * q + 2)) & 3;
            if (q < nt && tj == b) {
#pragma unroll
                for (int r = 0; r < 4; ++r) LB[(16 * ti + g + 4 * r) * LBLD + 16 * b + m] = acc[q][r];
            }
        }
        __syncthreads();
        TRACE2(1)
        if (wv == 0) {                                    // (2) eliminate: lane = row
            double t[16], rp[16], pv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) t[c] = LB[lane * LBLD + 16 * b + c];
            TRACE2(2)
            slab_steps<0>(t, rp, pv, dsh, 16 * b, pivtol, nbad);
            TRACE2(3)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                LB[lane * LBLD + 16 * b + c] = t[c];
                LS[lane * LSLD + c] = -t[c] * rp[c];
            }
            TRACE2(4)
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < 16; ++c) piv[16 * b + c] = pv[c];
            }
        } else if (b >= 1) {
            slab_image(LB, piv, Lz, dinv, b - 1, threadIdx.x - 64, 192);
        }
        __syncthreads();
        TRACE2(5)
        if (b == 3) break;
#pragma unroll
        for (int q = 0; q < 3; ++q) {                     // (3) rank-16 update of the tiles right of the slab
            const int ti = (tiles >> (4 * q)) & 3, tj = (tiles >> (4 * q + 2)) & 3;
            if (q < nt && tj > b) {
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4)
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(LS[(16 * ti + m) * LSLD + k0 + g],
                                                                  LB[(16 * tj + m) * LBLD + 16 * b + k0 + g], acc[q], 0, 0, 0);
            }
        }
    }
    { const int b = 1; TRACE2(7) }
    slab_image(LB, piv, Lz, dinv, 3, threadIdx.x, 256);
    { const int b = 1; TRACE2(8) }
    if (count && threadIdx.x == 0 && nbad) atomicAdd(flag, nbad);
}

__device__ __forceinline__ void tile_decode(int t, int& ti, int& tj) {
    ti = int((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((long)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while ((long)ti * (ti + 1) / 2 > t) --ti;
    tj = t - ti * (ti + 1) / 2;
}

// C_tile (64x64 at dst) -= A_tile * B_tile (TRANSB: B_tile') through the MFMA helper
// first: the tile has not been written in this factorisation yet -- its old content (the previous build's) counts as 0
template <bool TRANSB>
__device__ __forceinline__ void tile_update(double* smem, const double* __restrict__ Ag, const double* __restrict__ Bg,
                                            double* __restrict__ dst, int np, bool first = false) {
    double(*P)[CLD] = reinterpret_cast<double(*)[CLD]>(smem);
    double(*Q)[CLD] = reinterpret_cast<double(*)[CLD]>(smem + CB * CLD);
    load_block(P, Ag, np);
    load_block(Q, Bg, np);
    __syncthreads();
    v4d acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
    mma64<TRANSB>(P, Q, 0, CB, acc);
    if (first) acc_foreach(acc, [&](int i, int j, double v) { dst[(long)i * np + j] = -v; });
    else acc_foreach(acc, [&](int i, int j, double v) { dst[(long)i * np + j] -= v; });
}

#ifdef CHOL_TRACE
__device__ long long g_trace[16 * 32];
#ifdef CHOL_TRACE_D      /* the diagonal block of lane 0 (workgroup 0 of a kind-major grid) */
#define TRACE(slot) if (threadIdx.x == 0 && blockIdx.x == 0) g_trace[a.k * 16 + (slot)] = __builtin_amdgcn_s_memrealtime();
#else
#define TRACE(slot) if (threadIdx.x == 0 && b == 1) g_trace[a.k * 16 + (slot)] = __builtin_amdgcn_s_memrealtime();
#endif
#else
#define TRACE(slot)
#endif

struct CholStep {
    double* H; double* M; double* Mt; int np, nblk, k;   // Mt: written by the inverse-row blocks when not null
    const double* d0; double pivtol;
    double* Dfac;            // per panel: 64x64 zero-padded column image of L_kk (see subst16)
    double* dinvG;           // 1 / diag(L)
    int* flag;
    int* sync;               // nblk ints per lane: sync[k] = 1 once the image of L_kk is in Dfac (merged split step)
    int nP, nMS, nT, nR;
    size_t lane_bytes;       // lock-step batch: every pointer moves by lane * lane_bytes (lane from the block index,
                             // see k_chol_step)
    int nlanes;
    const int* mask;         // nlanes ints (or null): lanes switched off
    int phase;               // 0: one launch per panel step, every row block factorises L_kk itself (lowest latency,
                             //    one design); 1: split step for lock-step batches in ONE launch -- the diagonal block
                             //    (one per lane) publishes the image of L_kk in Dfac and raises sync[k]; the row
                             //    blocks, dispatched last, wait for it instead of repeating its 64 pivots;
                             //    3 / 2: the same in two launches (no flag)
};

constexpr int YLD = 65;                                  // staging tiles that are read one row per lane
constexpr int R0 = 0, R1 = CB * CLD, R2 = 2 * CB * CLD, R3 = R2 + 1152;
constexpr int STEP_LDS = R3 + 336;                       // 79.4 KB

// ---- in-launch hand-offs between workgroups (cdna_hip_programming.md guideline 16, recipe R1) -------------------
// A pivot counter at or above CHOL_SYNC_LOST (dev_common.h) means a bounded flag poll expired: the factorisation is void.
constexpr int CHOL_SPIN_LIMIT = 1 << 21;                  // x s_sleep(4) + one L2 round trip: several seconds

__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ONE lane polls ONE word (relaxed, agent scope: an sc1 load) until it reaches `want`; false + sentinel on expiry
__device__ __forceinline__ bool wait_flag(const int* word, int want, int* pivflag) {
    int spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins >= CHOL_SPIN_LIMIT) { atomicAdd(pivflag, CHOL_SYNC_LOST); return false; }
        __builtin_amdgcn_s_sleep(4);
    }
    return true;
}

template <bool FROM_IMAGE>
__device__ __forceinline__ void panel_block(const CholStep& a, int b, double* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int k = a.k, np = a.np;
    double* H = a.H;
    const long kk = (long)k * CB;
    double(*X)[CLD] = reinterpret_cast<double(*)[CLD]>(smem + R0);     // L_k,k-1, later the image of L_kk
    double* Y = smem + R1;                                             // staging (stride YLD)
    double(*AR)[CLD] = reinterpret_cast<double(*)[CLD]>(smem + R2);    // 16 rows of L_i,k-1
    double* dsh = smem + R3;                              // original diagonal of this block (64)
    double* dinv = dsh + CB;                              // pivots, then 1 / L_jj (64)
    const bool rows = b > 0;
    const long r0 = rows ? (long)(k + 1 + (b - 1) / 4) * CB + 16 * ((b - 1) & 3) : 0;
    // the wave's tiles of S = A_kk - L_k,k-1 L_k,k-1' (lower triangle, 16x16 tiles, MFMA D layout)
    const unsigned tiles = wave_tiles(wv);
    const int nt = int(tiles >> 12), m16 = lane & 15, g4 = lane >> 4;
    v4d acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    v4d accC = {0, 0, 0, 0};
    // the tiles of A_kk / A_ik the products are subtracted from: loads issued first, used after the products
    double hv[3][4], hc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int ti = (tiles >> (4 * q)) & 3, tj = (tiles >> (4 * q + 2)) & 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[q][r] = (!FROM_IMAGE && q < nt) ? H[(kk + 16 * ti + g4 + 4 * r) * np + kk + 16 * tj + m16] : 0.0;
    }
    if (rows) {
#pragma unroll
        for (int r = 0; r < 4; ++r) hc[r] = H[(r0 + g4 + 4 * r) * np + kk + 16 * wv + m16];
    }
    TRACE(0)
    if (k > 0) {
        const long km = kk - CB;
        load_block(X, H + kk * np + km, np);
        if (rows) {
            const double2 t0 = *reinterpret_cast<const double2*>(H + (r0 + (tid >> 5)) * np + km + 2 * (tid & 31));
            const double2 t1 = *reinterpret_cast<const double2*>(H + (r0 + 8 + (tid >> 5)) * np + km + 2 * (tid & 31));
            *reinterpret_cast<double2*>(&AR[tid >> 5][2 * (tid & 31)]) = t0;
            *reinterpret_cast<double2*>(&AR[8 + (tid >> 5)][2 * (tid & 31)]) = t1;
        }
        __syncthreads();
        TRACE(1)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int ti = (tiles >> (4 * q)) & 3, tj = (tiles >> (4 * q + 2)) & 3;
            if (!FROM_IMAGE && q < nt) {
#pragma unroll 4
                for (int k0 = 0; k0 < CB; k0 += 4) {
                    const int kx = k0 + g4;
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[16 * ti + m16][kx], X[16 * tj + m16][kx], acc[q], 0, 0, 0);
                }
            }
        }
        if (rows) {
#pragma unroll 4
            for (int k0 = 0; k0 < CB; k0 += 4) {
                const int kx = k0 + g4;
                accC = __builtin_amdgcn_mfma_f64_16x16x4f64(AR[m16][kx], X[16 * wv + m16][kx], accC, 0, 0, 0);
            }
        }
    }
    TRACE(2)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (q < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q][r] = hv[q][r] - acc[q][r];
        }
    }
    if (tid < CB) dsh[tid] = a.d0[kk + tid];
    double* LB = smem + R1;                               // 64 x LBLD: slabs, then S_rc of the whole lower triangle
    double* LS = smem + R2;                               // 64 x LSLD: the current slab scaled by -1 / pivot
    __syncthreads();                                      // also: everybody is done with X and AR
    TRACE(3)
    double* Lz = smem + R0;
    if (FROM_IMAGE) {                                     // the image of L_kk and 1 / diag(L_kk) as the diagonal block left them
        // the diagonal block of this lane runs in the SAME launch: wait for its flag.  The poll is bounded; a poll that
        // expires raises CHOL_SYNC_LOST in the lane's pivot counter, which the host turns into an error (the numbers
        // this block goes on to produce from the stale image are never used) -- no hang, no silent wrong factor.
        if (a.phase == 1) {
            if (tid == 0) wait_flag(a.sync + k, 1, a.flag);
            __syncthreads();
        }
        // Hand-off by write-through stores and L1-bypassing loads (cdna_hip_programming.md guideline 16, R1): every
        // byte of the image is stored sc1 by the diagonal block, each storing wave drains its stores (s_waitcnt
        // vmcnt(0)) before the workgroup barrier behind which ONE lane stores the flag, the flag is polled by ONE lane
        // with sc1 loads, the other waves pass a barrier after the poll, and every load of the image is an sc1 load
        // (served past this CU's L1, which another CU's stores never refresh) -- no L2 write-back / invalidate.
        double t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = __hip_atomic_load(a.Dfac + kk * CB + tid + 256 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u;
            Lz[(e >> 6) * ZLD + (e & 63)] = t[u];
        }
        if (tid < CB) dinv[tid] = __hip_atomic_load(a.dinvG + kk + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        potf2_slabs(acc, tiles, LB, LS, dsh, dsh + 2 * CB, Lz, dinv, a.pivtol, a.flag, b == 0);
    }
    TRACE(4)
    __syncthreads();                                      // image complete; LB is free again (Y aliases it)
    if (rows) {                                           // updated rows of A_ik (MFMA layout -> one row per DPP row)
        const int cc = 16 * wv + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = (lane >> 4) + 4 * r;
            Y[rr * YLD + cc] = hc[r] - accC[r];
        }
    }
    __syncthreads();
    if (!rows) {
        for (int e = tid; e < CB * CB; e += 256) __hip_atomic_store(a.Dfac + kk * CB + e, Lz[(e >> 6) * ZLD + (e & 63)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < CB) __hip_atomic_store(a.dinvG + kk + tid, dinv[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.phase == 1) {                               // merged split step: release the row blocks of this lane
            drain_stores();                               // every storing wave: its sc1 stores have left the CU ...
            __syncthreads();                              // ... before the one lane that signals for all of them does
            if (tid == 0) __hip_atomic_store(a.sync + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        TRACE(5)
        return;
    }
    const int rho = tid >> 4, lam = tid & 15;
    double v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = Y[rho * YLD + lam + 16 * i];
    TRACE(5)
    subst16(Lz, dinv, v);
    TRACE(6)
#pragma unroll
    for (int i = 0; i < 4; ++i) H[(r0 + rho) * np + kk + lam + 16 * i] = v[i];
    TRACE(7)
}

__device__ __forceinline__ void minv_block(const CholStep& a, int b, double* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int np = a.np, r = a.k - 1, j = b >> 2, c0 = 16 * (b & 3);
    const long kr = (long)r * CB;
    double* M = a.M;
    double* Lz = smem + R0;
    double(*A2)[CLD] = reinterpret_cast<double(*)[CLD]>(smem + R1);    // L_r,r-1
    double(*Bs)[17] = reinterpret_cast<double(*)[17]>(smem + R2);      // 16 columns of M_r-1,j
    double* Ct = smem + R1;                                            // staging [column][row], stride YLD
    double* dinv = smem + R3;
    {
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const double2*>(a.Dfac + kr * CB + 2 * (tid + 256 * u));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u;
            *reinterpret_cast<double2*>(Lz + (e >> 5) * ZLD + 2 * (e & 31)) = t[u];
        }
    }
    if (tid < CB) dinv[tid] = a.dinvG[kr + tid];
    v4d acc = {0, 0, 0, 0};
    if (j < r) {
        load_block(A2, a.H + kr * np + kr - CB, np);
        {
            double t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = tid + 256 * u; t[u] = M[(kr - CB + (e >> 4)) * np + (long)j * CB + c0 + (e & 15)]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = tid + 256 * u; Bs[e >> 4][e & 15] = t[u]; }
        }
        __syncthreads();
#pragma unroll 4
        for (int k0 = 0; k0 < CB; k0 += 4) {
            const int kx = k0 + (lane >> 4);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A2[16 * wv + (lane & 15)][kx], Bs[kx][lane & 15], acc, 0, 0, 0);
        }
    }
    __syncthreads();
    {
        // R_rj as the updates left it; tiles no update ever reached hold the previous build's numbers and stand for
        // their initial value: the identity on the diagonal (j == r), zero next to it (j == r - 1)
        const int c = lane & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = 16 * wv + (lane >> 4) + 4 * q;
            const double r0 = j == r ? (t == c0 + c ? 1.0 : 0.0) : j == r - 1 ? 0.0 : M[(kr + t) * np + (long)j * CB + c0 + c];
            Ct[c * YLD + t] = r0 - acc[q];
        }
    }
    __syncthreads();
    const int rho = tid >> 4, lam = tid & 15;
    double v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = Ct[rho * YLD + lam + 16 * i];
    subst16(Lz, dinv, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) Ct[rho * YLD + lam + 16 * i] = v[i];
    __syncthreads();
    const int t = tid >> 2, c4 = (tid & 3) * 4;
    double* dst = M + (kr + t) * np + (long)j * CB + c0 + c4;
    *reinterpret_cast<double2*>(dst) = make_double2(Ct[c4 * YLD + t], Ct[(c4 + 1) * YLD + t]);
    *reinterpret_cast<double2*>(dst + 2) = make_double2(Ct[(c4 + 2) * YLD + t], Ct[(c4 + 3) * YLD + t]);
    if (a.Mt) {                                           // the transpose for the second triangular GEMV, straight from the staging tile
        const int c = tid >> 4, t4 = (tid & 15) * 4;      // 16 rows of Mt (columns of this block) x 64 entries
        double* dt = a.Mt + ((long)j * CB + c0 + c) * np + kr + t4;
        dt[0] = Ct[c * YLD + t4]; dt[1] = Ct[c * YLD + t4 + 1]; dt[2] = Ct[c * YLD + t4 + 2]; dt[3] = Ct[c * YLD + t4 + 3];
    }
}

template <class T>
__device__ __forceinline__ T* lane_at(T* p, size_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(p)) + off); }

__global__ __launch_bounds__(256) void k_chol_step(CholStep a) {
    __shared__ __attribute__((aligned(16))) double smem[STEP_LDS];
    // 1-D grid, KIND-major over the lanes: the hardware hands out workgroups in index order, so all lanes' panel blocks
    // (the step's critical path: 64 sequential pivots) come first, then all lanes' inverse rows (64-step substitutions),
    // then the tile updates -- with the lane as the slow grid dimension the last lane's diagonal block queued behind
    // ~700 other workgroups of its own launch (and behind the other units' once several share the chip)
    // Merged split step (phase 1): the row blocks come LAST -- they spin on the diagonal block's flag after their own
    // preamble (tile loads, the panel k-1 update of their rows), and must not hold the CU slots the MS / T / RU blocks
    // could use meanwhile.  nR = row blocks per lane (phase 1), nP = panel blocks dispatched first (1, or 1 + 4 nrem
    // in the fused single-design step where every row block factorises L_kk itself).
    int lane, b, kind = 3;                                // kind: 0 P, 1 MS, 2 T, 3 RU, 4 R
    {
        const int nl = a.nlanes;
        const int total = int(gridDim.x) / nl;
        const int nRU = total - a.nP - a.nMS - a.nT - a.nR;
        int id = blockIdx.x;
        const int seg[5] = {a.nP, a.nMS, a.nT, nRU, a.nR};
        lane = 0; b = 0;
        bool found = false;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            if (!found) {
                if (id < seg[q] * nl) { lane = id % nl; b = id / nl; kind = q; found = true; }
                else id -= seg[q] * nl;
            }
        }
    }
    if (a.mask && !a.mask[lane]) return;
    if (lane) {
        const size_t off = (size_t)lane * a.lane_bytes;
        a.H = lane_at(a.H, off); a.M = lane_at(a.M, off); a.d0 = lane_at(a.d0, off); a.Dfac = lane_at(a.Dfac, off);
        a.dinvG = lane_at(a.dinvG, off); a.flag = lane_at(a.flag, off); a.sync = lane_at(a.sync, off);
        if (a.Mt) a.Mt = lane_at(a.Mt, off);
    }
    if (kind == 4) { panel_block<true>(a, b + 1, smem); return; }         // row blocks of a split step
    if (kind == 0) { panel_block<false>(a, b, smem); return; }
    if (kind == 1) { minv_block(a, b, smem); return; }
    const int k = a.k, np = a.np;
    if (kind == 2) {                                      // trailing update with panel k-1, columns > k
        int ti, tj;
        tile_decode(b, ti, tj);
        const long i0 = (long)(k + 1 + ti) * CB, j0 = (long)(k + 1 + tj) * CB, km = (long)(k - 1) * CB;
        tile_update<true>(smem, a.H + i0 * np + km, a.H + j0 * np + km, a.H + i0 * np + j0, np);
        return;
    }
    // R_ij -= L_i,k-2 M_k-2,j  (i >= k, j <= k-2)
    const int i = k + b / (k - 1), j = b % (k - 1);
    const long mm = (long)(k - 2) * CB;
    // (the inverse factor is not initialised: R = I is implied -- a tile's first update, by panel j = k - 2, WRITES it)
    tile_update<false>(smem, a.H + (long)i * CB * np + mm, a.M + mm * np + (long)j * CB, a.M + (long)i * CB * np + (long)j * CB, np, j == k - 2);
}

// d0 = diag(H), pivot-replacement counter = 0, panel flags = 0; one launch for all lanes.  The inverse factor is NOT
// initialised any more (64 MB of writes per unit of 8 at np = 1024): the update and inverse-row blocks imply R = I.
__global__ __launch_bounds__(256) void k_chol_init(const double* __restrict__ H, int np, double* __restrict__ d0,
                                                   int* __restrict__ flag, int* __restrict__ sync, size_t lane_bytes, const int* __restrict__ mask) {
    if (mask && !mask[blockIdx.y]) return;
    const size_t off = (size_t)blockIdx.y * lane_bytes;
    H = lane_at(H, off); d0 = lane_at(d0, off); flag = lane_at(flag, off); sync = lane_at(sync, off);
    if (blockIdx.x == 0 && threadIdx.x == 0) flag[0] = 0;
    if (blockIdx.x == 0 && threadIdx.x < np / CB + 1) sync[threadIdx.x] = 0;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < np; j += (long)gridDim.x * 256) d0[j] = H[j * np + j];
}
// also clears the lane's pivot-replacement counter
__global__ void k_diag_copy(const double* __restrict__ H, int np, double* __restrict__ d0, double* __restrict__ M,
                            int* __restrict__ flag, size_t lane_bytes, const int* __restrict__ mask) {
    if (mask && !mask[blockIdx.y]) return;
    const size_t off = (size_t)blockIdx.y * lane_bytes;
    H = lane_at(H, off); d0 = lane_at(d0, off); M = lane_at(M, off); flag = lane_at(flag, off);
    if (blockIdx.x == 0 && threadIdx.x == 0) flag[0] = 0;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < np) { d0[j] = H[(long)j * np + j]; M[(long)j * np + j] = 1.0; }      // R starts as the identity
}

// L (np x np, clean lower triangle) from the factored H and the diagonal-block images
__global__ void k_extract_L(const double* __restrict__ H, int np, const double* __restrict__ Dfac,
                            const double* __restrict__ dinvG, double* __restrict__ Lout) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)np * np) return;
    long i = e / np, j = e - i * np;
    double v = 0.0;
    if (j == i) v = 1.0 / dinvG[i];
    else if (j < i) v = ((i / CB) == (j / CB)) ? Dfac[(i / CB) * CB * CB + (j % CB) * CB + zpos(int(i % CB))] : H[e];
    Lout[e] = v;
}

__global__ void k_transpose(const double* __restrict__ M, double* __restrict__ Mt, int np, size_t lane_bytes,
                            const int* __restrict__ mask) {
    __shared__ double tile[32][33];
    if (mask && !mask[blockIdx.z]) return;
    M = lane_at(M, (size_t)blockIdx.z * lane_bytes); Mt = lane_at(Mt, (size_t)blockIdx.z * lane_bytes);
    int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) tile[r][threadIdx.x] = M[(long)(by + r) * np + bx + threadIdx.x];
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) Mt[(long)(bx + r) * np + by + threadIdx.x] = tile[threadIdx.x][r];
}

int chol_inv_launch(double* H, double* M, double* Mt, double* W1, int np, int* flag, hipStream_t st,
                    double* Lcopy, hipEvent_t e0, hipEvent_t e1, int nlanes, size_t lane_bytes, const int* mask) {
    int launches = 0;
    const int nblk = np / CB;
    // W1 layout: np doubles: original diagonal | 64 np doubles: images of the L_kk blocks | np: 1 / diag(L)
    CholStep a;
    a.H = H; a.M = M; a.np = np; a.nblk = nblk;
    a.d0 = W1; a.Dfac = W1 + np; a.dinvG = W1 + (long)(CB + 1) * np; a.flag = flag;
    a.pivtol = 1e-13;                            // oracle/conic_ipm.py PIVTOL
    a.lane_bytes = lane_bytes; a.mask = mask; a.nlanes = nlanes;
    a.sync = reinterpret_cast<int*>(W1 + (long)(CB + 2) * np);          // behind 1 / diag(L): nblk + 1 ints
    a.Mt = Mt;
    hipLaunchKernelGGL(k_chol_init, dim3(cdiv(np, 256), nlanes), dim3(256), 0, st, H, np, W1, flag, a.sync, lane_bytes, mask);
    if (e0) hipEventRecord(e0, st);
    // Lock-step batches split every step (see CholStep::phase): with several designs in flight the chip is no longer
    // empty, and the 4 * nrem row blocks of a step each repeating the 64-pivot factorisation of L_kk is what fills it.
    // The split costs no launch: the row blocks ride in the same launch, dispatched last, and wait for their lane's
    // diagonal block on a flag in global memory (MBFIR_CHOL_SPLIT=2: the older two-launch form).
    int split = nlanes >= 3 ? 1 : 0;
    if (const char* ev = std::getenv("MBFIR_CHOL_SPLIT")) split = std::atoi(ev);
    for (int k = 0; k <= nblk; ++k) {
        const int nrem = nblk - k - 1;
        a.k = k;
        a.nMS = k >= 1 ? 4 * k : 0;
        a.nT = (k >= 1 && k < nblk) ? nrem * (nrem + 1) / 2 : 0;
        const int nRU = (k >= 2 && k < nblk) ? (nblk - k) * (k - 1) : 0;
        const int rows = k < nblk ? 4 * nrem : 0;
        a.nP = k < nblk ? (split ? 1 : 1 + rows) : 0;
        a.nR = split == 1 ? rows : 0;
        a.phase = split == 1 ? 1 : (split ? 3 : 0);
        hipLaunchKernelGGL(k_chol_step, dim3((a.nP + a.nMS + a.nT + nRU + a.nR) * nlanes), dim3(256), 0, st, a);
        ++launches;
        if (split == 2 && rows > 0) {
            a.phase = 2; a.nP = 0; a.nMS = 0; a.nT = 0; a.nR = rows;
            hipLaunchKernelGGL(k_chol_step, dim3(rows * nlanes), dim3(256), 0, st, a);
            ++launches;
        }
    }
    if (e1) hipEventRecord(e1, st);
    if (Lcopy) hipLaunchKernelGGL(k_extract_L, dim3(cdiv((long)np * np, 256)), dim3(256), 0, st, H, np, a.Dfac, a.dinvG, Lcopy);
    return launches;                                      // k_chol_step launches issued
}

// y[v][i] = sum_j T[i][j] b[v][j] over the stored triangle; one wave per row, 16-byte loads.
template <int NV>
__global__ __launch_bounds__(256) void k_trigemv(const double* __restrict__ T, int np, int upper,
                                                 const double* __restrict__ b, const double* __restrict__ b2,
                                                 double* __restrict__ y, int ldv, size_t lane_bytes,
                                                 const int* __restrict__ mask) {
    if (mask && !mask[blockIdx.y]) return;
    if (blockIdx.y) {
        const size_t off = (size_t)blockIdx.y * lane_bytes;
        T = lane_at(T, off); b = lane_at(b, off); y = lane_at(y, off);
        if (b2) b2 = lane_at(b2, off);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    if (i >= np) return;
    int jlo = upper ? (i & ~1) : 0;
    int jhi = upper ? np : ((i + 2) & ~1);            // exclusive, even
    const double* row = T + (long)i * np;
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0;
    for (int j = jlo + 2 * lane; j < jhi; j += 128) {
        double2 t = *reinterpret_cast<const double2*>(row + j);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            double2 bb = *reinterpret_cast<const double2*>(b + (long)v * ldv + j);
            if (b2) {
                const double2 cc = *reinterpret_cast<const double2*>(b2 + (long)v * ldv + j);
                bb.x += cc.x; bb.y += cc.y;
            }
            acc[v] += t.x * bb.x + t.y * bb.y;
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double s = wave_sum(acc[v]);
        if (lane == 0) y[(long)v * ldv + i] = s;
    }
}

void trigemv_launch(const double* T, int np, int upper, const double* b, double* y, int nv, int ldv,
                    hipStream_t st, const double* b2, int nlanes, size_t lane_bytes, const int* mask) {
    dim3 grid(cdiv(np, 4), nlanes);
    if (nv == 1) hipLaunchKernelGGL(k_trigemv<1>, grid, dim3(256), 0, st, T, np, upper, b, b2, y, ldv, lane_bytes, mask);
    else if (nv == 2) hipLaunchKernelGGL(k_trigemv<2>, grid, dim3(256), 0, st, T, np, upper, b, b2, y, ldv, lane_bytes, mask);
    else throw HipError("trigemv_launch: nv must be 1 or 2");
}

}  // namespace mbfir
