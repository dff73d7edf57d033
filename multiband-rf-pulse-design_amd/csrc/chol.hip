// K4: dense KKT factorisation.  H = L L' (blocked right-looking Cholesky, 64-wide panels) with
// M = L^-1 carried along by forward substitution on the identity, so that every later solve is
// two triangular GEMVs (x = M'(M b)) instead of two latency-bound substitutions.
// All fp64; tile products on v_mfma_f64_16x16x4_f64.
#include "dev_common.h"
#include <algorithm>
#include <mutex>
#include <cstdlib>
#include <type_traits>

namespace mbfir {

constexpr int CB = 64;       // panel width
constexpr int CLD = 66;      // padded LDS leading dimension

// ---- 64x64 MFMA helper: each of the 4 waves owns a 32x32 quadrant (2x2 MFMA blocks) -----------
// acc[a][b] -= sum_k  Aop[i][k] * Bop[k][j]   with  Aop[i][k] = As[i][k]  (As row-major [64][CLD])
// and Bop[k][j] = transB ? Bs[j][k] : Bs[k][j].
// Every tile update of this file ACCUMULATES INTO THE OLD VALUE (round 5): the accumulator starts as the tile's current
// content and the sixteen MFMAs of a 64-deep product, with the A operand negated, run on it -- no product from zero followed
// by a subtraction.  One rounding less per update and half the accumulator registers of a task that keeps a tile in
// registers over several panels; every form of the factorisation does the same, so they stay bit-identical to one another.
// timing experiments (tools/exp only; the results are wrong): -DCHOL_EXP_NO_MFMA the tile products' k-loops are skipped,
// -DCHOL_EXP_NO_SUBST the 64-step substitutions
#ifdef CHOL_EXP_NO_MFMA
#define CHOL_KSTEPS(n) 1
#else
#define CHOL_KSTEPS(n) (n)
#endif
template <bool TRANSB>
__device__ __forceinline__ void mma64(const double (*As)[CLD], const double (*Bs)[CLD], int kbeg, int kend,
                                      v4d acc[2][2]) {
#ifdef CHOL_EXP_NO_MFMA
    kend = kbeg + 4;
#endif
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wi = wv >> 1, wj = wv & 1;
    for (int k0 = kbeg; k0 < kend; k0 += 4) {
        const int k = k0 + (lane >> 4);
        double af[2], bf[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) af[a] = -As[wi * 32 + a * 16 + (lane & 15)][k];
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
            for (int r = 0; r < 4; ++r) {
                double v = acc[a][b][r];
                f(wi * 32 + a * 16 + (lane >> 4) + 4 * r, wj * 32 + b * 16 + (lane & 15), v);
                acc[a][b][r] = v;                         // (f may take its value by reference)
            }
}

// ---- write-through / L1-bypassing accesses for data that passes between workgroups INSIDE one launch ---------------
// (the single-launch factorisation below; cdna_hip_programming.md guideline 16: per-XCD L2s are not coherent with each
// other and a CU's L1 is never refreshed by another CU's stores, so every byte handed over is stored sc1 -- written
// through -- and loaded sc1 -- past the L1 --; 16-byte accesses go through buffer instructions, whose cache policy the
// builtins expose, 8-byte ones through relaxed agent-scope atomics, which lower to global_load / store ... sc1)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
using rsrc_t = decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, short(0), 0, 0));

__device__ __forceinline__ rsrc_t make_rsrc(const double* p) {                       // p: wave-uniform
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned long long lo = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(a))));
    const unsigned long long hi = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(a >> 32))));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(hi << 32 | lo), short(0), 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ double2 ld2_sc1(rsrc_t r, unsigned byte_off) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, int(byte_off), 0, 16);    // aux 16 = sc1
    return make_double2(__hiloint2double(int(v.y), int(v.x)), __hiloint2double(int(v.w), int(v.z)));
}
__device__ __forceinline__ void st2_sc1(rsrc_t r, unsigned byte_off, double2 x) {
    v4u v;
    v.x = unsigned(__double2loint(x.x)); v.y = unsigned(__double2hiint(x.x));
    v.z = unsigned(__double2loint(x.y)); v.w = unsigned(__double2hiint(x.y));
    __builtin_amdgcn_raw_buffer_store_b128(v, r, int(byte_off), 0, 16);
}
// 8-byte forms through a buffer resource: ONE 32-bit per-lane offset register serves every access of a task (the wave-uniform
// part of the address -- a row stride times a small count, a panel -- goes into the scalar offset, the rest into the
// instruction's immediate), where global_load / store would keep a 64-bit address pair per row alive
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double ldb_sc1(rsrc_t r, unsigned voff, unsigned soff) {
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, int(voff), int(soff), 16);
    return __hiloint2double(int(v.y), int(v.x));
}
__device__ __forceinline__ void stb_sc1(rsrc_t r, unsigned voff, unsigned soff, double x) {
    v2u v;
    v.x = unsigned(__double2loint(x)); v.y = unsigned(__double2hiint(x));
    __builtin_amdgcn_raw_buffer_store_b64(v, r, int(voff), int(soff), 16);
}
// ---- hand-offs of the single-launch factorisation (k_chol_dag) ------------------------------------------------------------------
// write-through stores (aux 16 = sc1) and agent-scope counters.  An XCD-local form (every lane owned by one XCD, plain stores kept in
// its L2, L2 atomics) was built and measured in round 5: bit-identical and slower (16 lanes alone 562 -> 610 us); it lives on in
// tools/exp/chol_r5_switches.hip (-DCHOL_XCD_LOCAL=1), not here.
constexpr int DAG_AUX = 16;
__device__ __forceinline__ void dag_st2(rsrc_t r, unsigned byte_off, double2 x) {
    v4u v;
    v.x = unsigned(__double2loint(x.x)); v.y = unsigned(__double2hiint(x.x));
    v.z = unsigned(__double2loint(x.y)); v.w = unsigned(__double2hiint(x.y));
    __builtin_amdgcn_raw_buffer_store_b128(v, r, int(byte_off), 0, DAG_AUX);
}
__device__ __forceinline__ void dag_stb(rsrc_t r, unsigned voff, unsigned soff, double x) {
    v2u v;
    v.x = unsigned(__double2loint(x)); v.y = unsigned(__double2hiint(x));
    __builtin_amdgcn_raw_buffer_store_b64(v, r, int(voff), int(soff), DAG_AUX);
}
__device__ __forceinline__ void dag_st(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int dag_add(int* word, int n) { return __hip_atomic_fetch_add(word, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dag_set(int* word, int v) { __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 64x64 tile -> LDS, all 8 16-byte loads of a thread in flight before the first LDS store
// (256-thread blocks); SC1: the tile was written by another workgroup of this launch
template <bool SC1 = false>
__device__ __forceinline__ void load_block(double (*S)[CLD], const double* __restrict__ src, int ld) {
    double2 t[8];
    if constexpr (SC1) {
        const rsrc_t r = make_rsrc(src);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = threadIdx.x + 256 * u;
            t[u] = ld2_sc1(r, unsigned(((e >> 5) * ld + 2 * (e & 31)) * 8));
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = threadIdx.x + 256 * u;
            t[u] = *reinterpret_cast<const double2*>(src + (long)(e >> 5) * ld + 2 * (e & 31));
        }
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

#ifdef CHOL_EXP_NO_SUBST
constexpr int SUBST_STEPS = 1;
#else
constexpr int SUBST_STEPS = 64;
#endif
template <int J>
__device__ __forceinline__ void subst16_steps(const double* __restrict__ lz, double dm0, double dm1, double dm2, double dm3,
                                              double& v0, double& v1, double& v2, double& v3) {
    if constexpr (J < SUBST_STEPS) {
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

// Two right-hand sides per thread, the two chains interleaved: a substitution step is a dependent mul -> DPP -> fma
// chain of ~22 ns against ~3 ns of issue per instruction, so a second, independent chain rides in the shadow of the
// first (same arithmetic per chain as subst16: bit-identical results)
template <int J>
__device__ __forceinline__ void subst16x2_steps(const double* __restrict__ lz, double dm0, double dm1, double dm2, double dm3,
                                                double& a0, double& a1, double& a2, double& a3, double& b0, double& b1, double& b2, double& b3) {
    if constexpr (J < SUBST_STEPS) {
        constexpr int I = J >> 4;
        // (the image's LDS reads of all 64 steps have constant offsets from one base: left alone, the scheduler may cluster
        //  them far ahead of their steps -- 256 registers' worth -- and the allocator then spills; keep them within 8 steps)
        if constexpr (J % 8 == 0 && J > 0) __builtin_amdgcn_sched_barrier(0);
        const double ca = I == 0 ? a0 * dm0 : I == 1 ? a1 * dm1 : I == 2 ? a2 * dm2 : a3 * dm3;
        const double cb = I == 0 ? b0 * dm0 : I == 1 ? b1 * dm1 : I == 2 ? b2 * dm2 : b3 * dm3;
        const double xa = row_bcast<(J & 15)>(ca), xb = row_bcast<(J & 15)>(cb);
        const double* lr = lz + J * ZLD;
        if constexpr (I < 2) {
            const double2 lo = *reinterpret_cast<const double2*>(lr);
            if constexpr (I == 0) { a0 -= xa * lo.x; b0 -= xb * lo.x; }
            a1 -= xa * lo.y; b1 -= xb * lo.y;
        }
        const double2 hi = *reinterpret_cast<const double2*>(lr + 32);
        if constexpr (I < 3) { a2 -= xa * hi.x; b2 -= xb * hi.x; }
        a3 -= xa * hi.y; b3 -= xb * hi.y;
        subst16x2_steps<J + 1>(lz, dm0, dm1, dm2, dm3, a0, a1, a2, a3, b0, b1, b2, b3);
    }
}
__device__ __forceinline__ void subst16x2(const double* __restrict__ Lz, const double* __restrict__ dinv, double (&va)[4], double (&vb)[4]) {
    const int lam = threadIdx.x & 15;
    const double dm0 = dinv[lam], dm1 = dinv[lam + 16], dm2 = dinv[lam + 32], dm3 = dinv[lam + 48];
    double a0 = va[0], a1 = va[1], a2 = va[2], a3 = va[3], b0 = vb[0], b1 = vb[1], b2 = vb[2], b3 = vb[3];
    subst16x2_steps<0>(Lz + 2 * lam, dm0, dm1, dm2, dm3, a0, a1, a2, a3, b0, b1, b2, b3);
    va[0] = a0 * dm0; va[1] = a1 * dm1; va[2] = a2 * dm2; va[3] = a3 * dm3;
    vb[0] = b0 * dm0; vb[1] = b1 * dm1; vb[2] = b2 * dm2; vb[3] = b3 * dm3;
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

// C_tile (64x64 at dst) -= A_tile * B_tile (TRANSB: B_tile') through the MFMA helper, accumulated into the old tile
// first: the tile has not been written in this factorisation yet -- its old content (the previous build's) counts as 0
// (the per-step forms; the single-launch form's single-tile updates go through strip_update)
template <bool TRANSB>
__device__ __forceinline__ void tile_update(double* smem, const double* __restrict__ Ag, const double* __restrict__ Bg,
                                            double* __restrict__ dst, int np, bool first = false) {
    double(*P)[CLD] = reinterpret_cast<double(*)[CLD]>(smem);
    double(*Q)[CLD] = reinterpret_cast<double(*)[CLD]>(smem + CB * CLD);
    load_block<false>(P, Ag, np);
    load_block<false>(Q, Bg, np);
    v4d acc[2][2];
    acc_foreach(acc, [&](int i, int j, double& v) { v = first ? 0.0 : dst[(long)i * np + j]; });
    __syncthreads();
    mma64<TRANSB>(P, Q, 0, CB, acc);
    acc_foreach(acc, [&](int i, int j, double& v) { dst[(long)i * np + j] = v; });
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
    int* cnt;                // single-launch form (k_chol_dag): the lane's dependency counters, see DagCnt
    int ntasks;              // ... and its number of tasks
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
constexpr int CHOL_SPIN_LIMIT_DEFAULT = 1 << 21;          // x s_sleep(4) + one L2 round trip: several seconds
// test hooks (MBFIR_TEST_LOSE_FLAG, tests/test_switches_gpu.py): the poll bound, and the panel step whose diagonal block
// "forgets" to publish its image -- the row blocks' polls must then expire and the solve must fail loudly
__device__ int g_chol_spin_limit = CHOL_SPIN_LIMIT_DEFAULT;
__device__ int g_chol_lose_step = -1;
#define CHOL_SPIN_LIMIT g_chol_spin_limit

// (drain_stores(): dev_common.h)
// the thread index as a value of its own per task: the compiler then cannot merge the address arithmetic of different task kinds
// and hoist it to the top of the single-launch kernel, where it would have to live -- or spill -- through every task's branch
__device__ __forceinline__ int task_tid() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
// pause between two polls (a back-off that grows with the wait was measured in round 5: no gain)
__device__ __forceinline__ void poll_pause(int) { __builtin_amdgcn_s_sleep(4); }

// ONE lane polls ONE word (relaxed, agent scope: an sc1 load) until it reaches `want`; false + sentinel on expiry
#ifdef CHOL_DAG_STATS
__shared__ long long s_dag_wait;
__shared__ long long s_ph[9];          // [0..7]: ticks per phase of the task (thread 0's view), [8]: the last stamp
#define DAG_WAIT_BEGIN const long long tw0 = __builtin_amdgcn_s_memrealtime();
#define DAG_WAIT_END s_dag_wait += __builtin_amdgcn_s_memrealtime() - tw0;
// phases: 0 ticket + decode, 1 polls, 2 operands landed, 3 products, 4 product epilogue (subtract, stores issued), 5 substitution
// passes, 6 drain + signal, 7 other
#define PH(n) if (threadIdx.x == 0) { const long long ph_now = __builtin_amdgcn_s_memrealtime(); s_ph[(n)] += ph_now - s_ph[8]; s_ph[8] = ph_now; }
#else
#define DAG_WAIT_BEGIN
#define DAG_WAIT_END
#define PH(n)
#endif
__device__ __forceinline__ bool wait_flag(const int* word, int want, int* pivflag) {
    int spins = 0;
    DAG_WAIT_BEGIN
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins >= CHOL_SPIN_LIMIT) { atomicAdd(pivflag, CHOL_SYNC_LOST); return false; }
        poll_pause(spins);
    }
    DAG_WAIT_END
    return true;
}

// up to three words at once (all loads in flight together; a null word counts as reached)
__device__ __forceinline__ bool wait_flags(const int* w0, int want0, const int* w1, int want1, const int* w2, int want2, int* pivflag) {
    int spins = 0;
    DAG_WAIT_BEGIN
    for (;;) {
        const int v0 = w0 ? __hip_atomic_load(w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want0;
        const int v1 = w1 ? __hip_atomic_load(w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want1;
        const int v2 = w2 ? __hip_atomic_load(w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want2;
        if (v0 >= want0 && v1 >= want1 && v2 >= want2) { DAG_WAIT_END return true; }
        if (++spins >= CHOL_SPIN_LIMIT) { atomicAdd(pivflag, CHOL_SYNC_LOST); return false; }
        poll_pause(spins);
    }
}
// signal for the whole workgroup: every storing wave has drained its write-through stores, then ONE lane adds
__device__ __forceinline__ void signal_add(int* word) {
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void dag_signal_add(int* word) {
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) dag_add(word, 1);
}

// Dependency counters of the single-launch factorisation, nblk x nblk ints each, per lane (zeroed by k_chol_init):
//   rowdone[k][i]  row blocks of tile (i, k) that have stored their 16 rows of L_ik                     (complete: 4)
//   tver[i][j]     panels applied to the trailing tile (i, j) by the tile-update blocks                  (0 .. j - 1)
//   msdone[r][j]   16-column blocks of the inverse tile M_rj stored                                      (complete: 4)
//   ruver[i][j]    updates applied to the inverse's tile R_ij by the inverse-update tasks  (nblk > 32 only: up to
//                  nblk = 32 the inverse rows accumulate their updates themselves -- minv_strip)   (0 .. i - j - 1)
//   img[k]         1 once the image of L_kk and 1 / diag(L_kk) are in Dfac / dinvG;   ticket: the lane's task counter
struct DagCnt {
    int *rowdone, *tver, *msdone, *ruver, *img, *ticket;
    int nblk;
    __device__ DagCnt(int* base, int nb) : nblk(nb) {
        rowdone = base; tver = base + nb * nb; msdone = base + 2 * nb * nb; ruver = base + 3 * nb * nb; img = base + 4 * nb * nb;
        ticket = img + nb;
    }
    __device__ int* at(int* arr, int a, int b) const { return arr + a * nblk + b; }
};
__host__ __device__ inline int dag_cnt_ints(int nblk) { return 4 * nblk * nblk + nblk + 4; }
__host__ __device__ inline bool dag_ruform(int nblk) { return nblk > 32; }     // see dag_step

template <bool FROM_IMAGE, bool DAG = false>
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
    const int irow = rows ? k + 1 + (b - 1) / 4 : k;      // tile row of this block
    const long r0 = rows ? (long)irow * CB + 16 * ((b - 1) & 3) : 0;
    const DagCnt dc(a.cnt, a.nblk);
    if constexpr (DAG) {
        // the tile this block factorises / solves has received the panels 0 .. k-2 from the tile-update blocks
        if (k >= 2) {
            if (tid == 0) wait_flags(dc.at(dc.tver, irow, k), k - 1, nullptr, 0, nullptr, 0, a.flag);
            __syncthreads();
        }
    }
    // the wave's tiles of S = A_kk - L_k,k-1 L_k,k-1' (lower triangle, 16x16 tiles, MFMA D layout)
    const unsigned tiles = wave_tiles(wv);
    const int nt = int(tiles >> 12), m16 = lane & 15, g4 = lane >> 4;
    // the accumulators start as the tiles of A_kk / the rows of A_ik the panel k-1 update is applied to (see mma64)
    v4d acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    v4d accC = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int ti = (tiles >> (4 * q)) & 3, tj = (tiles >> (4 * q + 2)) & 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double* src = H + (kk + 16 * ti + g4 + 4 * r) * np + kk + 16 * tj + m16;
            acc[q][r] = (!FROM_IMAGE && q < nt) ? (DAG ? ld_sc1(src) : *src) : 0.0;
        }
    }
    if (rows) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double* src = H + (r0 + g4 + 4 * r) * np + kk + 16 * wv + m16;
            accC[r] = DAG ? ld_sc1(src) : *src;
        }
    }
    TRACE(0)
    if (k > 0) {
        const long km = kk - CB;
        if constexpr (DAG) {
            // panel k-1: L_k,k-1 (four row blocks) and, for a row block, its own rows of L_i,k-1 -- the loads above stay
            // in flight behind this poll (the diagonal block's is the one on the chain of the factorisation)
            if (tid == 0) wait_flags(dc.at(dc.rowdone, k - 1, k), 4, rows ? dc.at(dc.rowdone, k - 1, irow) : nullptr, 4, nullptr, 0, a.flag);
            __syncthreads();
        }
        load_block<DAG>(X, H + kk * np + km, np);
        if (rows) {
            double2 t0, t1;
            if constexpr (DAG) {
                const rsrc_t rr = make_rsrc(H + r0 * np + km);
                t0 = ld2_sc1(rr, unsigned(((tid >> 5) * np + 2 * (tid & 31)) * 8));
                t1 = ld2_sc1(rr, unsigned(((8 + (tid >> 5)) * np + 2 * (tid & 31)) * 8));
            } else {
                t0 = *reinterpret_cast<const double2*>(H + (r0 + (tid >> 5)) * np + km + 2 * (tid & 31));
                t1 = *reinterpret_cast<const double2*>(H + (r0 + 8 + (tid >> 5)) * np + km + 2 * (tid & 31));
            }
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
                for (int k0 = 0; k0 < CHOL_KSTEPS(CB); k0 += 4) {
                    const int kx = k0 + g4;
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[16 * ti + m16][kx], X[16 * tj + m16][kx], acc[q], 0, 0, 0);
                }
            }
        }
        if (rows) {
#pragma unroll 4
            for (int k0 = 0; k0 < CHOL_KSTEPS(CB); k0 += 4) {
                const int kx = k0 + g4;
                accC = __builtin_amdgcn_mfma_f64_16x16x4f64(-AR[m16][kx], X[16 * wv + m16][kx], accC, 0, 0, 0);
            }
        }
    }
    TRACE(2)
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
        if (DAG || a.phase == 1) {
            if (tid == 0) wait_flag(DAG ? dc.img + k : a.sync + k, 1, a.flag);
            __syncthreads();
        }
        // Hand-off by write-through stores and L1-bypassing loads (cdna_hip_programming.md guideline 16, R1): every
        // byte of the image is stored sc1 by the diagonal block, each storing wave drains its stores (s_waitcnt
        // vmcnt(0)) before the workgroup barrier behind which ONE lane stores the flag, the flag is polled by ONE lane
        // with sc1 loads, the other waves pass a barrier after the poll, and every load of the image is an sc1 load
        // (served past this CU's L1, which another CU's stores never refresh) -- no L2 write-back / invalidate.
        if constexpr (DAG) {                              // 16-byte sc1 loads (the diagonal block stored it the same way)
            const rsrc_t ri = make_rsrc(a.Dfac + kk * CB);
            double2 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = ld2_sc1(ri, unsigned(2 * (tid + 256 * u) * 8));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = 2 * (tid + 256 * u);
                *reinterpret_cast<double2*>(&Lz[(e >> 6) * ZLD + (e & 63)]) = t[u];
            }
        } else {
            double t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = __hip_atomic_load(a.Dfac + kk * CB + tid + 256 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = tid + 256 * u;
                Lz[(e >> 6) * ZLD + (e & 63)] = t[u];
            }
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
            Y[rr * YLD + cc] = accC[r];
        }
    }
    __syncthreads();
    if (!rows) {
        if constexpr (DAG) {
            const rsrc_t ri = make_rsrc(a.Dfac + kk * CB);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = 2 * (tid + 256 * u);
                dag_st2(ri, unsigned(e * 8), *reinterpret_cast<const double2*>(&Lz[(e >> 6) * ZLD + (e & 63)]));
            }
        } else {
            for (int e = tid; e < CB * CB; e += 256) __hip_atomic_store(a.Dfac + kk * CB + e, Lz[(e >> 6) * ZLD + (e & 63)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < CB) { if constexpr (DAG) dag_st(a.dinvG + kk + tid, dinv[tid]); else __hip_atomic_store(a.dinvG + kk + tid, dinv[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (DAG || a.phase == 1) {                        // release the row blocks (and the inverse-row blocks) of this lane
            drain_stores();                               // every storing wave: its sc1 stores have left the CU ...
            __syncthreads();                              // ... before the one lane that signals for all of them does
            if (tid == 0 && k != g_chol_lose_step) { if constexpr (DAG) dag_set(dc.img + k, 1); else __hip_atomic_store(a.sync + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
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
    if constexpr (DAG) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dag_st(H + (r0 + rho) * np + kk + lam + 16 * i, v[i]);
        dag_signal_add(dc.at(dc.rowdone, k, irow));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) H[(r0 + rho) * np + kk + lam + 16 * i] = v[i];
    }
    TRACE(7)
}

template <bool DAG = false>
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
    const DagCnt dc(a.cnt, a.nblk);
    if constexpr (DAG) {
        // the image of L_rr; L_r,r-1 and the 16 columns of M_r-1,j (all four blocks of that tile: one counter);
        // R_rj with every update it gets (panels j .. r-2)
        if (tid == 0) {
            wait_flags(dc.img + r, 1, j < r ? dc.at(dc.rowdone, r - 1, r) : nullptr, 4, j < r ? dc.at(dc.msdone, r - 1, j) : nullptr, 4, a.flag);
            if (j <= r - 2) wait_flag(dc.at(dc.ruver, r, j), r - j - 1, a.flag);
        }
        __syncthreads();
    }
    {
        double2 t[8];
        if constexpr (DAG) {
            const rsrc_t ri = make_rsrc(a.Dfac + kr * CB);
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = ld2_sc1(ri, unsigned(2 * (tid + 256 * u) * 8));
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const double2*>(a.Dfac + kr * CB + 2 * (tid + 256 * u));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u;
            *reinterpret_cast<double2*>(Lz + (e >> 5) * ZLD + 2 * (e & 31)) = t[u];
        }
    }
    if (tid < CB) dinv[tid] = DAG ? ld_sc1(a.dinvG + kr + tid) : a.dinvG[kr + tid];
    // the accumulator starts as R_rj as the updates left it; tiles no update ever reached hold the previous build's numbers
    // and stand for their initial value: the identity on the diagonal (j == r), zero next to it (j == r - 1)
    v4d acc;
    {
        const int c = lane & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = 16 * wv + (lane >> 4) + 4 * q;
            const double* src = M + (kr + t) * np + (long)j * CB + c0 + c;
            acc[q] = j == r ? (t == c0 + c ? 1.0 : 0.0) : j == r - 1 ? 0.0 : (DAG ? ld_sc1(src) : *src);
        }
    }
    if (j < r) {
        load_block<DAG>(A2, a.H + kr * np + kr - CB, np);
        {
            double t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tid + 256 * u;
                const double* src = M + (kr - CB + (e >> 4)) * np + (long)j * CB + c0 + (e & 15);
                t[u] = DAG ? ld_sc1(src) : *src;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = tid + 256 * u; Bs[e >> 4][e & 15] = t[u]; }
        }
        __syncthreads();
#pragma unroll 4
        for (int k0 = 0; k0 < CHOL_KSTEPS(CB); k0 += 4) {
            const int kx = k0 + (lane >> 4);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-A2[16 * wv + (lane & 15)][kx], Bs[kx][lane & 15], acc, 0, 0, 0);
        }
    }
    __syncthreads();
    {
        const int c = lane & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q) Ct[c * YLD + 16 * wv + (lane >> 4) + 4 * q] = acc[q];
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
    const double2 lo = make_double2(Ct[c4 * YLD + t], Ct[(c4 + 1) * YLD + t]), hi = make_double2(Ct[(c4 + 2) * YLD + t], Ct[(c4 + 3) * YLD + t]);
    if constexpr (DAG) {
        const rsrc_t rm = make_rsrc(M + kr * np + (long)j * CB + c0);
        st2_sc1(rm, unsigned((t * np + c4) * 8), lo);
        st2_sc1(rm, unsigned((t * np + c4 + 2) * 8), hi);
    } else {
        *reinterpret_cast<double2*>(dst) = lo;
        *reinterpret_cast<double2*>(dst + 2) = hi;
    }
    if (a.Mt) {                                           // the transpose for the second triangular GEMV, straight from the staging tile
        const int c = tid >> 4, t4 = (tid & 15) * 4;      // 16 rows of Mt (columns of this block) x 64 entries
        double* dt = a.Mt + ((long)j * CB + c0 + c) * np + kr + t4;
        dt[0] = Ct[c * YLD + t4]; dt[1] = Ct[c * YLD + t4 + 1]; dt[2] = Ct[c * YLD + t4 + 2]; dt[3] = Ct[c * YLD + t4 + 3];
    }
    if constexpr (DAG) signal_add(dc.at(dc.msdone, r, j));
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

// ---- coarse tasks of the single-launch form ------------------------------------------------------------------------
// Measured with one task per 64x64 tile product / 16-row block (tools/exp/chol_dag_exp.hip): a tile update occupied its
// workgroup slot for 8.5-10 us, of which the matrix cores worked 1.5: ticket, polls, tile loads, product, restaging, store
// and drain are serial phases of 1-2 us each, and the chip's 512 slots were full (425 busy on average) -- the
// factorisation was bound by slot-time, not by its chain.  So a task now walks a STRIP of up to four tiles that share
// an operand, with the next tile's loads in flight behind the current product and ONE drain + signal at the end.
#ifndef CHOL_STRIP
#define CHOL_STRIP 4
#endif
constexpr int STRIP = CHOL_STRIP;

// lane t (< n <= 64) of wave 0 polls its own word; everybody passes when all have reached their value
template <class F>
__device__ __forceinline__ void wait_many(int n, F get, int* pivflag) {
    if (threadIdx.x < 64) {
        const int* w = nullptr; int want = 0;
        if (int(threadIdx.x) < n) get(int(threadIdx.x), w, want);
        int spins = 0;
        DAG_WAIT_BEGIN
        for (;;) {
            const int v = w ? __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
            if (__all(v >= want)) break;
            if (++spins >= CHOL_SPIN_LIMIT) { if (threadIdx.x == 0) atomicAdd(pivflag, CHOL_SYNC_LOST); break; }
            poll_pause(spins);
        }
        if (threadIdx.x == 0) { DAG_WAIT_END }
    }
    __syncthreads();
}

// dst_t -= A * B_t (TRANSB: B_t')  for t < cnt, with B_t = B0 + t * bstep, dst_t = D0 + t * dstep and A (64x64 at Ag)
// shared by the strip; tile tfirst (if any) has not been written in this factorisation yet: its old content counts as 0.
// The old tile is read straight into the accumulator layout (8-byte write-through accesses: 16 lanes cover one 128-byte
// line of a row, so every line is still written whole by one wave instruction) and IS the accumulator of the product
// (mma64); the next tile's operand and old content are in flight behind the current product.
template <bool TRANSB>
__device__ __forceinline__ void strip_update(double* smem, const double* __restrict__ Ag, const double* __restrict__ B0, long bstep,
                                             double* __restrict__ D0, long dstep, int tfirst, int cnt, int np) {
    double(*P)[CLD] = reinterpret_cast<double(*)[CLD]>(smem);
    double(*Q)[CLD] = reinterpret_cast<double(*)[CLD]>(smem + CB * CLD);
    const int tid = task_tid();
    load_block<true>(P, Ag, np);
    double2 bt[8];
    v4d nxt[2][2];
    auto fetch = [&](int t) {
        const rsrc_t rb = make_rsrc(B0 + t * bstep);
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = tid + 256 * u; bt[u] = ld2_sc1(rb, unsigned(((e >> 5) * np + 2 * (e & 31)) * 8)); }
        const double* dst = D0 + t * dstep;
        const bool first = t == tfirst;
        acc_foreach(nxt, [&](int i, int j, double& v) {
#ifdef CHOL_EXP_NO_RMW      /* timing experiment only (tools/exp): no read of the old tile */
            v = 0.0;
#else
            v = first ? 0.0 : ld_sc1(dst + (long)i * np + j);
#endif
        });
    };
    fetch(0);
#pragma unroll 1
    for (int t = 0; t < cnt; ++t) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = tid + 256 * u; *reinterpret_cast<double2*>(&Q[e >> 5][2 * (e & 31)]) = bt[u]; }
        v4d acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = nxt[a][b];
        __syncthreads();                                  // P (first pass) and Q in place
        PH(2)
        double* dst = D0 + t * dstep;
        if (t + 1 < cnt) fetch(t + 1);
        mma64<TRANSB>(P, Q, 0, CB, acc);
        PH(3)
#ifndef CHOL_EXP_NO_RMW     /* (the timing experiment does not write either) */
        acc_foreach(acc, [&](int i, int j, double& v) { dag_st(dst + (long)i * np + j, v); });
#endif
        __syncthreads();                                  // everybody is done reading Q: free for the next operand
        PH(4)
    }
}

// Row blocks of ONE tile (i, k), all 64 rows: the updates by the panels p0 .. k-1 (p0 = k-2 in the left-looking form: the
// look-ahead update of a tile off the chain rides here, in the shadow of the diagonal block this task waits for anyway --
// one task, one read-modify-write of the tile and one set of polls less per tile), then two passes of two interleaved 16-row
// substitutions against the image of L_kk.  A wave owns 16 rows and all 64 columns, its rows of L_ip in registers in the matrix
// cores' operand layout (see trail_left2); L_kp goes through LDS.  Same arithmetic per row as panel_block<true>.
__device__ __forceinline__ void row_tile_block(const CholStep& a, int irow, int p0, double* smem) {
    const int tid = task_tid(), lane = tid & 63, wv = tid >> 6, m16 = lane & 15, g4 = lane >> 4;
    const int k = a.k, np = a.np;
    const long kk = (long)k * CB, r0 = (long)irow * CB;
    double(*X)[CLD] = reinterpret_cast<double(*)[CLD]>(smem + R0);     // L_kp, later the staging tile Y
    double* Y = smem + R0;
    double* Lz = smem + R1;
    double* dinv = smem + R3 + CB;
    const DagCnt dc(a.cnt, a.nblk);
    const int npan = k - p0;                              // 0 (k = 0), 1 or 2
    wait_many(1 + 2 * npan, [&](int t, const int*& w, int& want) {
        if (t == 0) { w = k >= 2 ? dc.at(dc.tver, irow, k) : nullptr; want = p0; }
        else { w = dc.at(dc.rowdone, p0 + ((t - 1) >> 1), ((t - 1) & 1) ? irow : k); want = 4; }
    }, a.flag);
    PH(1)
    const rsrc_t rA = make_rsrc(a.H + r0 * np);
    const unsigned voA = unsigned(((16 * wv + m16) * np + g4) * 8), voX = unsigned(((16 * wv + g4) * np + m16) * 8);
    v4d x[4];                                             // the tile itself (accumulator layout): the accumulator of the updates
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[b][r] = ldb_sc1(rA, voX + 128 * b, unsigned((4 * r * np + kk) * 8));
    if (npan > 0) {
        double af[16];
        double2 bt[8];
        auto fetchB = [&](int p) {
            const rsrc_t rb = make_rsrc(a.H + kk * np + (long)p * CB);
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = tid + 256 * u; bt[u] = ld2_sc1(rb, unsigned(((e >> 5) * np + 2 * (e & 31)) * 8)); }
        };
        fetchB(p0);
#pragma unroll
        for (int q = 0; q < 16; ++q) af[q] = ldb_sc1(rA, voA + 32 * q, unsigned(p0 * CB * 8));
#pragma unroll 1
        for (int p = p0; p < k; ++p) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = tid + 256 * u; *reinterpret_cast<double2*>(&X[e >> 5][2 * (e & 31)]) = bt[u]; }
            __syncthreads();
            PH(2)
            const bool more = p + 1 < k;
            if (more) fetchB(p + 1);
            const unsigned pn = unsigned((more ? p + 1 : p) * CB * 8);
            double bf[2][4];
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[0][b] = X[16 * b + m16][g4];
#pragma unroll
            for (int q = 0; q < CHOL_KSTEPS(16); ++q) {
                if (q + 1 < 16) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) bf[(q + 1) & 1][b] = X[16 * b + m16][4 * (q + 1) + g4];
                }
                const double na = -af[q];
#pragma unroll
                for (int b = 0; b < 4; ++b) x[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(na, bf[q & 1][b], x[b], 0, 0, 0);
                if (more) af[q] = ldb_sc1(rA, voA + 32 * q, pn);
            }
            PH(3)
            if (more) __syncthreads();                    // everybody is done reading X
        }
    }
    if (tid == 0) wait_flag(dc.img + k, 1, a.flag);
    __syncthreads();                                      // also: everybody is done with X
    PH(1)
    {                                                     // the image of L_kk
        const rsrc_t ri = make_rsrc(a.Dfac + kk * CB);
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ld2_sc1(ri, unsigned(2 * (tid + 256 * u) * 8));
#pragma unroll
        for (int b = 0; b < 4; ++b)                       // updated rows of A_ik (MFMA layout -> one row per DPP row), all 64
#pragma unroll
            for (int r = 0; r < 4; ++r) Y[(16 * wv + g4 + 4 * r) * YLD + 16 * b + m16] = x[b][r];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = 2 * (tid + 256 * u);
            *reinterpret_cast<double2*>(&Lz[(e >> 6) * ZLD + (e & 63)]) = t[u];
        }
    }
    if (tid < CB) dinv[tid] = ld_sc1(a.dinvG + kk + tid);
    __syncthreads();
    PH(2)
    const int rho = tid >> 4, lam = tid & 15;
    const unsigned voS = unsigned((rho * np + lam) * 8);  // substitution layout: row 16 q + rho, columns lam + 16 i
#pragma unroll 1
    for (int q = 0; q < 4; q += 2) {                      // rows 16 q + rho and 16 (q + 1) + rho: two interleaved substitutions
        double va[4], vb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { va[i] = Y[(16 * q + rho) * YLD + lam + 16 * i]; vb[i] = Y[(16 * q + 16 + rho) * YLD + lam + 16 * i]; }
        subst16x2(Lz, dinv, va, vb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dag_stb(rA, voS + 128 * i, unsigned((16 * q * np + kk) * 8), va[i]);
            dag_stb(rA, voS + 128 * i, unsigned(((16 * q + 16) * np + kk) * 8), vb[i]);
        }
    }
    PH(5)
    drain_stores();
    __syncthreads();
    if (tid == 0) dag_add(dc.at(dc.rowdone, k, irow), 4);
    PH(6)
}

// Trailing tiles (i, j) and -- PAIR -- (i + 1, j), LEFT-LOOKING over the panels p0 .. p1-1:  A_ij -= L_ip L_jp'  one panel after the
// other, accumulated in the task's registers, every tile read once and written once -- the same updates in the same order and
// with the same arithmetic per update as the read-modify-write strips (each product accumulated into the tile's value, see
// mma64): bit-identical results.
// Macro tile (round 5): the two row tiles share the operand L_jp, which goes through LDS once; a wave owns the rows
// 16 w .. 16 w + 15 of BOTH tiles and all 64 columns (2 x 4 MFMA blocks: four LDS reads feed eight MFMAs), and holds its rows of
// L_ip, L_i+1,p in REGISTERS in the matrix cores' operand layout, loaded straight from global memory -- the next panel's
// fragment of a k-step is requested into the same register as soon as the current panel's MFMAs of that k-step are issued (a
// whole panel's product, ~7 us, of cover), and the next L_jp is in flight behind the product as before.  Per product 48 KB of
// operands through the L2 instead of 64, no LDS traffic for the A operand, one barrier pair per TWO products.
// the compiler waits for EVERY outstanding load (s_waitcnt vmcnt(0)) in front of the LDS stores at the head of the panel loop: the
// fragments of the last ALATE k-steps are therefore not re-requested in place at the end of a panel (their latency would be
// exposed right there) but at the head of the next one, 16 - ALATE k-steps before they are used
#ifndef CHOL_ALATE
#define CHOL_ALATE 8
#endif
constexpr int ALATE = CHOL_ALATE;
template <bool PAIR>
__device__ __forceinline__ void trail_left2(const CholStep& a, int i, int j, int p0, int p1, double* smem) {
    const int tid = task_tid(), lane = tid & 63, wv = tid >> 6, m16 = lane & 15, g4 = lane >> 4, np = a.np;
    const DagCnt dc(a.cnt, a.nblk);
    const long i0 = (long)i * CB, j0 = (long)j * CB;
    const int npan = p1 - p0;                             // <= TCHUNK: 2 + 3 npan words to poll
    wait_many(2 + 3 * npan, [&](int t, const int*& w, int& want) {
        if (t < 2) { w = (p0 > 0 && (t == 0 || PAIR)) ? dc.at(dc.tver, i + t, j) : nullptr; want = p0; }
        else {
            const int p = p0 + (t - 2) / 3, q = (t - 2) % 3;
            w = (q == 2 && !PAIR) ? nullptr : dc.at(dc.rowdone, p, q == 0 ? j : i + q - 1);
            want = 4;
        }
    }, a.flag);
    PH(1)
    double(*Q)[CLD] = reinterpret_cast<double(*)[CLD]>(smem + R0);
    const rsrc_t rA = make_rsrc(a.H + i0 * np);                           // tile row i (and i + 1): L_ip, and the tiles (i, j), (i + 1, j)
    const unsigned voA = unsigned(((16 * wv + m16) * np + g4) * 8);        // the wave's rows of L_ip, operand layout: + 64 p (scalar) + 4 q
    const unsigned voX = unsigned(((16 * wv + g4) * np + m16) * 8);        // accumulator layout: + 4 r rows, column j0 (scalar) + 16 b
    const unsigned row1 = unsigned(CB * np * 8);                           // tile row i + 1
    double a0[16], a1[16];
    double2 bt[8];
    auto fetchB = [&](int p) {
        const rsrc_t rb = make_rsrc(a.H + j0 * np + (long)p * CB);
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = tid + 256 * u; bt[u] = ld2_sc1(rb, unsigned(((e >> 5) * np + 2 * (e & 31)) * 8)); }
    };
    // (request order: the tiles, L_jp, the fragments -- as inside the loop, where L_jp of the next panel goes out before the
    //  fragments: the loads outstanding behind L_jp are the same 32 at the head of every trip)
    v4d x0[4], x1[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x0[b][r] = ldb_sc1(rA, voX + 128 * b, unsigned((4 * r * np + j0) * 8));
            x1[b][r] = PAIR ? ldb_sc1(rA, voX + 128 * b, row1 + unsigned((4 * r * np + j0) * 8)) : 0.0;
        }
    fetchB(p0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        a0[q] = ldb_sc1(rA, voA + 32 * q, unsigned(p0 * CB * 8));
        if (PAIR) a1[q] = ldb_sc1(rA, voA + 32 * q, row1 + unsigned(p0 * CB * 8));
    }
#pragma unroll 1
    for (int p = p0; p < p1; ++p) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = tid + 256 * u; *reinterpret_cast<double2*>(&Q[e >> 5][2 * (e & 31)]) = bt[u]; }
        __syncthreads();
        PH(2)
        const bool more = p + 1 < p1;
        // (the last panel requests its own operands again instead of branching around the loads: the k-steps stay one
        // straight line of code, LDS reads one k-step ahead of the MFMAs -- and the loads outstanding at the head of the loop
        // are the same on every trip, so that the wait in front of the LDS stores covers L_jp alone (vmcnt(32)), not the
        // fragments requested during the last k-steps)
        if (p > p0) {
#pragma unroll
            for (int q = 16 - ALATE; q < 16; ++q) {
                a0[q] = ldb_sc1(rA, voA + 32 * q, unsigned(p * CB * 8));
                if (PAIR) a1[q] = ldb_sc1(rA, voA + 32 * q, row1 + unsigned(p * CB * 8));
            }
        }
        fetchB(more ? p + 1 : p);
        const unsigned pn = unsigned((more ? p + 1 : p) * CB * 8);
        double bf[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) bf[0][b] = Q[16 * b + m16][g4];
#pragma unroll
        for (int q = 0; q < CHOL_KSTEPS(16); ++q) {
            if (q + 1 < 16) {
#pragma unroll
                for (int b = 0; b < 4; ++b) bf[(q + 1) & 1][b] = Q[16 * b + m16][4 * (q + 1) + g4];
            }
            const double na0 = -a0[q], na1 = -a1[q];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                x0[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(na0, bf[q & 1][b], x0[b], 0, 0, 0);
                if (PAIR) x1[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(na1, bf[q & 1][b], x1[b], 0, 0, 0);
            }
            if (q < 16 - ALATE) {
                a0[q] = ldb_sc1(rA, voA + 32 * q, pn);
                if (PAIR) a1[q] = ldb_sc1(rA, voA + 32 * q, row1 + pn);
            }
        }
        PH(3)
        __syncthreads();                                  // everybody is done reading Q
        PH(4)
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dag_stb(rA, voX + 128 * b, unsigned((4 * r * np + j0) * 8), x0[b][r]);
            if (PAIR) dag_stb(rA, voX + 128 * b, row1 + unsigned((4 * r * np + j0) * 8), x1[b][r]);
        }
    drain_stores();
    __syncthreads();
    if (tid < (PAIR ? 2 : 1)) dag_add(dc.at(dc.tver, i + tid, j), npan);
    PH(6)
}

// Inverse row r = k-1, tile j, LEFT-LOOKING: the task first accumulates  R_rj = - sum_{p = j}^{r-2} L_rp M_pj  in its
// accumulator registers (operands streamed through LDS, the next pair in flight behind the current product) -- the
// updates the per-step forms apply one panel at a time as read-modify-writes of the inverse's trailing tiles, in the
// same order and with the same arithmetic per update (first update writes the negated product, the others subtract) --
// and then, when row r-1 of the inverse and the image of L_rr are there, does what minv_block does: the product with
// L_r,r-1 (held in REGISTERS in the matrix cores' operand layout), two passes of 32 columns, two interleaved
// substitutions per thread.  No inverse-update tasks, no read-modify-write traffic on the inverse at all.
__device__ __forceinline__ void minv_strip(const CholStep& a, int j, double* smem) {
    const int tid = task_tid(), lane = tid & 63, wv = tid >> 6;
    const int np = a.np, r = a.k - 1;
    const long kr = (long)r * CB;
    double* M = a.M;
    double* Lz = smem + R0;
    double(*Bs)[33] = reinterpret_cast<double(*)[33]>(smem + R1);          // 32 columns of M_r-1,j
    double* Ct = smem + R1 + 64 * 33;                                      // staging [column][row], stride YLD, 32 columns
    double* dinv = smem + R3;
    const DagCnt dc(a.cnt, a.nblk);
    const int c = lane & 15;
    double rold[2][2][4];                                 // R_rj in the layout of the two 32-column passes: [pass][half][q]
    if (j <= r - 2 && dag_ruform(a.nblk)) {
        // (nblk > 32)  R_rj as the inverse-update tasks left it
        if (tid == 0) wait_flag(dc.at(dc.ruver, r, j), r - j - 1, a.flag);
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    rold[ps][h][q] = ld_sc1(M + (kr + 16 * wv + (lane >> 4) + 4 * q) * np + (long)j * CB + 32 * ps + 16 * h + c);
    } else if (j <= r - 2) {
        // ---- accumulate the panels j .. r-2 (their tiles of L and of the inverse were finished steps ago)
        const int npan = r - 1 - j;
        wait_many(min(2 * npan, 64), [&](int t, const int*& w, int& want) {
            const int pq = j + (t >> 1);
            w = (t & 1) ? dc.at(dc.msdone, pq, j) : dc.at(dc.rowdone, pq, r);
            want = 4;
        }, a.flag);
        if (2 * npan > 64 && tid == 0)                    // (more than 32 panels: the rest one by one)
            for (int pq = j + 32; pq <= r - 2; ++pq) { wait_flag(dc.at(dc.rowdone, pq, r), 4, a.flag); wait_flag(dc.at(dc.msdone, pq, j), 4, a.flag); }
        __syncthreads();
        PH(1)
        double(*P)[CLD] = reinterpret_cast<double(*)[CLD]>(smem + R0);
        double(*Q)[CLD] = reinterpret_cast<double(*)[CLD]>(smem + R1);
        double2 at[8], bt[8];
        auto fetch_pair = [&](int pq) {
            const rsrc_t ra = make_rsrc(a.H + kr * np + (long)pq * CB), rb = make_rsrc(M + (long)pq * CB * np + (long)j * CB);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + 256 * u;
                at[u] = ld2_sc1(ra, unsigned(((e >> 5) * np + 2 * (e & 31)) * 8));
                bt[u] = ld2_sc1(rb, unsigned(((e >> 5) * np + 2 * (e & 31)) * 8));
            }
        };
        fetch_pair(j);
        v4d x[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
#pragma unroll 1
        for (int pq = j; pq <= r - 2; ++pq) {
#ifdef CHOL_DAG_STATS      /* phase 7: waiting for the prefetched operand pair; phase 5 (no substitution ran yet): the LDS stores; phase 2: the barrier */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PH(7)
#endif
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + 256 * u;
                *reinterpret_cast<double2*>(&P[e >> 5][2 * (e & 31)]) = at[u];
                *reinterpret_cast<double2*>(&Q[e >> 5][2 * (e & 31)]) = bt[u];
            }
#ifdef CHOL_DAG_STATS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PH(0)
#endif
            __syncthreads();
            PH(2)
            if (pq + 1 <= r - 2) fetch_pair(pq + 1);
            mma64<false>(P, Q, 0, CB, x);                 // (the first update writes the negated product: x starts as zero)
            PH(3)
            __syncthreads();                              // everybody is done reading P and Q
            PH(4)
        }
        // accumulator layout -> the layout of the passes, through LDS
        double(*X)[CLD] = reinterpret_cast<double(*)[CLD]>(smem + R1);
        acc_foreach(x, [&](int i, int jj, double& v) { X[i][jj] = v; });
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q) rold[ps][h][q] = X[16 * wv + (lane >> 4) + 4 * q][32 * ps + 16 * h + c];
        __syncthreads();                                  // R0 / R1 are free for the image and the passes' staging
        PH(4)
    } else {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int t = 16 * wv + (lane >> 4) + 4 * q, cc = 32 * ps + 16 * h + c;
                    rold[ps][h][q] = j == r ? (t == cc ? 1.0 : 0.0) : 0.0;       // identity on the diagonal tile, zero next to it
                }
    }
    // ---- row r-1 of the inverse (this column), L_r,r-1 and the image of L_rr
    if (tid == 0) wait_flags(dc.img + r, 1, j < r ? dc.at(dc.rowdone, r - 1, r) : nullptr, 4, j < r ? dc.at(dc.msdone, r - 1, j) : nullptr, 4, a.flag);
    __syncthreads();
    PH(1)
    {
        const rsrc_t ri = make_rsrc(a.Dfac + kr * CB);
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ld2_sc1(ri, unsigned(2 * (tid + 256 * u) * 8));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u;
            *reinterpret_cast<double2*>(Lz + (e >> 5) * ZLD + 2 * (e & 31)) = t[u];
        }
    }
    if (tid < CB) dinv[tid] = ld_sc1(a.dinvG + kr + tid);
    double af[16];                                        // -L_r,r-1: row 16 wv + (lane & 15), columns 4 q + (lane >> 4) (negated: see mma64)
    if (j < r) {
#pragma unroll
        for (int q = 0; q < 16; ++q) af[q] = -ld_sc1(a.H + (kr + 16 * wv + (lane & 15)) * np + kr - CB + 4 * q + (lane >> 4));
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) af[q] = 0.0;
    }
    // 32 columns of M_r-1,j per pass; the next pass's are in flight behind the current pass's product and substitution
    double bnext[8];
    auto fetch = [&](int pass) {
        const int c0 = 32 * pass;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u;
            bnext[u] = j < r ? ld_sc1(M + (kr - CB + (e >> 5)) * np + (long)j * CB + c0 + (e & 31)) : 0.0;
        }
    };
    fetch(0);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int c0 = 32 * pass;
        v4d acc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[h][q] = rold[pass][h][q];
        if (j < r) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = tid + 256 * u; Bs[e >> 5][e & 31] = bnext[u]; }
        }
        __syncthreads();                                  // Bs (and, first pass, Lz) in place; the previous pass is done with Ct
        PH(2)
        if (j < r) {
#pragma unroll
            for (int q = 0; q < CHOL_KSTEPS(16); ++q) {
                const int kx = 4 * q + (lane >> 4);
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[q], Bs[kx][c], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[q], Bs[kx][16 + c], acc[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = 16 * wv + (lane >> 4) + 4 * q;
                Ct[(16 * h + c) * YLD + t] = acc[h][q];
            }
        if (pass == 0) fetch(1);                          // in flight behind the substitution
        __syncthreads();
        PH(3)
        const int rho = tid >> 4, lam = tid & 15;
        double va[4], vb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { va[i] = Ct[rho * YLD + lam + 16 * i]; vb[i] = Ct[(16 + rho) * YLD + lam + 16 * i]; }
        subst16x2(Lz, dinv, va, vb);
#pragma unroll
        for (int i = 0; i < 4; ++i) { Ct[rho * YLD + lam + 16 * i] = va[i]; Ct[(16 + rho) * YLD + lam + 16 * i] = vb[i]; }
        __syncthreads();
        {
            const int t = tid >> 2, c8 = (tid & 3) * 8;  // row t of the tile, columns c0 + c8 .. + 7
            const rsrc_t rm = make_rsrc(M + kr * np + (long)j * CB + c0);
#pragma unroll
            for (int u = 0; u < 8; u += 2)
                dag_st2(rm, unsigned((t * np + c8 + u) * 8), make_double2(Ct[(c8 + u) * YLD + t], Ct[(c8 + u + 1) * YLD + t]));
        }
        if (a.Mt) {                                       // the transpose for the second triangular GEMV, straight from the staging tile
            const int cr = tid >> 3, t8 = (tid & 7) * 8;  // 32 rows of Mt (columns of this pass) x 64 entries
            double* dt = a.Mt + ((long)j * CB + c0 + cr) * np + kr + t8;
#pragma unroll
            for (int u = 0; u < 8; ++u) dt[u] = Ct[cr * YLD + t8 + u];
        }
        PH(5)
    }
    drain_stores();
    __syncthreads();
    if (tid == 0) dag_add(dc.at(dc.msdone, r, j), 4);
    PH(6)
}


// =================================================================================================
// The whole factorisation (and inverse) in ONE launch: every block of the seventeen k_chol_step launches above becomes
// a task of one grid; the launch boundaries are replaced by the dependency counters of DagCnt.
//  * Forward progress does not rest on the order in which the hardware dispatches workgroups: a workgroup takes its
//    task from its lane's TICKET counter, tickets are numbered in a topological order of the task graph (step by step;
//    inside a step the diagonal block first, the row blocks that wait for it last), and a ticket is only ever drawn by
//    a workgroup that is already running -- so a task only waits for tasks that have started.  Every poll is bounded;
//    a poll that expires raises CHOL_SYNC_LOST in the lane's pivot counter (the host turns it into an error).
//  * Dependencies are per tile, not per step: the diagonal block of step k+1 starts when the four row blocks of
//    L_k+1,k and the tile update of A_k+1,k+1 have signalled (it has loaded A_k+1,k+1 by then), not when everything of
//    step k is done; the lanes of a lock-step batch no longer wait for one another; a tile update of step k+1 starts
//    when its two tiles of panel k are there.
//  * Hand-offs (cdna_hip_programming.md guideline 16, R1): every tile that passes between tasks is stored sc1
//    (written through) in 16-byte pieces, drained by every storing wave before the workgroup barrier behind which one
//    lane adds to the counter; the consumer polls with one lane, passes a barrier and loads with sc1 (L1-bypassing)
//    loads.  No L2 write-back or invalidate anywhere.
//  * Same arithmetic per tile, in the same order, as the multi-launch forms: the results are bit-identical
//    (tests/test_kernels_gpu.py::test_cholesky_split_step_equals_the_fused_step).
// Task order inside step k (per lane): D(k) | the tile updates of block column k+1 (what D(k+1) and the row blocks of
// step k+1 wait for) | the other tile updates | inverse row k-1 (left-looking: its updates included) | row blocks of step k.
struct DagStep { int nD, nLA, nT, nTc, nMS, nRU1, nRU, nRq, nRt, nrem, ruc; };
__host__ __device__ inline int strips_of(int tiles) { return (tiles + STRIP - 1) / STRIP; }
// LEFT-LOOKING trailing updates (round 5; nblk <= 32): a trailing tile (i, j), j >= 3, no longer receives the panels 0 .. j-3 one
// read-modify-write per panel step (the strips below, which remain for nblk > 32), but in ONE task that accumulates them in
// registers and writes the tile once (trail_left) -- drawn in step j - 2, when panel j - 3 is complete; panel j - 2 then comes
// from the look-ahead task of step j - 1 and panel j - 1 inside the row / diagonal block of step j, as before.  A task is
// capped at TCHUNK panels: the columns further right get the chunk of panels [k - TCHUNK, k) in the steps k that are
// multiples of TCHUNK (one more read-modify-write of those tiles), so that no task holds its slot for more than TCHUNK products.
#ifndef CHOL_TCHUNK
#define CHOL_TCHUNK 8
#endif
constexpr int TCHUNK = CHOL_TCHUNK;
__host__ __device__ inline bool dag_leftT(int nblk) { return nblk <= 32; }
// sum over c = 1 .. m of ceil(c / 2): the two-row tasks of columns with 1 .. m tiles
__host__ __device__ inline int pair_tasks(int m) { const int h = m / 2; return (m & 1) ? (h + 1) * (h + 1) : h * (h + 1); }
// ruform: the inverse's trailing updates as tasks of their own (read-modify-write strips, the form of the per-step
// launches) instead of inside the inverse-row tasks -- for nblk > 32, where a left-looking inverse row would hold its
// slot for up to 62 products
__host__ __device__ inline DagStep dag_step(int nblk, int k) {
    const int nrem = nblk - k - 1;
    DagStep s;
    s.nrem = nrem;
    s.nD = k < nblk ? 1 : 0;
    // single tile updates of block column k + 1: all its tiles, or -- left-looking form -- the two on the chain, (k+1, k+1)
    // and (k+2, k+1), alone (the others take panel k - 1 inside their row-block task of step k + 1: row_tile_block)
    s.nLA = (k >= 1 && k < nblk) ? (dag_leftT(nblk) ? min(2, nrem) : nrem) : 0;
    s.nT = 0; s.nTc = 0;
    if (dag_leftT(nblk)) {
        if (k >= 1 && k < nblk && nrem >= 2) {
            // the tiles of column k + 2: their last (or only) chunk, panels .. k - 1.  The two tiles on the chain, (k+2, k+2) and
            // (k+3, k+2), one task each (the look-ahead tasks of the next step wait for them); the others two row tiles per task
            s.nT = min(2, nrem - 1) + (max(nrem - 3, 0) + 1) / 2;
            if (k % TCHUNK == 0 && nrem >= 3) s.nTc = pair_tasks(nrem - 2);            // the full chunk [k - TCHUNK, k) of the columns > k + 2
        }
    } else if (k >= 1 && k < nblk && nrem >= 2) {         // strips over the tiles (i, j), k + 2 <= j <= i, of each row i:
        const int m = nrem - 1, q = m / STRIP, r = m % STRIP;      // sum over c = 1 .. nrem - 1 of ceil(c / STRIP), in closed form
        s.nT = STRIP * q * (q + 1) / 2 + r * (q + 1);
    }
    // inverse rows (two rows per task -- minv_pair, tools/exp/chol_r5_switches.hip -DCHOL_MS_PAIR=1 -- was measured in round 5 and
    // is slower: the row-after-row chain of the inverse gets links twice as long)
    s.nMS = k >= 1 ? k : 0;
                                                          // the k tiles of inverse row k - 1, one task each (its updates by the
                                                          // panels before, then two 32-column passes): row r of the inverse waits
                                                          // for row r - 1, so more tiles per task would be a longer chain (measured:
                                                          // four tiles per task doubled the build)
    const bool ru = dag_ruform(nblk);
    s.ruc = (ru && k >= 2) ? strips_of(k - 1) : 0;
    s.nRU1 = (ru && k >= 2 && k < nblk) ? k - 1 : 0;      // inverse updates of row i = k (the next inverse row waits for them): single tiles
    s.nRU = (ru && k >= 2 && k < nblk) ? (nblk - k - 1) * s.ruc : 0;        // rows i > k: strips
    s.nRq = (k < nblk && nrem >= 1) ? 4 : 0;              // tile (k + 1, k) as four 16-row blocks (on the chain)
    s.nRt = (k < nblk && nrem >= 2) ? nrem - 1 : 0;       // the other tiles of panel k, one block each
    return s;
}
__host__ __device__ inline int dag_step_tasks(const DagStep& s) { return s.nD + s.nLA + s.nT + s.nTc + s.nMS + s.nRU1 + s.nRU + s.nRq + s.nRt; }

#ifdef CHOL_DAG_STATS      /* tools/exp/chol_dag_exp.hip: one record per task of the unit whose H is g_dag_log_H -- kind, begin, end, ticks in polls */
__device__ long long* g_dag_log;
__device__ const double* g_dag_log_H;
__device__ int g_dag_log_tasks;        // tasks per lane
#define DAG_REC 12
#define DAG_STAT_BEGIN const long long t_begin = __builtin_amdgcn_s_memrealtime(); const bool dag_log = a.H == g_dag_log_H; \
        if (threadIdx.x == 0) { s_dag_wait = 0; for (int q = 0; q < 8; ++q) s_ph[q] = 0; s_ph[8] = t_begin; }
#define DAG_STAT_END(kind) if (dag_log && threadIdx.x == 0) { long long* rec = g_dag_log + DAG_REC * ((long)lane * g_dag_log_tasks + s_ticket); \
        rec[0] = (kind); rec[1] = t_begin; rec[2] = __builtin_amdgcn_s_memrealtime(); rec[3] = s_dag_wait; for (int q = 0; q < 8; ++q) rec[4 + q] = s_ph[q]; }
#else
#define DAG_STAT_BEGIN
#define DAG_STAT_END(kind)
#endif

// one task: `a` is the lane's own (pointers shifted), tk its ticket, incl the running task totals of the steps (lane q of the wave:
// steps 0 .. q; see k_chol_dag)
__device__ __forceinline__ void dag_task(CholStep a, const int lane, const int tk, const int incl, double* smem, int& s_ticket) {
    const DagCnt dc(a.cnt, a.nblk);
    DAG_STAT_BEGIN
    int t = tk;
    const int k = __builtin_amdgcn_readfirstlane(__popcll(__ballot(incl <= t)));              // complete steps before the ticket (step 64 of nblk = 64: beyond the lanes)
    if (k > a.nblk || t < 0) { if (threadIdx.x == 0) atomicAdd(a.flag, CHOL_SYNC_LOST); return; }      // (a ticket word somebody else touched)
    if (k > 0) t -= __builtin_amdgcn_readfirstlane(__shfl(incl, k - 1, 64));
    const DagStep st = dag_step(a.nblk, k);
    if (t >= dag_step_tasks(st)) { if (threadIdx.x == 0) atomicAdd(a.flag, CHOL_SYNC_LOST); return; }
    a.k = k;
    const int np = a.np;
    PH(0)
    if (t < st.nD) { panel_block<false, true>(a, 0, smem); DAG_STAT_END(0) return; }
    t -= st.nD;
    if (t < st.nLA) {
        // block column k+1 first (what D(k+1) and the row blocks of step k+1 wait for), one tile per task:
        // A_i,k+1 -= L_i,k-1 L_k+1,k-1'
        const int i = k + 1 + t, j = k + 1;
        if (threadIdx.x == 0)
            wait_flags(dc.at(dc.rowdone, k - 1, i), 4, dc.at(dc.rowdone, k - 1, j), 4, dc.at(dc.tver, i, j), k - 1, a.flag);
        __syncthreads();
        const long i0 = (long)i * CB, j0 = (long)j * CB, km = (long)(k - 1) * CB;
        strip_update<true>(smem, a.H + i0 * np + km, a.H + j0 * np + km, 0, a.H + i0 * np + j0, 0, -1, 1, np);
        dag_signal_add(dc.at(dc.tver, i, j));
        DAG_STAT_END(1)
        return;
    }
    t -= st.nLA;
    // inverse row k - 1 right behind the tiles on the chain: its inputs are all from earlier steps, so these tasks hardly wait -- and
    // they give the row blocks of panel k - 1 (the last tickets of the step before) time to finish before the trailing tiles that
    // need them poll
    if (t < st.nMS) {
        minv_strip(a, t, smem);                           // row k - 1, tile t
        DAG_STAT_END(2)
        return;
    }
    t -= st.nMS;
    if (dag_leftT(a.nblk)) {
        if (t < st.nT + st.nTc) {
            // left-looking trailing tiles: column k + 2 gets its last chunk of panels (.. k - 1), and in the steps that are
            // multiples of TCHUNK the columns further right get the full chunk [k - TCHUNK, k)
            int i, j, p0;
            bool single = false;
            if (t < st.nT) { j = k + 2; single = t < 2; i = t < 2 ? j + t : j + 2 + 2 * (t - 2); p0 = ((k - 1) / TCHUNK) * TCHUNK; }
            else {
                t -= st.nT;
                int c = st.nrem - 2;                      // column j = k + 3 has nrem - 2 tiles, the next one fewer, ...: two per task
                j = k + 3;
                while (t >= (c + 1) / 2) { t -= (c + 1) / 2; --c; ++j; }
                i = j + 2 * t; p0 = k - TCHUNK;
            }
            if (i + 1 < a.nblk && !single) trail_left2<true>(a, i, j, p0, k, smem);
            else trail_left2<false>(a, i, j, p0, k, smem);
            DAG_STAT_END(5)
            return;
        }
        t -= st.nT + st.nTc;
    } else if (t < st.nT) {
        // trailing update with panel k-1, a strip of row i: A_ij -= L_i,k-1 L_j,k-1'  (k + 2 <= j <= i)
        int c = 1;
        while (t >= strips_of(c)) { t -= strips_of(c); ++c; }             // row i = k + 1 + c has c such tiles
        const int i = k + 1 + c, j0 = k + 2 + STRIP * t, cnt = min(STRIP, c - STRIP * t);
        wait_many(1 + 2 * cnt, [&](int q, const int*& w, int& want) {
            if (q == 0) { w = dc.at(dc.rowdone, k - 1, i); want = 4; }
            else if (q & 1) { w = dc.at(dc.rowdone, k - 1, j0 + (q - 1) / 2); want = 4; }
            else { w = dc.at(dc.tver, i, j0 + (q - 2) / 2); want = k - 1; }
        }, a.flag);
        PH(1)
        const long i0 = (long)i * CB, km = (long)(k - 1) * CB;
        strip_update<true>(smem, a.H + i0 * np + km, a.H + (long)j0 * CB * np + km, (long)CB * np, a.H + i0 * np + (long)j0 * CB, CB, -1, cnt, np);
        drain_stores();
        __syncthreads();
        if (int(threadIdx.x) < cnt) dag_add(dc.at(dc.tver, i, j0 + int(threadIdx.x)), 1);
        PH(6)
        DAG_STAT_END(5)
        return;
    } else t -= st.nT;
    if (t < st.nRU1 + st.nRU) {
        // (nblk > 32 only)  R_ij -= L_i,k-2 M_k-2,j  (i >= k, j <= k-2); a tile's first update, by panel j = k - 2, WRITES it.
        // Row i = k tile by tile (inverse row k, one step on, waits for it), the rows below in strips.
        int i, j0, cnt;
        if (t < st.nRU1) { i = k; j0 = t; cnt = 1; }
        else { t -= st.nRU1; i = k + 1 + t / st.ruc; j0 = STRIP * (t % st.ruc); cnt = min(STRIP, k - 1 - j0); }
        const long mm = (long)(k - 2) * CB;
        wait_many(1 + 2 * cnt, [&](int q, const int*& w, int& want) {
            if (q == 0) { w = dc.at(dc.rowdone, k - 2, i); want = 4; }
            else if (q & 1) { w = dc.at(dc.msdone, k - 2, j0 + (q - 1) / 2); want = 4; }
            else { const int j = j0 + (q - 2) / 2; w = dc.at(dc.ruver, i, j); want = k - 2 - j; }
        }, a.flag);
        strip_update<false>(smem, a.H + (long)i * CB * np + mm, a.M + mm * np + (long)j0 * CB, CB, a.M + (long)i * CB * np + (long)j0 * CB, CB,
                            k - 2 - j0, cnt, np);
        drain_stores();
        __syncthreads();
        if (int(threadIdx.x) < cnt) dag_add(dc.at(dc.ruver, i, j0 + int(threadIdx.x)), 1);
        DAG_STAT_END(3)
        return;
    }
    t -= st.nRU1 + st.nRU;
    if (t < st.nRq) { panel_block<true, true>(a, t + 1, smem); DAG_STAT_END(4) return; }       // tile (k+1, k): four 16-row blocks
    t -= st.nRq;
    {
        const int pfirst = dag_leftT(a.nblk) ? max(0, k - 2) : max(0, k - 1);      // (left-looking form: the look-ahead update rides here)
        row_tile_block(a, k + 2 + t, pfirst, smem);       // tiles (i, k), i >= k + 2
    }
    DAG_STAT_END(4)
}

#ifndef CHOL_DAG_WPS
#define CHOL_DAG_WPS 2
#endif
// Grid (lanes, tasks per lane): x runs fastest, so the lanes' workgroups are dealt out alternately.  A workgroup serves ONE ticket of
// the lane of its column.  (Lane ownership by XCD with sweeper workgroups -- round 5, measured and slower -- is in
// tools/exp/chol_r5_switches.hip.)
__global__ __launch_bounds__(256, CHOL_DAG_WPS) void k_chol_dag(CholStep a) {
    __shared__ __attribute__((aligned(16))) double smem[STEP_LDS];
    __shared__ int s_ticket, s_lane;
    // ticket -> (step, task of the step): lane q of every wave counts the tasks of step q, a wave scan gives the running totals --
    // all of it while the ticket's atomic is in flight (a scalar loop over the steps took 1-2.5 us per task)
    const int sl = int(threadIdx.x) & 63;
    int incl = sl <= a.nblk ? dag_step_tasks(dag_step(a.nblk, sl)) : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d, 64); if (sl >= d) incl += u; }
    const int nl = a.nlanes;
    int pref = int(blockIdx.x) < nl ? int(blockIdx.x) : int(blockIdx.x) % nl;
    if (int(blockIdx.x) >= nl) return;
    // one ticket: draw, run the task
    auto serve = [&]() -> bool {
        if (threadIdx.x == 0) {
            int got_lane = -1, got_t = 0;
            // the ticket is drawn whether or not the lane is switched off (a switched-off lane's counter is nobody's: k_chol_init
            // clears it when the lane runs again)
            const DagCnt dl(lane_at(a.cnt, (size_t)pref * a.lane_bytes), a.nblk);
            const int t = dag_add(dl.ticket, 1);
            if (!(a.mask && !a.mask[pref]) && t < a.ntasks) { got_lane = pref; got_t = t; }
            s_lane = got_lane; s_ticket = got_t;
        }
        __syncthreads();
        const int lane = __builtin_amdgcn_readfirstlane(s_lane);
        if (lane < 0) return false;
        const int tk = __builtin_amdgcn_readfirstlane(s_ticket);
        CholStep al = a;
        {
            const size_t off = (size_t)lane * a.lane_bytes;
            al.H = lane_at(a.H, off); al.M = lane_at(a.M, off); al.d0 = lane_at(a.d0, off); al.Dfac = lane_at(a.Dfac, off);
            al.dinvG = lane_at(a.dinvG, off); al.flag = lane_at(a.flag, off); al.cnt = lane_at(a.cnt, off);
            if (a.Mt) al.Mt = lane_at(a.Mt, off);
        }
        dag_task(al, lane, tk, incl, smem, s_ticket);
        __syncthreads();                                  // everybody is done with the task's LDS and with s_lane / s_ticket
        pref = lane;
        return true;
    };
    // (one call, not a loop over tickets: inside a loop the compiler hoists every task kind's per-thread address arithmetic in
    //  front of it and spills what it hoisted -- the persistent-workgroup variant of round 4 measured slower for it)
    if (!serve()) return;
}

// nsync: ints to clear at sync (the panel flags of the split step, or the counters of the single-launch form);
// poison (MBFIR_POISON=1, a test switch): the images of the diagonal blocks and 1 / diag(L) are filled with NaN, so that
// a block which reads them before this build's diagonal block has published them produces NaN instead of plausible
// numbers from the previous build
__global__ __launch_bounds__(256) void k_chol_init(const double* __restrict__ H, int np, double* __restrict__ d0,
                                                   int* __restrict__ flag, int* __restrict__ sync, int nsync, size_t lane_bytes,
                                                   const int* __restrict__ mask, double* __restrict__ poison) {
    if (mask && !mask[blockIdx.y]) return;
    const size_t off = (size_t)blockIdx.y * lane_bytes;
    H = lane_at(H, off); d0 = lane_at(d0, off); flag = lane_at(flag, off); sync = lane_at(sync, off);
    if (blockIdx.x == 0 && threadIdx.x == 0) flag[0] = 0;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < nsync; j += gridDim.x * 256) sync[j] = 0;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < np; j += (long)gridDim.x * 256) d0[j] = H[j * np + j];
    if (poison) {
        poison = lane_at(poison, off);
        for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < (long)(CB + 1) * np; j += (long)gridDim.x * 256) poison[j] = __builtin_nan("");
    }
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

// (declared in dev_common.h for the kernel test hooks of solver.hip)
__global__ void k_extract_L_pub(const double* __restrict__ H, int np, const double* __restrict__ Dfac,
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
    a.cnt = a.sync;                                                     // (single-launch form: dag_cnt_ints(nblk) ints)
    a.Mt = Mt;
    // Lock-step batches split every step (see CholStep::phase): with several designs in flight the chip is no longer
    // empty, and the 4 * nrem row blocks of a step each repeating the 64-pivot factorisation of L_kk is what fills it.
    // MBFIR_CHOL_SPLIT: 4 = the whole factorisation in ONE launch (k_chol_dag; default), 1 = one launch per step with
    // the row blocks waiting for their lane's diagonal block on a flag, 2 = two launches per step, 0 = fused step
    // (every row block factorises L_kk itself; one launch per step).
    // default: the single launch for lock-step batches and, from np = 4096 on, for single designs too (1.9 ms against 2.4 for
    // the split and 2.7 for the fused per-step form); one or two smaller designs keep the fused step, one launch per step
    // (every row block factorises L_kk itself: lowest latency -- 350 against 375 us at np = 1024)
    int split = (nlanes >= 3 || np >= 4096) ? 4 : 0;
    if (const char* ev = std::getenv("MBFIR_CHOL_SPLIT")) split = std::atoi(ev);
    bool poison = false;
    if (const char* ev = std::getenv("MBFIR_POISON")) poison = std::atoi(ev) != 0;
    {   // test hook: lose the hand-off of one panel step (see g_chol_lose_step); the bound shrinks so that the test takes a second.
        // The symbols are per device and the contexts of a batch call this from parallel host threads: the value each device
        // holds is remembered per device, under a mutex.
        static std::mutex lose_mu;
        static int lose_now[64];
        static bool lose_init = false;
        int lose = -1;
        if (const char* ev = std::getenv("MBFIR_TEST_LOSE_FLAG")) lose = std::atoi(ev);
        int dev = 0;
        hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(lose_mu);
        if (!lose_init) { for (int& v : lose_now) v = -1; lose_init = true; }
        if (dev >= 0 && dev < 64 && lose != lose_now[dev]) {
            const int limit = lose >= 0 ? (1 << 14) : CHOL_SPIN_LIMIT_DEFAULT;
            hipMemcpyToSymbolAsync(HIP_SYMBOL(g_chol_lose_step), &lose, sizeof(int), 0, hipMemcpyHostToDevice, st);
            hipMemcpyToSymbolAsync(HIP_SYMBOL(g_chol_spin_limit), &limit, sizeof(int), 0, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
            lose_now[dev] = lose;
        }
    }
    if ((long)dag_cnt_ints(nblk) * 4 > ((long)np * np - (long)np) * 8) split = split == 4 ? (nlanes >= 3 ? 1 : 0) : split;   // (W1 too small: np = 64)
    const int nsync = split == 4 ? dag_cnt_ints(nblk) : nblk + 1;
    hipLaunchKernelGGL(k_chol_init, dim3(cdiv(std::max(np, nsync), 256), nlanes), dim3(256), 0, st, H, np, W1, flag, a.sync, nsync, lane_bytes, mask,
                       poison ? W1 + np : (double*)nullptr);
    if (e0) hipEventRecord(e0, st);
    if (split == 4) {
        int ntasks = 0;
        for (int k = 0; k <= nblk; ++k) ntasks += dag_step_tasks(dag_step(nblk, k));
        a.k = 0; a.phase = 1; a.nP = a.nMS = a.nT = a.nR = 0;
        a.ntasks = ntasks;
        // (x runs fastest: the lanes' workgroups are dealt out alternately)
        hipLaunchKernelGGL(k_chol_dag, dim3(nlanes, ntasks), dim3(256), 0, st, a);
        if (e1) hipEventRecord(e1, st);
        if (Lcopy) hipLaunchKernelGGL(k_extract_L, dim3(cdiv((long)np * np, 256)), dim3(256), 0, st, H, np, a.Dfac, a.dinvG, Lcopy);
        return 1;
    }
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

// x = M'(M b) in ONE pass over the inverse factor:  x = sum_i m_i (m_i . b), m_i = row i of M.  A wave owns a row at a
// time: every lane holds its 2 x U columns of the row (U = np / 128), the dot product is a wave reduction, the row -- still
// in registers -- is added to the lane's slice of x.  Workgroup g of HS_G takes the rows g, g + HS_G, ... (the rows of a
// triangle are unequal: dealt cyclically), reduces its four waves' slices through LDS and writes one partial vector;
// k_hsolve_fold adds the HS_G partials in a fixed order.  M is read ONCE per application (the two triangular GEMVs read
// M and its stored transpose: twice the bytes), and the transpose need not be stored at all.  np <= 1024 (U = ceil(np / 128) <= 8).
constexpr int HS_G = HS_PARTS;
template <int NV, int U>
__global__ __launch_bounds__(256) void k_hsolve(const double* __restrict__ M, int np, const double* __restrict__ b,
                                                const double* __restrict__ b2, double* __restrict__ part, int ldv,
                                                size_t lane_bytes, const int* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) double hs_lds[];      // [4 waves][NV][np]
    if (mask && !mask[blockIdx.y]) return;
    if (blockIdx.y) {
        const size_t off = (size_t)blockIdx.y * lane_bytes;
        M = lane_at(M, off); b = lane_at(b, off); part = lane_at(part, off);
        if (b2) b2 = lane_at(b2, off);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = blockIdx.x;
    double2 bb[NV][U], xa[NV][U];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = 2 * lane + 128 * u;
            bb[v][u] = j < np ? *reinterpret_cast<const double2*>(b + (long)v * ldv + j) : make_double2(0.0, 0.0);
            if (b2 && j < np) { const double2 c = *reinterpret_cast<const double2*>(b2 + (long)v * ldv + j); bb[v][u].x += c.x; bb[v][u].y += c.y; }
            xa[v][u] = make_double2(0.0, 0.0);
        }
    for (int i = g + HS_G * wv; i < np; i += 4 * HS_G) {
        const double* row = M + (long)i * np;
        const int jhi = ((i >> 6) + 1) << 6;              // the row ends with its diagonal tile (exact zeros above the diagonal there)
        double2 m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = 2 * lane + 128 * u;
            m[u] = j < jhi ? *reinterpret_cast<const double2*>(row + j) : make_double2(0.0, 0.0);
        }
        double al[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            double a = 0;
#pragma unroll
            // (explicit fused multiply-adds in a fixed order: left to the compiler's contraction the U = 1 and U = 2 instantiations
            //  rounded a row's products differently -- and a lane of a lock-step unit whose capacitance matrix is padded to the unit's
            //  largest strong set runs another instantiation than its single solve)
            for (int u = 0; u < U; ++u) a = fma(m[u].y, bb[v][u].y, fma(m[u].x, bb[v][u].x, a));
            al[v] = __shfl(wave_sum(a), 0, 64);          // (wave_sum leaves the total in lane 0)
        }
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int u = 0; u < U; ++u) { xa[v][u].x = fma(al[v], m[u].x, xa[v][u].x); xa[v][u].y = fma(al[v], m[u].y, xa[v][u].y); }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int u = 0; u < U; ++u) *reinterpret_cast<double2*>(hs_lds + ((long)wv * NV + v) * (U * 128) + 2 * lane + 128 * u) = xa[v][u];
    __syncthreads();
    for (int e = threadIdx.x; e < NV * np; e += 256) {
        const int v = e / np, j = e - v * np;
        const long w = U * 128;
        const double s = ((hs_lds[(0L * NV + v) * w + j] + hs_lds[(1L * NV + v) * w + j]) + hs_lds[(2L * NV + v) * w + j]) + hs_lds[(3L * NV + v) * w + j];
        part[((long)g * NV + v) * np + j] = s;
    }
}
template <int NV>
__global__ __launch_bounds__(256) void k_hsolve_fold(const double* __restrict__ part, int np, double* __restrict__ out, int ldv,
                                                     size_t lane_bytes, const int* __restrict__ mask) {
    if (mask && !mask[blockIdx.y]) return;
    if (blockIdx.y) { const size_t off = (size_t)blockIdx.y * lane_bytes; part = lane_at(part, off); out = lane_at(out, off); }
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= np) return;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double s = 0;
        for (int g = 0; g < HS_G; ++g) s += part[((long)g * NV + v) * np + j];
        out[(long)v * ldv + j] = s;
    }
}
bool hsolve_fused_ok(int np, int nv) { return np % 64 == 0 && np <= 1024 && (nv == 1 || nv == 2); }
size_t hsolve_part_doubles(int np) { return (size_t)HS_G * 2 * np; }
// out = M'M (b + b2); part: hsolve_part_doubles(np) doubles of scratch per lane
void hsolve_launch(const double* M, int np, const double* b, const double* b2, double* out, double* part, int nv, int ldv,
                   hipStream_t st, int nlanes, size_t lane_bytes, const int* mask, bool fold) {
    if (!hsolve_fused_ok(np, nv)) throw HipError("hsolve_launch: unsupported size");
    const dim3 grid(HS_G, nlanes), gf(cdiv(np, 256), nlanes);
    const int U = (np + 127) / 128;
    const size_t lds = (size_t)4 * nv * U * 128 * sizeof(double);
#define HS_CASE(NVX, UX)                                                                                                   \
    hipLaunchKernelGGL((k_hsolve<NVX, UX>), grid, dim3(256), lds, st, M, np, b, b2, part, ldv, lane_bytes, mask);
    if (nv == 1) {
        switch (U) { case 1: HS_CASE(1, 1) break; case 2: HS_CASE(1, 2) break; case 3: HS_CASE(1, 3) break; case 4: HS_CASE(1, 4) break;
                     case 5: HS_CASE(1, 5) break; case 6: HS_CASE(1, 6) break; case 7: HS_CASE(1, 7) break; default: HS_CASE(1, 8) break; }
        if (fold) hipLaunchKernelGGL(k_hsolve_fold<1>, gf, dim3(256), 0, st, part, np, out, ldv, lane_bytes, mask);
    } else {
        switch (U) { case 1: HS_CASE(2, 1) break; case 2: HS_CASE(2, 2) break; case 3: HS_CASE(2, 3) break; case 4: HS_CASE(2, 4) break;
                     case 5: HS_CASE(2, 5) break; case 6: HS_CASE(2, 6) break; case 7: HS_CASE(2, 7) break; default: HS_CASE(2, 8) break; }
        if (fold) hipLaunchKernelGGL(k_hsolve_fold<2>, gf, dim3(256), 0, st, part, np, out, ldv, lane_bytes, mask);
    }
#undef HS_CASE
    if (lds > 64 * 1024) MBFIR_HIP(hipGetLastError());    // (a launch above 64 KB of LDS is rejected on a device whose attribute was never set)
}
// Dynamic LDS above 64 KB is an attribute of the device function of the CURRENT device: set once per device (Solver's
// constructor, under its mutex), not behind process-wide flags at the launch sites -- those only ever reached the first
// context's device, and the 8-GPU node runs a context per device.
template <int NVX, int UX>
static void hsolve_attr() {
    MBFIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hsolve<NVX, UX>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * NVX * UX * 128 * 8));
}
void chol_warm_kernels() {
    hsolve_attr<1, 1>(); hsolve_attr<1, 2>(); hsolve_attr<1, 3>(); hsolve_attr<1, 4>(); hsolve_attr<1, 5>(); hsolve_attr<1, 6>(); hsolve_attr<1, 7>(); hsolve_attr<1, 8>();
    hsolve_attr<2, 1>(); hsolve_attr<2, 2>(); hsolve_attr<2, 3>(); hsolve_attr<2, 4>(); hsolve_attr<2, 5>(); hsolve_attr<2, 6>(); hsolve_attr<2, 7>(); hsolve_attr<2, 8>();
}

void trigemv_launch(const double* T, int np, int upper, const double* b, double* y, int nv, int ldv,
                    hipStream_t st, const double* b2, int nlanes, size_t lane_bytes, const int* mask) {
    dim3 grid(cdiv(np, 4), nlanes);
    if (nv == 1) hipLaunchKernelGGL(k_trigemv<1>, grid, dim3(256), 0, st, T, np, upper, b, b2, y, ldv, lane_bytes, mask);
    else if (nv == 2) hipLaunchKernelGGL(k_trigemv<2>, grid, dim3(256), 0, st, T, np, upper, b, b2, y, ldv, lane_bytes, mask);
    else throw HipError("trigemv_launch: nv must be 1 or 2");
}

}  // namespace mbfir
