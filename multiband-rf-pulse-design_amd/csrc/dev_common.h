// Shared device/host helpers for the gfx950 kernels (wave64, fp64).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace mbfir {

struct HipError : std::runtime_error {
    explicit HipError(const std::string& s) : std::runtime_error(s) {}
};

#define MBFIR_HIP(expr)                                                                        \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            throw mbfir::HipError(std::string(#expr) + ": " + hipGetErrorString(e__) + " at " + \
                                  __FILE__ + ":" + std::to_string(__LINE__));                  \
    } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
// Block-wide sum; result valid in every thread.  sh must hold >= 17 doubles.
__device__ __forceinline__ double block_sum(double v, double* sh) {
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int i = 0; i < nw; ++i) t += sh[i];
        sh[16] = t;
    }
    __syncthreads();
    return sh[16];
}
__device__ __forceinline__ double block_max(double v, double* sh) {
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
        for (int i = 1; i < nw; ++i) t = fmax(t, sh[i]);
        sh[16] = t;
    }
    __syncthreads();
    return sh[16];
}

// Drain of a wave's outstanding stores, in front of the barrier behind which one lane hands data over to ANOTHER workgroup of the same
// launch (chol.hip k_chol_dag, ddlin.hip k_dd_trsv_mw, solver.hip gt_resid_tail: write-through stores, then a counter / flag).  A
// workgroup-scope release fence emits no such wait (found as a run-to-run difference in 1 of 720 fuzz jobs in round 5).  vmcnt counts
// stores on the gfx9 family only -- gfx10 and later count them in vscnt -- so this file refuses to build the wait for anything else
// (ADVICE r5).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "drain_stores(): s_waitcnt vmcnt(0) drains stores on gfx942 / gfx950 only"
#endif
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

inline int cdiv(long a, long b) { return int((a + b - 1) / b); }
inline long round_up(long a, long b) { return ((a + b - 1) / b) * b; }

// ---- kernels implemented in gram.hip / chol.hip / specfact.hip --------------------------------
struct GramPlan {
    int ld = 0;        // padded column count of A1 (multiple of 128)
    int ntile = 0;     // ld / 128
    int ntiles = 0;    // lower-triangular tile count
    int nsplit = 0;    // split-K factor
    int chunks = 0;    // K chunks (16 rows each) per split
    int Mpad = 0;      // padded row count (multiple of 16*nsplit)
    int nw = 1;        // number of weight vectors
    size_t slab_doubles = 0;
};
GramPlan gram_plan(int Mf, int Nt, int nw, int nlaunch = 1);   // nlaunch: the product goes in that many launches (gram_chunk_tables), each sized to fill the chip
// T[w] (ld x ld, full symmetric) = A' diag(d[w]) A ; A is Mpad x ld row-major, d is nw x Mpad.
// d_stride: doubles between the weight vectors of d (0: gp.Mpad; a lane of a heterogeneous unit keeps the unit's stride)
void gram_launch(const GramPlan& gp, const double* A, const double* d, double* slab, double* T,
                 const int* tile_ij, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, size_t d_stride = 0);
// The same product in chunks of tiles, each chunk's tiles folded into a PACKED buffer (tile p of the walk order at Tp + p * 128 * 128, whole
// tiles): what the dense row-sharded build all-reduces chunk by chunk on a second stream while the next chunk is computed.
struct GramChunk { int plo = 0, phi = 0, blocks = 0, pair_off = 0; };   // tiles order[plo .. phi), workgroups, offset of its pair table
void gram_chunk_tables(const GramPlan& gp, int nchunks, std::vector<int>& table, std::vector<GramChunk>& chunks);
void gram_chunk_launch(const GramPlan& gp, const GramChunk& ck, const double* A, const double* d, double* slab,
                       const int* tile_ij, const int* chunk_table, double* Tp, hipStream_t st);
void gram_unpack_launch(const GramPlan& gp, const double* Tp, const int* tile_ij, const int* chunk_table, double* T, hipStream_t st);
int gram_grid_blocks(const GramPlan& gp);                 // workgroups of one k_gram launch
int gram_table_ints(const GramPlan& gp);                  // length of the table gram_tiles_host fills
void gram_tiles_host(const GramPlan& gp, int* tile_ij);   // tile list + the XCD-aware (tile, split) pair of every workgroup

// Cholesky + inverse of the Cholesky factor.  H is np x np row-major (np multiple of 64), lower
// triangle referenced; on exit M = L^-1 (lower), Mt = M'.
// W1 is a workspace of 66*np doubles.  flag[0] counts replaced (noise-level) pivots.
// e0 / e1 (optional) are recorded right before / after the np/64 + 1 k_chol_step launches.
// Lock-step batch: nlanes designs, the buffers of lane b at + b * lane_bytes, mask (nlanes ints or null) = lanes to do.
// Returns the number of k_chol_step launches it issued (one per panel step, two for lock-step batches).
int chol_inv_launch(double* H, double* M, double* Mt, double* W1, int np, int* flag, hipStream_t st,
                    double* Lcopy = nullptr, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, int nlanes = 1,
                    size_t lane_bytes = 0, const int* mask = nullptr);   // on exit H is scratch; Lcopy (optional) receives L

// L (np x np, clean lower triangle) from a factored H, the diagonal-block images (W1 + np) and 1 / diag(L) (W1 + 65 np)
__global__ void k_extract_L_pub(const double* __restrict__ H, int np, const double* __restrict__ Dfac,
                                const double* __restrict__ dinvG, double* __restrict__ Lout);

// y[v] = Lo * (b[v] + b2[v]) for a row-major lower (upper=0) or upper (upper=1) triangular np x np
// matrix; b2 may be null.
void trigemv_launch(const double* T, int np, int upper, const double* b, double* y, int nv, int ldv,
                    hipStream_t st, const double* b2 = nullptr, int nlanes = 1, size_t lane_bytes = 0, const int* mask = nullptr);

// out = M'M (b + b2) in one pass over M (np % 128 == 0, np <= 1024, nv 1 or 2: hsolve_fused_ok); part: scratch of
// hsolve_part_doubles(np) doubles per lane
bool hsolve_fused_ok(int np, int nv);
size_t hsolve_part_doubles(int np);
// fold = false (round 5): the HS_PARTS partial vectors stay in `part` ([g][v][np]) for a consumer that adds them itself (solver.hip
// k_fold_cg_start: one launch instead of k_hsolve_fold + k_cg_start)
constexpr int HS_PARTS = 32;
void hsolve_launch(const double* M, int np, const double* b, const double* b2, double* out, double* part, int nv, int ldv,
                   hipStream_t st, int nlanes = 1, size_t lane_bytes = 0, const int* mask = nullptr, bool fold = true);

// Double-double dense kernels (ddlin.hip): H(dd) = Hh + sum_{r < *kcount} X[r] U[r] U[r]' on the lower-triangle
// tiles; in-place dd Cholesky (L in the lower triangle of (Hh, Hl), L' in (Lth, Ltl), 1/diag(L) in (rih, ril),
// flag[0] = replaced pivots, d0 = np doubles of work space); (L L')^-1 B for nv = 1 or 2 dd right-hand sides.
void dd_syrk_launch(const double* U, int ldu, const double* X, const int* kcount, int np, double* Hh, double* Hl,
                    hipStream_t st);
void dd_chol_launch(double* Hh, double* Hl, double* Lth, double* Ltl, double* rih, double* ril, double* d0, int np,
                    double pivtol, int* flag, hipStream_t st, double* dinv = nullptr);     // dinv: 2 np x 64 doubles, the inverses of the 64 x 64 diagonal blocks of L
// per-device kernel preparation (code objects resolved, dynamic-LDS attributes set on the CURRENT device): called by the
// Solver's constructor once per device id, under its mutex
void dd_warm_kernels();      // ddlin.hip
void chol_warm_kernels();    // chol.hip
void dd_trsv_launch(const double* Lh, const double* Ll, const double* Lth, const double* Ltl, const double* rih,
                    const double* ril, int np, double* Bh, double* Bl, int nv, int ldv, hipStream_t st, int* flags = nullptr, int epoch = 0,
                    int* lost = nullptr, const double* dinv = nullptr);
// Capacitance form of the extended-precision solve (capkkt.hip): Yt = U M', Zt = Yt M, S = Yt Yt' + X^-1 on the fp64 matrix cores
// (U: kp x np with zero rows from k on; M: inverse Cholesky factor of the capped normal matrix, lower triangle; S: kp x kp,
// lower tiles, unit diagonal on the padding rows), and the vector kernels of its solves
// (nlanes, lane_bytes, mask: lock-step units; kcnt != null: the lane's own k is read from its arena, `k` and `kp` are the unit's largest)
void cap_build_launch(const double* U, int k, int kp, int np, const double* M, const double* X, double* Yt, double* Zt, double* S, double* part,
                      hipStream_t st, int nlanes = 1, size_t lane_bytes = 0, const int* mask = nullptr, const int* kcnt = nullptr);
size_t cap_part_doubles(int kmax, int np);                // the split-K slab `part` of cap_build_launch
void cap_add_launch(const double* a, const double* b, double* out, int n, int np, int ldv, int nv, hipStream_t st,
                    int nlanes = 1, size_t lane_bytes = 0, const int* mask = nullptr);
void cap_uy_launch(const double* U, int k, int kp, int n, int np, const double* y, int ldv, const double* t, double* w, int ldk, int nv, hipStream_t st,
                   int nlanes = 1, size_t lane_bytes = 0, const int* mask = nullptr, const int* kcnt = nullptr);
void cap_dx_launch(const double* Zt, int k, int n, int np, const double* zeta, int ldk, const double* y, double* dx, int ldv, int nv, hipStream_t st,
                   int nlanes = 1, size_t lane_bytes = 0, const int* mask = nullptr, const int* kcnt = nullptr);
void cap_flag_add_launch(int* flag, const int* more, hipStream_t st, int nlanes = 1, size_t lane_bytes = 0, const int* mask = nullptr);
// In-launch hand-offs between workgroups (chol.hip, ddlin.hip) poll with a bound; a poll that expires adds this to the
// pivot-replacement counter of the factorisation it belongs to, and the host turns a counter at or above it into an error.
constexpr int CHOL_SYNC_LOST = 1 << 20;
constexpr int DD_SYNC_LOST = CHOL_SYNC_LOST;

// Spectral factorisation (fir_ap_cvx.m:185-186,264-304): x (2n-1) -> n taps (re, im interleaved
// in hout[2n]).  work must hold 6*lp doubles, lp = 8*2^ceil(log2(2n-1)).
int specfact_lp(int n);
void specfact_launch(const double* x, int n, double* work, double* hout, hipStream_t st, int nlanes = 1, size_t lane_bytes = 0);

// Inverse SLR (slr.hip; b2a.m:15-32, ab2rf.m:14-29).  b2a: work holds 48 n doubles; a_il / b_il / rf_il are
// interleaved (re, im) device arrays of 2 n doubles.  ab2rf: n <= 2048.
void slr_b2a_launch(const double* b_re, const double* b_im, int n, double* work, double* a_il, hipStream_t st);
void slr_abr_launch(const double* rf_il, const double* g, int n, const double* x, int nx, int mode, double* a_il, double* b_il,
                    hipStream_t st);
// Bloch simulation with relaxation (slr.hip k_bloch; blochC.c:283-512).  step: ntime x 8 per-sample quantities.
void bloch_launch(const double* step, int ntime, const double* df, int nf, const double* pos3, int npos, int mode, double* mx,
                  double* my, double* mz, hipStream_t st);
void slr_ab2rf_launch(const double* a_il, const double* b_il, int n, double* rf_il, hipStream_t st);

}  // namespace mbfir
