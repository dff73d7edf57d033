// K2: scaled Gram  T_w = A' diag(d_w) A  on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// A is the materialised frequency matrix (Mpad x ld, row-major: one row per frequency, so the
// reduction index -- the frequency -- is the slow index and both MFMA operands are read as
// 128-byte contiguous segments).  Work decomposition:
//   * output tiles of 128 x 128, lower triangle only (T is symmetric);
//   * split-K over the frequency axis (blockIdx.y) so that a (2n-1)^2 output still fills 256 CUs;
//     partial tiles go to a slab and a second kernel adds them in a fixed order (deterministic);
//   * per workgroup: 4 waves in a 2 x 2 arrangement, each owning a 64 x 64 sub-tile = 4 x 4 MFMA
//     blocks (128 accumulator VGPRs per weight vector);
//   * 16-row K chunks staged through LDS (register double-buffered, one barrier per chunk);
//     LDS rows padded to 144 doubles so the 4 k-slices of one ds_read_b64 hit disjoint banks.
// Algorithmic work: Mf * Nt * (Nt + 1) flop per weight vector (lower triangle).
#include "dev_common.h"
#include <vector>

namespace mbfir {

constexpr int GT = 128;
constexpr int GKB = 16;
constexpr int GLDP = 144;

template <int NW>
__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ A, int ld,
                                              const double* __restrict__ d, int Mpad, int chunks,
                                              const int* __restrict__ tile_ij, const int* __restrict__ pair_tab, int ntiles,
                                              double* __restrict__ slab) {
    __shared__ double As[2][GKB][GLDP];
    __shared__ double Bs[2][GKB][GLDP];
    __shared__ double Ds[2][NW][GKB];
    // (tile, split) of this workgroup from the XCD-aware table behind the tile list (gram_tiles_host): workgroups are
    // dealt round-robin over the 8 XCDs, and the table gives every XCD a compact 2 x 2 cluster of tiles with all their
    // K slices, so the column panels a cluster shares are fetched once per XCD (its L2) instead of once per workgroup
    // (pair_tab: the table of this launch -- the whole product's, or one chunk's: gram_chunk_tables)
    const int* pair = pair_tab + 2 * blockIdx.x;
    const int t = pair[0], split = pair[1];
    if (t < 0) return;
    const int ti = tile_ij[2 * t], tj = tile_ij[2 * t + 1];
    const bool diag = ti == tj;
    const int I0 = ti * GT, J0 = tj * GT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wi = wv >> 1, wj = wv & 1;
    const long k0 = (long)split * chunks * GKB;

    v4d acc[NW][4][4];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[w][a][b] = (v4d){0, 0, 0, 0};

    // Staging registers for the next chunk (explicit scalars + macros: lambdas capturing the arrays
    // by reference kept them in scratch memory -- 144 B/lane, 1 GB of spill traffic per launch).
    double2 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    double rd = 0;
    const int lrow = tid >> 6, lc2 = tid & 63;          // this thread loads rows lrow + 4 q, columns 2 lc2..+1
#define GRAM_GLOAD(c)                                                                              \
    {                                                                                              \
        const double* p_ = A + (k0 + (long)(c) * GKB + lrow) * ld + 2 * lc2;                       \
        ra0 = *reinterpret_cast<const double2*>(p_ + I0);                                          \
        ra1 = *reinterpret_cast<const double2*>(p_ + 4L * ld + I0);                                \
        ra2 = *reinterpret_cast<const double2*>(p_ + 8L * ld + I0);                                \
        ra3 = *reinterpret_cast<const double2*>(p_ + 12L * ld + I0);                               \
        if (!diag) {                                                                               \
            rb0 = *reinterpret_cast<const double2*>(p_ + J0);                                      \
            rb1 = *reinterpret_cast<const double2*>(p_ + 4L * ld + J0);                            \
            rb2 = *reinterpret_cast<const double2*>(p_ + 8L * ld + J0);                            \
            rb3 = *reinterpret_cast<const double2*>(p_ + 12L * ld + J0);                           \
        }                                                                                          \
        if (tid < NW * GKB) rd = d[(long)(tid / GKB) * Mpad + k0 + (long)(c) * GKB + (tid % GKB)]; \
    }
#define GRAM_SSTORE(buf)                                                                           \
    {                                                                                              \
        *reinterpret_cast<double2*>(&As[buf][lrow][2 * lc2]) = ra0;                                \
        *reinterpret_cast<double2*>(&As[buf][lrow + 4][2 * lc2]) = ra1;                            \
        *reinterpret_cast<double2*>(&As[buf][lrow + 8][2 * lc2]) = ra2;                            \
        *reinterpret_cast<double2*>(&As[buf][lrow + 12][2 * lc2]) = ra3;                           \
        if (!diag) {                                                                               \
            *reinterpret_cast<double2*>(&Bs[buf][lrow][2 * lc2]) = rb0;                            \
            *reinterpret_cast<double2*>(&Bs[buf][lrow + 4][2 * lc2]) = rb1;                        \
            *reinterpret_cast<double2*>(&Bs[buf][lrow + 8][2 * lc2]) = rb2;                        \
            *reinterpret_cast<double2*>(&Bs[buf][lrow + 12][2 * lc2]) = rb3;                       \
        }                                                                                          \
        if (tid < NW * GKB) Ds[buf][tid / GKB][tid % GKB] = rd;                                    \
    }

    GRAM_GLOAD(0)
    GRAM_SSTORE(0)
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < chunks) GRAM_GLOAD(c + 1)
        const double(*Bsrc)[GLDP] = diag ? As[buf] : Bs[buf];
#pragma unroll
        for (int kk = 0; kk < GKB / 4; ++kk) {
            const int krow = kk * 4 + (lane >> 4);
            double af[4], bf[4], dv[NW];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = As[buf][krow][wi * 64 + a * 16 + (lane & 15)];
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = Bsrc[krow][wj * 64 + b * 16 + (lane & 15)];
#pragma unroll
            for (int w = 0; w < NW; ++w) dv[w] = Ds[buf][w][krow];
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double bw = bf[b] * dv[w];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[w][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bw, acc[w][a][b], 0, 0, 0);
                }
        }
        if (c + 1 < chunks) GRAM_SSTORE(buf ^ 1)
        __syncthreads();
    }
#undef GRAM_GLOAD
#undef GRAM_SSTORE
    // D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg.
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        double* out = slab + (((long)split * ntiles + t) * NW + w) * (GT * GT);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int i = wi * 64 + a * 16 + (lane >> 4) + 4 * r;
                    int j = wj * 64 + b * 16 + (lane & 15);
                    out[i * GT + j] = acc[w][a][b][r];
                }
    }
}

// Sum the split-K partial tiles in a fixed order and write the full symmetric T.
__global__ __launch_bounds__(256) void k_gram_reduce(const double* __restrict__ slab, int nsplit,
                                                     int ntiles, int nw,
                                                     const int* __restrict__ tile_ij, int ld,
                                                     double* __restrict__ T) {
    // grid (ntiles, 16): block y folds 8 rows of the tile
    const int t = blockIdx.x, w = 0;
    const int ti = tile_ij[2 * t], tj = tile_ij[2 * t + 1];
    const bool diag = ti == tj;
    double* Tw = T + (long)w * ld * ld;
    const int e0 = blockIdx.y * (GT * GT / 16);
    for (int e = e0 + threadIdx.x; e < e0 + GT * GT / 16; e += 256) {
        int i = e >> 7, j = e & 127;
        if (diag && j > i) continue;
        double v = 0;
        for (int s = 0; s < nsplit; ++s) v += slab[(((long)s * ntiles + t) * nw + w) * (GT * GT) + e];
        long gi = (long)ti * GT + i, gj = (long)tj * GT + j;
        Tw[gi * ld + gj] = v;
        Tw[gj * ld + gi] = v;
    }
}

GramPlan gram_plan(int Mf, int Nt, int nw, int nlaunch) {
    GramPlan gp;
    gp.nw = nw;
    gp.ld = int(round_up(Nt, GT));
    gp.ntile = gp.ld / GT;
    gp.ntiles = gp.ntile * (gp.ntile + 1) / 2;
    int total_chunks = cdiv(Mf, GKB);
    // two workgroups fit on a CU (74 KB LDS, 196 VGPRs): 512 slots on 256 CUs.  Pick the split so
    // that ntiles * nsplit fills ONE round of slots -- one workgroup too many doubles the run time.
    // nlaunch > 1: the tile list goes in that many launches (gram_chunk_tables), each of which should fill the slots by itself.
    const int per_launch = cdiv(gp.ntiles, std::max(1, nlaunch));
    int want = 512 / per_launch;
    if (nlaunch > 1 || per_launch > 256) {
        // More tiles than half the slots (n > 1400 taps), or a chunk of tiles: one round cannot be filled with whole tiles, so take
        // the split whose workgroups fill their LAST round best (BASELINE config 5, 528 tiles: one slice each = 528 workgroups =
        // two rounds, the second with 16 of 512 slots busy: 52.4 ms; 7 slices: 3696 = 7.2 rounds: 41 ms).  At least 8 chunks of
        // 16 frequencies per slice, at most 16 slices (the partial tiles are 128 KB each).
        double best = -1;
        for (int ns = 1; ns <= 16 && ns * 8 <= std::max(8, total_chunks); ++ns) {
            const long blocks = (long)per_launch * ns;
            const double eff = double(blocks) / double(512L * cdiv(blocks, 512));
            if (eff > best + 1e-9) { best = eff; want = ns; }
        }
    }
    gp.nsplit = std::max(1, std::min(want, total_chunks));
    gp.chunks = cdiv(total_chunks, gp.nsplit);
    gp.nsplit = cdiv(total_chunks, gp.chunks);
    gp.Mpad = gp.nsplit * gp.chunks * GKB;
    gp.slab_doubles = (size_t)gp.nsplit * gp.ntiles * nw * GT * GT;
    return gp;
}

int gram_grid_blocks(const GramPlan& gp) { return 8 * cdiv((long)gp.ntiles * gp.nsplit, 8); }
int gram_table_ints(const GramPlan& gp) { return 2 * gp.ntiles + 2 * gram_grid_blocks(gp); }

// tile list (ti, tj per tile) followed by the (tile, split) pair of every workgroup of the k_gram grid.
// Workgroup b runs on XCD b % 8 (observed dispatch; only speed depends on it): XCD c gets the c-th eighth of the pair
// list, which walks the tiles in 2 x 2 clusters (a cluster's four tiles touch four column panels instead of eight)
// with all K slices of a tile together.
// the tiles (index in the (i, j <= i) row-major list) in the order the workgroups walk them: 2 x 2 clusters
static std::vector<int> gram_tile_order(const GramPlan& gp) {
    std::vector<int> order;
    for (int bi = 0; 2 * bi < gp.ntile; ++bi)
        for (int bj = 0; bj <= bi; ++bj)
            for (int di = 0; di < 2; ++di)
                for (int dj = 0; dj < 2; ++dj) {
                    const int i = 2 * bi + di, j = 2 * bj + dj;
                    if (i < gp.ntile && j <= i) order.push_back(i * (i + 1) / 2 + j);
                }
    return order;
}
// pair table of the tiles order[lo .. hi): 8 * ceil((hi - lo) nsplit / 8) workgroups, dealt over the XCDs as described above
static void gram_deal_pairs(const GramPlan& gp, const std::vector<int>& order, int lo, int hi, int* pair) {
    const int nblocks = 8 * cdiv((long)(hi - lo) * gp.nsplit, 8), seg = nblocks / 8;
    for (int b = 0; b < nblocks; ++b) { pair[2 * b] = -1; pair[2 * b + 1] = 0; }
    long q = 0;
    for (int p = lo; p < hi; ++p)
        for (int s = 0; s < gp.nsplit; ++s, ++q) {
            const int c = int(q / seg), pos = int(q % seg);          // XCD c, its pos-th workgroup
            pair[2 * (8 * pos + c)] = order[p]; pair[2 * (8 * pos + c) + 1] = s;
        }
}
void gram_tiles_host(const GramPlan& gp, int* tile_ij) {
    int t = 0;
    for (int i = 0; i < gp.ntile; ++i)
        for (int j = 0; j <= i; ++j) { tile_ij[2 * t] = i; tile_ij[2 * t + 1] = j; ++t; }
    gram_deal_pairs(gp, gram_tile_order(gp), 0, gp.ntiles, tile_ij + 2 * gp.ntiles);
}

// ---- the product in CHUNKS of tiles (dense row-sharded builds: chunk c's all-reduce runs while chunk c + 1 is computed) ------------
// table = [order: ntiles ints][pair table of chunk 0][pair table of chunk 1] ...; a tile's place in the packed output is its
// position in `order`, so a chunk's tiles are one contiguous stretch of it
void gram_chunk_tables(const GramPlan& gp, int nchunks, std::vector<int>& table, std::vector<GramChunk>& chunks) {
    const std::vector<int> order = gram_tile_order(gp);
    nchunks = std::max(1, std::min(nchunks, gp.ntiles));
    table.assign(order.begin(), order.end());
    chunks.clear();
    for (int c = 0; c < nchunks; ++c) {
        GramChunk ck;
        ck.plo = int((long)gp.ntiles * c / nchunks); ck.phi = int((long)gp.ntiles * (c + 1) / nchunks);
        ck.blocks = 8 * cdiv((long)(ck.phi - ck.plo) * gp.nsplit, 8);
        ck.pair_off = int(table.size());
        table.resize(table.size() + 2 * (size_t)ck.blocks);
        gram_deal_pairs(gp, order, ck.plo, ck.phi, table.data() + ck.pair_off);
        chunks.push_back(ck);
    }
}
// fold of the split-K partials of the tiles order[0 .. gridDim.x) into the packed output (whole 128 x 128 tiles, diagonal ones too)
__global__ __launch_bounds__(256) void k_gram_reduce_packed(const double* __restrict__ slab, int nsplit, int ntiles,
                                                            const int* __restrict__ order, double* __restrict__ Tp) {
    const int t = order[blockIdx.x];
    const int e0 = blockIdx.y * (GT * GT / 16);
    for (int e = e0 + threadIdx.x; e < e0 + GT * GT / 16; e += 256) {
        double v = 0;
        for (int s = 0; s < nsplit; ++s) v += slab[((long)s * ntiles + t) * (GT * GT) + e];
        Tp[(long)blockIdx.x * (GT * GT) + e] = v;
    }
}
// the full symmetric T from the packed tiles
__global__ __launch_bounds__(256) void k_gram_unpack(const double* __restrict__ Tp, const int* __restrict__ order,
                                                     const int* __restrict__ tile_ij, int ld, double* __restrict__ T) {
    const int t = order[blockIdx.x];
    const int ti = tile_ij[2 * t], tj = tile_ij[2 * t + 1];
    const bool diag = ti == tj;
    const int e0 = blockIdx.y * (GT * GT / 16);
    for (int e = e0 + threadIdx.x; e < e0 + GT * GT / 16; e += 256) {
        const int i = e >> 7, j = e & 127;
        if (diag && j > i) continue;
        const double v = Tp[(long)blockIdx.x * (GT * GT) + e];
        const long gi = (long)ti * GT + i, gj = (long)tj * GT + j;
        T[gi * ld + gj] = v;
        T[gj * ld + gi] = v;
    }
}
void gram_chunk_launch(const GramPlan& gp, const GramChunk& ck, const double* A, const double* d, double* slab,
                       const int* tile_ij, const int* chunk_table, double* Tp, hipStream_t st) {
    if (gp.nw != 1) throw HipError("gram_chunk_launch: one weight vector");
    hipLaunchKernelGGL(k_gram<1>, dim3(ck.blocks), dim3(256), 0, st, A, gp.ld, d, gp.Mpad, gp.chunks, tile_ij, chunk_table + ck.pair_off,
                       gp.ntiles, slab);
    hipLaunchKernelGGL(k_gram_reduce_packed, dim3(ck.phi - ck.plo, 16), dim3(256), 0, st, slab, gp.nsplit, gp.ntiles, chunk_table + ck.plo,
                       Tp + (size_t)ck.plo * GT * GT);
}
void gram_unpack_launch(const GramPlan& gp, const double* Tp, const int* tile_ij, const int* chunk_table, double* T, hipStream_t st) {
    hipLaunchKernelGGL(k_gram_unpack, dim3(gp.ntiles, 16), dim3(256), 0, st, Tp, chunk_table, tile_ij, gp.ld, T);
}

void gram_launch(const GramPlan& gp, const double* A, const double* d, double* slab, double* T,
                 const int* tile_ij, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, size_t d_stride) {
    if (!d_stride) d_stride = (size_t)gp.Mpad;
    // One k_gram launch per weight vector: three accumulator sets (384 VGPRs) would spill, and the
    // kernel is MFMA-bound, so re-reading A from L2/MALL costs nothing measurable.  The optional
    // events bracket the k_gram launches only (roofline timing), the split-K fold comes after.
    dim3 grid(gram_grid_blocks(gp));
    const size_t per_w = (size_t)gp.nsplit * gp.ntiles * GT * GT;
    if (ev0) hipEventRecord(ev0, st);
    for (int w = 0; w < gp.nw; ++w)
        hipLaunchKernelGGL(k_gram<1>, grid, dim3(256), 0, st, A, gp.ld, d + (size_t)w * d_stride, gp.Mpad,
                           gp.chunks, tile_ij, tile_ij + 2 * gp.ntiles, gp.ntiles, slab + w * per_w);
    if (ev1) hipEventRecord(ev1, st);
    for (int w = 0; w < gp.nw; ++w)
        hipLaunchKernelGGL(k_gram_reduce, dim3(gp.ntiles, 16), dim3(256), 0, st, slab + w * per_w, gp.nsplit,
                           gp.ntiles, 1, tile_ij, gp.ld, T + (size_t)w * gp.ld * gp.ld);
}

}  // namespace mbfir
