// Experiment (round 5): the lattice moment kernel (K3: G'v) of a lock-step unit whose lanes share the frequency grid, as ONE
// product on the fp64 matrix cores -- the trig values of a (frequency, moment point) pair are generated once per unit instead
// of once per lane -- against a per-lane VALU kernel that performs the same arithmetic in the same order (a lane solved alone
// must stay bit-identical to the same lane inside a unit).
//   part 1: is v_mfma_f64_16x16x4_f64 a chain of four fused multiply-adds in k order?
//   part 2: both kernels on the headline shape (8197 folded frequencies in chunks of 64, 512 moment points, 16 lanes),
//           bitwise comparison, time per launch.
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 trig_mf_exp.hip -o trig_mf_exp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- part 1
__global__ void k_mfma_once(const double* A /*16x4*/, const double* B /*4x16*/, const double* C /*16x16*/, double* D) {
    const int l = threadIdx.x;
    v4d c;
    for (int q = 0; q < 4; ++q) c[q] = C[(l / 16 + 4 * q) * 16 + l % 16];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], c, 0, 0, 0);
    for (int q = 0; q < 4; ++q) D[(l / 16 + 4 * q) * 16 + l % 16] = c[q];
}

// ---------------------------------------------------------------- part 2
struct Rot { double c, s; };
__device__ __forceinline__ Rot rot(Rot p, Rot r) {       // p advanced by the angle of r
    const double t = p.s * r.s, u = p.c * r.s;
    return Rot{fma(p.c, r.c, -t), fma(p.s, r.c, u)};
}
// the four stride-4 chains of a chunk at one moment point: start values (j = 0..3 steps behind the seed) and the 4-step rotation
__device__ __forceinline__ void chains(const double4 sd, Rot (&p)[4], Rot& r4) {
    const Rot r1{sd.z, sd.w}, r2 = rot(r1, r1);
    r4 = rot(r2, r2);
    p[0] = Rot{sd.x, sd.y};
    p[1] = rot(p[0], r1);
    p[2] = rot(p[0], r2);
    p[3] = rot(p[2], r1);
}

constexpr int CHK = 64, LN = 16;

// one wave = 16 moment points x one group of chunks, all 16 lanes of the unit at once.
// OPS[v][k][16 lanes] (pe, po); partial[lane][group][v][kind][LDM]
template <int NV>
__global__ __launch_bounds__(256) void k_mom_mf(const double2* __restrict__ OPS, const double4* __restrict__ seeds, int npts,
                                                const int* __restrict__ ch_start, const int* __restrict__ ch_count, int nchunk,
                                                int cgrp, int Kpad, double* __restrict__ partial, int LDM, long lane_stride, int ngroups) {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, j16 = l & 15, kq = l >> 4;
    const int m = blockIdx.x * 64 + wv * 16 + j16, mc = min(m, npts - 1);
    v4d ac[NV], as[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { ac[v] = (v4d){0, 0, 0, 0}; as[v] = (v4d){0, 0, 0, 0}; }
    constexpr int S = 16 / NV;                // steps per register batch
    for (int cc = 0; cc < cgrp; ++cc) {
        const int ch = blockIdx.y * cgrp + cc;
        if (ch >= nchunk) break;
        const double4 sd = seeds[(long)ch * npts + mc];
        Rot p[4], r4;
        chains(sd, p, r4);
        Rot cur = kq == 0 ? p[0] : kq == 1 ? p[1] : kq == 2 ? p[2] : p[3];
        const int cnt = ch_count[ch], k0 = ch_start[ch];
        for (int q0 = 0; q0 < CHK; q0 += 4 * S) {
            if (q0 >= cnt) break;
            double2 op[S][NV];
#pragma unroll
            for (int st = 0; st < S; ++st)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int q = q0 + 4 * st + kq;
                    op[st][v] = q < cnt ? OPS[((long)v * Kpad + k0 + q) * LN + j16] : make_double2(0.0, 0.0);
                }
#pragma unroll
            for (int st = 0; st < S; ++st) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    ac[v] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[st][v].x, cur.c, ac[v], 0, 0, 0);
                    as[v] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[st][v].y, cur.s, as[v], 0, 0, 0);
                }
                cur = rot(cur, r4);
            }
        }
    }
    if (m >= npts) return;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dl = kq + 4 * r;
            double* o = partial + dl * lane_stride + (((long)blockIdx.y * NV + v) * 2) * LDM + m;
            o[0] = ac[v][r];
            o[LDM] = as[v][r];
        }
}

// the same, software-pipelined: the operands of batch b + 1 (and the seeds of its chunk) are fetched while batch b multiplies
template <int NV>
struct Batch { double2 op[16 / NV][NV]; double4 sd; };
template <int NV>
__device__ __forceinline__ void fetch_batch(Batch<NV>& B, int it, int nb, const double2* __restrict__ OPS, const double4* __restrict__ seeds, int npts,
                                            const int* __restrict__ ch_start, const int* __restrict__ ch_count, int nchunk, int ch0, int Kpad,
                                            int mc, int j16, int kq) {
    constexpr int S = 16 / NV;
    const int ch = ch0 + it / NV;                      // NV batches per chunk
    const bool live = it < nb && ch < nchunk;
    const int chc = live ? ch : 0;
    const int cnt = live ? ch_count[chc] : 0, k0 = ch_start[chc], q0 = (it % NV) * 4 * S;
    B.sd = seeds[(long)chc * npts + mc];
#pragma unroll
    for (int st = 0; st < S; ++st)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int q = q0 + 4 * st + kq;
            B.op[st][v] = q < cnt ? OPS[((long)v * Kpad + k0 + q) * LN + j16] : make_double2(0.0, 0.0);
        }
}
template <int NV>
__device__ __forceinline__ void mult_batch(const Batch<NV>& B, int it, Rot& cur, Rot& r4, int kq, v4d (&ac)[NV], v4d (&as)[NV]) {
    constexpr int S = 16 / NV;
    if (it % NV == 0) {                                // a new chunk: its chains
        Rot p[4];
        chains(B.sd, p, r4);
        cur = kq == 0 ? p[0] : kq == 1 ? p[1] : kq == 2 ? p[2] : p[3];
    }
#pragma unroll
    for (int st = 0; st < S; ++st) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            ac[v] = __builtin_amdgcn_mfma_f64_16x16x4f64(B.op[st][v].x, cur.c, ac[v], 0, 0, 0);
            as[v] = __builtin_amdgcn_mfma_f64_16x16x4f64(B.op[st][v].y, cur.s, as[v], 0, 0, 0);
        }
        cur = rot(cur, r4);
    }
}
template <int NV>
__global__ __launch_bounds__(256) void k_mom_mf2(const double2* __restrict__ OPS, const double4* __restrict__ seeds, int npts,
                                                 const int* __restrict__ ch_start, const int* __restrict__ ch_count, int nchunk,
                                                 int cgrp, int Kpad, double* __restrict__ partial, int LDM, long lane_stride, int ngroups) {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, j16 = l & 15, kq = l >> 4;
    const int m = blockIdx.x * 64 + wv * 16 + j16, mc = min(m, npts - 1);
    v4d ac[NV], as[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { ac[v] = (v4d){0, 0, 0, 0}; as[v] = (v4d){0, 0, 0, 0}; }
    const int ch0 = blockIdx.y * cgrp, nb = min(cgrp, nchunk - ch0) * NV;
    Batch<NV> A, B;
    Rot cur{0, 0}, r4{0, 0};
    fetch_batch<NV>(A, 0, nb, OPS, seeds, npts, ch_start, ch_count, nchunk, ch0, Kpad, mc, j16, kq);
    for (int it = 0; it < nb; it += 2) {
        fetch_batch<NV>(B, it + 1, nb, OPS, seeds, npts, ch_start, ch_count, nchunk, ch0, Kpad, mc, j16, kq);
        mult_batch<NV>(A, it, cur, r4, kq, ac, as);
        fetch_batch<NV>(A, it + 2, nb, OPS, seeds, npts, ch_start, ch_count, nchunk, ch0, Kpad, mc, j16, kq);
        if (it + 1 < nb) mult_batch<NV>(B, it + 1, cur, r4, kq, ac, as);
    }
    if (m >= npts) return;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dl = kq + 4 * r;
            double* o = partial + dl * lane_stride + (((long)blockIdx.y * NV + v) * 2) * LDM + m;
            o[0] = ac[v][r];
            o[LDM] = as[v][r];
        }
}

// the per-lane kernel with the same arithmetic: one thread per moment point, four chains, the products of a step in k order
template <int NV>
__global__ __launch_bounds__(256) void k_mom_lane(const double2* __restrict__ PP /*[lane][v][Kpad]*/, const double4* __restrict__ seeds, int npts,
                                                  const int* __restrict__ ch_start, const int* __restrict__ ch_count, int nchunk,
                                                  int cgrp, int Kpad, double* __restrict__ partial, int LDM, long lane_stride, int CGRPMAX) {
    __shared__ double2 pp[NV][4][CHK];
    const double2* src = PP + (long)blockIdx.z * NV * Kpad;
    partial += blockIdx.z * lane_stride;
    const int tid = threadIdx.x, ch0 = blockIdx.y * cgrp;
    for (int e = tid; e < cgrp * CHK; e += 256) {
        const int cc = e / CHK, q = e - cc * CHK, ch = ch0 + cc;
        const bool live = ch < nchunk && q < ch_count[ch < nchunk ? ch : 0];
        const int k = live ? ch_start[ch] + q : 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) pp[v][cc][q] = live ? src[(long)v * Kpad + k] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    const int m = blockIdx.x * 256 + tid;
    if (m >= npts) return;
    double ag[NV], as[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) ag[v] = as[v] = 0;
    double4 sdv[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) sdv[cc] = seeds[(long)min(ch0 + min(cc, cgrp - 1), nchunk - 1) * npts + m];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int ch = ch0 + cc;
        if (cc >= cgrp || ch >= nchunk) break;
        Rot p[4], r4;
        chains(sdv[cc], p, r4);
        const int cnt = ch_count[ch];
#pragma unroll 2
        for (int q = 0; q < cnt; q += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const double2 o = pp[v][cc][q + j];
                    ag[v] = fma(o.x, p[j].c, ag[v]);
                    as[v] = fma(o.y, p[j].s, as[v]);
                }
                p[j] = rot(p[j], r4);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        partial[(((long)blockIdx.y * NV + v) * 2) * LDM + m] = ag[v];
        partial[(((long)blockIdx.y * NV + v) * 2 + 1) * LDM + m] = as[v];
    }
}

// today's kernel (solver.hip k_trig_moments), for the time only
template <int NV>
__global__ __launch_bounds__(256) void k_mom_old(const double2* __restrict__ PP, const double4* __restrict__ seeds, int npts,
                                                 const int* __restrict__ ch_start, const int* __restrict__ ch_count, int nchunk,
                                                 int cgrp, int Kpad, double* __restrict__ partial, int LDM, long lane_stride) {
    __shared__ double2 pp[NV][4][CHK];
    const double2* src = PP + (long)blockIdx.z * NV * Kpad;
    partial += blockIdx.z * lane_stride;
    const int tid = threadIdx.x, ch0 = blockIdx.y * cgrp;
    for (int e = tid; e < cgrp * CHK; e += 256) {
        const int cc = e / CHK, q = e - cc * CHK, ch = ch0 + cc;
        const bool live = ch < nchunk && q < ch_count[ch < nchunk ? ch : 0];
        const int k = live ? ch_start[ch] + q : 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) pp[v][cc][q] = live ? src[(long)v * Kpad + k] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    const int m = blockIdx.x * 256 + tid;
    if (m >= npts) return;
    double ag[NV], as[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) ag[v] = as[v] = 0;
    for (int cl = 0; cl < cgrp; cl += 2) {
        const int cha = ch0 + cl, chb = cha + 1;
        if (cha >= nchunk) break;
        const bool two = cl + 1 < cgrp && chb < nchunk;
        const double4 sa = seeds[(long)cha * npts + m];
        const double4 sb = two ? seeds[(long)chb * npts + m] : make_double4(0.0, 0.0, 0.0, 0.0);
        double c0 = sa.x, s0 = sa.y, c1 = sb.x, s1 = sb.y;
        const int cnt = max(ch_count[cha], two ? ch_count[chb] : 0);
        const int clb = two ? cl + 1 : cl;
#pragma unroll 4
        for (int q = 0; q < cnt; ++q) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double2 pa = pp[v][cl][q], pb = pp[v][clb][q];
                ag[v] += pa.x * c0; as[v] += pa.y * s0;
                ag[v] += pb.x * c1; as[v] += pb.y * s1;
            }
            const double n0 = c0 * sa.z - s0 * sa.w, n1 = c1 * sb.z - s1 * sb.w;
            s0 = s0 * sa.z + c0 * sa.w; s1 = s1 * sb.z + c1 * sb.w;
            c0 = n0; c1 = n1;
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        partial[(((long)blockIdx.y * NV + v) * 2) * LDM + m] = ag[v];
        partial[(((long)blockIdx.y * NV + v) * 2 + 1) * LDM + m] = as[v];
    }
}

template <int NV>
static void part2(int nfold, int npts, int cgrp, int reps) {
    const int nchunk = (nfold + CHK - 1) / CHK, Kpad = nchunk * CHK, ngroups = (nchunk + cgrp - 1) / cgrp, LDM = (npts + 63) / 64 * 64;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<int> cs(nchunk), cn(nchunk);
    for (int c = 0; c < nchunk; ++c) { cs[c] = c * CHK; cn[c] = std::min(CHK, nfold - c * CHK); }
    cn[nchunk / 2] = 37;                                          // a ragged chunk in the middle (a band's end)
    std::vector<double4> sd((size_t)nchunk * npts);
    for (int c = 0; c < nchunk; ++c) {
        const double w0 = 3.1 * U(rng), dw = 2e-4 * (1 + U(rng));
        for (int m = 0; m < npts; ++m) {
            const double t = -255.5 + m;
            sd[(size_t)c * npts + m] = make_double4(cos(w0 * t), sin(w0 * t), cos(dw * t), sin(dw * t));
        }
    }
    std::vector<double2> PP((size_t)LN * NV * Kpad), OPS((size_t)NV * Kpad * LN);
    for (int ln = 0; ln < LN; ++ln)
        for (int v = 0; v < NV; ++v)
            for (int k = 0; k < Kpad; ++k) {
                const double2 x = make_double2(U(rng), U(rng));
                PP[((size_t)ln * NV + v) * Kpad + k] = x;
                OPS[((size_t)v * Kpad + k) * LN + ln] = x;
            }
    const long lane_stride = (long)ngroups * NV * 2 * LDM;
    double2 *dPP, *dOPS; double4* dsd; int *dcs, *dcn; double *pa, *pb, *pc;
    CK(hipMalloc(&dPP, PP.size() * 16)); CK(hipMalloc(&dOPS, OPS.size() * 16)); CK(hipMalloc(&dsd, sd.size() * 32));
    CK(hipMalloc(&dcs, nchunk * 4)); CK(hipMalloc(&dcn, nchunk * 4));
    CK(hipMalloc(&pa, LN * lane_stride * 8)); CK(hipMalloc(&pb, LN * lane_stride * 8)); CK(hipMalloc(&pc, LN * lane_stride * 8));
    CK(hipMemcpy(dPP, PP.data(), PP.size() * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dOPS, OPS.data(), OPS.size() * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsd, sd.data(), sd.size() * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcs, cs.data(), nchunk * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcn, cn.data(), nchunk * 4, hipMemcpyHostToDevice));
    CK(hipMemset(pa, 0xff, LN * lane_stride * 8)); CK(hipMemset(pb, 0xff, LN * lane_stride * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto&& f, const char* name) {
        f(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) f();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  NV %d  %-28s %8.2f us per launch\n", NV, name, 1e3 * ms / reps);
    };
    const dim3 gmf((npts + 63) / 64, ngroups), gl((npts + 255) / 256, ngroups, LN);
    timeit([&] { hipLaunchKernelGGL(k_mom_mf<NV>, gmf, dim3(256), 0, 0, dOPS, dsd, npts, dcs, dcn, nchunk, cgrp, Kpad, pa, LDM, lane_stride, ngroups); }, "matrix cores, whole unit");
    timeit([&] { hipLaunchKernelGGL(k_mom_mf2<NV>, gmf, dim3(256), 0, 0, dOPS, dsd, npts, dcs, dcn, nchunk, cgrp, Kpad, pa, LDM, lane_stride, ngroups); }, "matrix cores, pipelined");
    timeit([&] { hipLaunchKernelGGL(k_mom_lane<NV>, gl, dim3(256), 0, 0, dPP, dsd, npts, dcs, dcn, nchunk, cgrp, Kpad, pb, LDM, lane_stride, 4); }, "per lane, same arithmetic");
    timeit([&] { hipLaunchKernelGGL(k_mom_old<NV>, gl, dim3(256), 0, 0, dPP, dsd, npts, dcs, dcn, nchunk, cgrp, Kpad, pc, LDM, lane_stride); }, "per lane, today's kernel");
    const dim3 g1((npts + 255) / 256, ngroups, 1);
    timeit([&] { hipLaunchKernelGGL(k_mom_lane<NV>, g1, dim3(256), 0, 0, dPP, dsd, npts, dcs, dcn, nchunk, cgrp, Kpad, pb, LDM, lane_stride, 4); }, "ONE lane, same arithmetic");
    timeit([&] { hipLaunchKernelGGL(k_mom_old<NV>, g1, dim3(256), 0, 0, dPP, dsd, npts, dcs, dcn, nchunk, cgrp, Kpad, pc, LDM, lane_stride); }, "ONE lane, today's kernel");
    hipLaunchKernelGGL(k_mom_lane<NV>, gl, dim3(256), 0, 0, dPP, dsd, npts, dcs, dcn, nchunk, cgrp, Kpad, pb, LDM, lane_stride, 4);
    CK(hipDeviceSynchronize());
    std::vector<double> ha(LN * lane_stride), hb(LN * lane_stride), hc(LN * lane_stride);
    CK(hipMemcpy(ha.data(), pa, ha.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), pb, hb.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), pc, hc.size() * 8, hipMemcpyDeviceToHost));
    long diff = 0, cnt = 0; double maxrel = 0, maxold = 0;
    for (int ln = 0; ln < LN; ++ln)
        for (int g = 0; g < ngroups; ++g)
            for (int v = 0; v < 2 * NV; ++v)
                for (int m = 0; m < npts; ++m) {
                    const size_t o = ln * lane_stride + ((size_t)g * 2 * NV + v) * LDM + m;
                    ++cnt;
                    if (memcmp(&ha[o], &hb[o], 8)) { ++diff; maxrel = std::max(maxrel, fabs(ha[o] - hb[o])); }
                    maxold = std::max(maxold, fabs(hc[o] - hb[o]));
                }
    printf("  NV %d  matrix cores against per lane: %ld of %ld values differ (max abs %.3e); today's kernel differs by at most %.3e\n", NV, diff, cnt, maxrel, maxold);
    hipFree(dPP); hipFree(dOPS); hipFree(dsd); hipFree(dcs); hipFree(dcn); hipFree(pa); hipFree(pb); hipFree(pc);
}

int main(int argc, char** argv) {
    {   // part 1
        std::mt19937_64 rng(3);
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        double A[64], B[64], C[256], D[256];
        int fwd = 0, rev = 0, trials = 200;
        double *dA, *dB, *dC, *dD;
        CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 2048)); CK(hipMalloc(&dD, 2048));
        for (int t = 0; t < trials; ++t) {
            for (double& x : A) x = U(rng) * std::ldexp(1.0, int(20 * U(rng)));
            for (double& x : B) x = U(rng);
            for (double& x : C) x = U(rng);
            CK(hipMemcpy(dA, A, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, C, 2048, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_mfma_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
            CK(hipMemcpy(D, dD, 2048, hipMemcpyDeviceToHost));
            bool f = true, r = true;
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double x = C[i * 16 + j], y = C[i * 16 + j];
                    for (int k = 0; k < 4; ++k) x = std::fma(A[i * 4 + k], B[k * 16 + j], x);
                    for (int k = 3; k >= 0; --k) y = std::fma(A[i * 4 + k], B[k * 16 + j], y);
                    if (memcmp(&x, &D[i * 16 + j], 8)) f = false;
                    if (memcmp(&y, &D[i * 16 + j], 8)) r = false;
                }
            fwd += f; rev += r;
        }
        printf("v_mfma_f64_16x16x4: %d of %d random products equal the fused chain k = 0,1,2,3 bit for bit; %d the chain k = 3,2,1,0\n", fwd, trials, rev);
    }
    const int reps = argc > 1 ? atoi(argv[1]) : 50;
    for (int cgrp : {4, 2}) {
        printf("headline shape: 8197 folded frequencies, 512 moment points, 16 lanes, groups of %d chunks\n", cgrp);
        part2<1>(8197, 512, cgrp, reps);
        part2<2>(8197, 512, cgrp, reps);
    }
    printf("normal-matrix moments: 1535 points, groups of 4\n");
    part2<2>(8197, 1535, 4, reps);
    return 0;
}
