// Single-launch factorisation (k_chol_dag) against the one-launch-per-step forms: bit-identical factor / inverse /
// transpose / images on every lane, alone and with several units in flight on separate streams (uneven load: the
// streams start staggered and run different lane counts), and the time per factorisation of each form.
//   hipcc -O3 --offload-arch=gfx950 chol_dag_exp.hip -o chol_dag_exp ;  ./chol_dag_exp [np=1024] [lanes=8] [streams=4] [reps=10]
#define CHOL_DAG_STATS 1
#include "../../multiband-rf-pulse-design_amd/csrc/chol.hip"
#include <vector>
#include <cmath>
#include <cstring>
#include <chrono>
using namespace mbfir;

struct Unit {
    int np, nl; size_t lane_doubles, lane_bytes;
    double* base = nullptr; int* df = nullptr; hipStream_t st = nullptr;
    double *dH, *dM, *dMt, *dW;
    void alloc(int np_, int nl_) {
        np = np_; nl = nl_;
        lane_doubles = 4 * (size_t)np * np + 80 * (size_t)np + 64; lane_bytes = lane_doubles * 8;
        hipMalloc(&base, lane_bytes * nl);
        hipMemset(base, 0, lane_bytes * nl);
        dH = base; dM = base + (size_t)np * np; dMt = dM + (size_t)np * np; dW = dMt + (size_t)np * np;
        df = reinterpret_cast<int*>(dW + (size_t)np * np + 70 * (size_t)np);
        hipStreamCreate(&st);
    }
    void upload(const std::vector<double>& H, double shift) {
        std::vector<double> Hb(H);
        for (int b = 0; b < nl; ++b) {
            for (int i = 0; i < np; ++i) Hb[i * (size_t)np + i] = H[i * (size_t)np + i] + shift + 0.125 * b;
            hipMemcpyAsync(reinterpret_cast<char*>(dH) + b * lane_bytes, Hb.data(), np * (size_t)np * 8, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
        }
    }
    int run(hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) { return chol_inv_launch(dH, dM, dMt, dW, np, df, st, nullptr, e0, e1, nl, lane_bytes, nullptr); }
    std::vector<double> snapshot() {
        hipStreamSynchronize(st);
        std::vector<double> out(lane_doubles * nl);
        hipMemcpy(out.data(), base, lane_bytes * nl, hipMemcpyDeviceToHost);
        return out;
    }
};

// compare what the solver reads afterwards: L (lower tiles of H below the diagonal blocks), images, 1/diag, M lower, Mt upper
static long diff(const Unit& u, const std::vector<double>& a, const std::vector<double>& b) {
    long bad = 0;
    const int np = u.np;
    for (int l = 0; l < u.nl; ++l) {
        const double* A = a.data() + l * u.lane_doubles; const double* B = b.data() + l * u.lane_doubles;
        const size_t oM = (size_t)np * np, oMt = 2 * oM, oW = 3 * oM;
        for (int i = 0; i < np; ++i)
            for (int j = 0; j <= i; ++j) {
                if (i / 64 != j / 64 && std::memcmp(&A[i * (size_t)np + j], &B[i * (size_t)np + j], 8)) ++bad;          // L_ik tiles
                if (std::memcmp(&A[oM + i * (size_t)np + j], &B[oM + i * (size_t)np + j], 8)) ++bad;                    // M
                if (std::memcmp(&A[oMt + j * (size_t)np + i], &B[oMt + j * (size_t)np + i], 8)) ++bad;                  // Mt
            }
        for (size_t e = np; e < (size_t)66 * np; ++e) if (std::memcmp(&A[oW + e], &B[oW + e], 8)) ++bad;                // images, 1/diag
        int fa, fb; std::memcpy(&fa, &A[oW + (size_t)np * np + 70 * (size_t)np], 4); std::memcpy(&fb, &B[oW + (size_t)np * np + 70 * (size_t)np], 4);
        if (fa != fb || fa != 0) { printf("lane %d: pivot counters %d %d\n", l, fa, fb); ++bad; }
    }
    return bad;
}

static long long* d_log = nullptr;
static int log_tasks = 0, log_lanes = 0;
static void dag_log_on(const Unit& u) {
    int nt = 0;
    for (int k = 0; k <= u.np / 64; ++k) nt += dag_step_tasks(dag_step(u.np / 64, k));
    log_tasks = nt; log_lanes = u.nl;
    if (!d_log) hipMalloc(&d_log, sizeof(long long) * DAG_REC * (size_t)nt * 64);
    hipMemset(d_log, 0, sizeof(long long) * DAG_REC * (size_t)nt * u.nl);
    const double* hp = u.dH;
    hipMemcpyToSymbol(HIP_SYMBOL(g_dag_log), &d_log, sizeof(d_log)); hipMemcpyToSymbol(HIP_SYMBOL(g_dag_log_H), &hp, sizeof(hp));
    hipMemcpyToSymbol(HIP_SYMBOL(g_dag_log_tasks), &nt, sizeof(nt));
}
static void dag_stats(const char* what) {
    hipDeviceSynchronize();
    std::vector<long long> L(DAG_REC * (size_t)log_tasks * log_lanes);
    hipMemcpy(L.data(), d_log, L.size() * 8, hipMemcpyDeviceToHost);
    const int NK = 6;
    const char* names[NK] = {"D (diagonal block)", "LA (look-ahead tile)", "MS (inverse row)", "RU (inverse update)", "R (row block)", "T (strip)"};
    double cnt[NK] = {0}, dur[NK] = {0}, wt[NK] = {0}, ph[NK][8] = {{0}};
    long long t0 = 0, t1 = 0;
    for (size_t q = 0; q < (size_t)log_tasks * log_lanes; ++q) {
        const long long* r = &L[DAG_REC * q];
        if (!r[1]) continue;
        cnt[r[0]] += 1; dur[r[0]] += double(r[2] - r[1]); wt[r[0]] += double(r[3]);
        for (int f = 0; f < 8; ++f) ph[r[0]][f] += double(r[4 + f]);
        if (!t0 || r[1] < t0) t0 = r[1];
        if (r[2] > t1) t1 = r[2];
    }
    double tot = 0, totw = 0;
    for (int q = 0; q < NK; ++q) { tot += dur[q]; totw += wt[q]; }
    printf("task statistics of one build of one unit, %s: first start to last end %.1f us; workgroup-time %.1f ms = %.1f slots busy on average; %.1f %% of it in polls\n",
           what, 0.01 * double(t1 - t0), 1e-5 * tot, tot / double(t1 - t0), 100.0 * totw / tot);
    for (int q = 0; q < NK; ++q) {
        if (!cnt[q]) continue;
        printf("  %-22s %6.0f tasks  %6.2f us each, of which %5.2f us in polls   %5.1f %% of the workgroup-time", names[q], cnt[q], cnt[q] ? 0.01 * dur[q] / cnt[q] : 0.0,
               cnt[q] ? 0.01 * wt[q] / cnt[q] : 0.0, 100.0 * dur[q] / tot);
        printf("   phases (us per task) ticket %.2f poll %.2f landed %.2f products %.2f epilogue %.2f subst %.2f drain %.2f other %.2f\n", 0.01 * ph[q][0] / cnt[q], 0.01 * ph[q][1] / cnt[q],
               0.01 * ph[q][2] / cnt[q], 0.01 * ph[q][3] / cnt[q], 0.01 * ph[q][4] / cnt[q], 0.01 * ph[q][5] / cnt[q], 0.01 * ph[q][6] / cnt[q], 0.01 * ph[q][7] / cnt[q]);
    }
    // the chain: start-to-start of the diagonal blocks of lane 0
    printf("  lane 0 diagonal blocks (start, duration, in polls; us):");
    long long prev = 0;
    int tk = 0;
    for (int k = 0; k < log_tasks && tk < 17; ++k) {
        const long long* r = &L[DAG_REC * (size_t)k];
        if (r[1] && r[0] == 0) { printf(" [%.1f %.1f %.1f]", prev ? 0.01 * double(r[1] - prev) : 0.0, 0.01 * double(r[2] - r[1]), 0.01 * double(r[3])); prev = r[1]; ++tk; }
    }
    printf("\n");
    const double* hp = nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(g_dag_log_H), &hp, sizeof(hp));
}

int main(int argc, char** argv) {
    const int np = argc > 1 ? atoi(argv[1]) : 1024, nl = argc > 2 ? atoi(argv[2]) : 8, ns = argc > 3 ? atoi(argv[3]) : 4, reps = argc > 4 ? atoi(argv[4]) : 10;
    std::vector<double> H(np * (size_t)np);
    for (int i = 0; i < np; ++i) for (int j = 0; j < np; ++j) H[i * (size_t)np + j] = 1.0 / (1.0 + abs(i - j)) + (i == j ? 2.0 : 0.0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<Unit> U(ns);
    for (int s = 0; s < ns; ++s) U[s].alloc(np, s == 0 ? nl : std::max(1, nl - (s % 3)));      // uneven lane counts
    // reference: split step, one launch per panel step (the previous default)
    std::vector<std::vector<double>> ref(ns);
    for (int s = 0; s < ns; ++s) {
        setenv("MBFIR_CHOL_SPLIT", U[s].nl >= 3 ? "1" : "0", 1);
        U[s].upload(H, 1.0); U[s].run();              // a different matrix first: later builds find its numbers in the buffers
        U[s].upload(H, 0.0); U[s].run();
        ref[s] = U[s].snapshot();
    }
    float ms = 0;
    for (const char* mode : {"1", "4"}) {
        setenv("MBFIR_CHOL_SPLIT", mode, 1);
        double tot = 0; int n = 0;
        for (int r = 0; r < reps + 1; ++r) {
            U[0].upload(H, 0.0);
            n = U[0].run(e0, e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (r) tot += ms;
        }
        printf("np %d lanes %d  MBFIR_CHOL_SPLIT=%s: %.1f us per factorisation (%d launches)  %s\n", np, nl, mode, 1e3 * tot / reps, n,
               hipGetErrorString(hipGetLastError()));
    }
    // the single-launch form alone: bit-identical?
    setenv("MBFIR_CHOL_SPLIT", "4", 1);
    U[0].upload(H, 1.0); U[0].run(); U[0].upload(H, 0.0); U[0].run();
    { const auto got = U[0].snapshot(); printf("single launch, alone: %ld words differ from the per-step form\n", diff(U[0], ref[0], got)); }
    dag_log_on(U[0]);
    U[0].upload(H, 0.0); U[0].run(); hipDeviceSynchronize();
    dag_stats("one unit alone");
    // several units in flight, staggered: every build checked
    long bad = 0; int builds = 0;
    for (int r = 0; r < reps; ++r) {
        for (int s = 0; s < ns; ++s) { U[s].upload(H, 1.0); }
        for (int s = 0; s < ns; ++s) { U[s].run(); if (s % 2) U[s].run(); }          // some streams run two builds back to back
        for (int s = 0; s < ns; ++s) hipStreamSynchronize(U[s].st);
        for (int s = 0; s < ns; ++s) U[s].upload(H, 0.0);
        for (int s = 0; s < ns; ++s) U[(s + r) % ns].run();
        for (int s = 0; s < ns; ++s) { const auto got = U[s].snapshot(); bad += diff(U[s], ref[s], got); ++builds; }
    }
    printf("single launch, %d units in flight: %ld words differ over %d checked builds  %s\n", ns, bad, builds, hipGetErrorString(hipGetLastError()));
    // throughput with ns units in flight
    for (const char* mode : {"1", "4"}) {
        setenv("MBFIR_CHOL_SPLIT", mode, 1);
        for (int s = 0; s < ns; ++s) U[s].upload(H, 0.0);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipDeviceSynchronize();
        const int inner = 8;
        if (mode[0] == '4') dag_log_on(U[0]);             // (every build of unit 0 overwrites the log: the last one stays)
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < inner; ++r) for (int s = 0; s < ns; ++s) U[s].run();
        hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        int lanes = 0; for (int s = 0; s < ns; ++s) lanes += U[s].nl;
        printf("MBFIR_CHOL_SPLIT=%s, %d units in flight (%d lanes): %.1f us per unit-build, %.2f us per design-build\n", mode, ns, lanes, us / (inner * ns), us / (inner * lanes));
        if (mode[0] == '4') dag_stats("units in flight");

    }
    return 0;
}
