// Where the diagonal block of a split step spends its time: 8 lanes, merged split step, timestamps of lane 0's
// diagonal block (10 ns ticks of s_memrealtime) at: 0 tile loads issued | 1 L_k,k-1 in LDS | 2 panel k-1 update done
// | 3 operands settled | 4 potf2 + image done | 5 image stored, flag raised; and the launch durations.
#define CHOL_TRACE 1
#define CHOL_TRACE_D 1
#include "../../multiband-rf-pulse-design_amd/csrc/chol.hip"
#include <vector>
#include <cmath>
using namespace mbfir;
int main(int argc, char** argv) {
    const int np = argc > 1 ? atoi(argv[1]) : 1024, nl = argc > 2 ? atoi(argv[2]) : 8;
    std::vector<double> H(np * (size_t)np);
    for (int i = 0; i < np; ++i) for (int j = 0; j < np; ++j) H[i * (size_t)np + j] = 1.0 / (1.0 + abs(i - j)) + (i == j ? 2.0 : 0.0);
    const size_t lane_doubles = 4 * (size_t)np * np + 80 * (size_t)np + 64, lane_bytes = lane_doubles * 8;
    double* base; hipMalloc(&base, lane_bytes * nl);
    double *dH = base, *dM = base + (size_t)np * np, *dMt = dM + (size_t)np * np, *dW = dMt + (size_t)np * np;
    int* df = reinterpret_cast<int*>(dW + (size_t)np * np + 70 * (size_t)np);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        for (int b = 0; b < nl; ++b) hipMemcpy(reinterpret_cast<char*>(dH) + b * lane_bytes, H.data(), np * (size_t)np * 8, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        int n = chol_inv_launch(dH, dM, dMt, dW, np, df, 0, nullptr, nullptr, nullptr, nl, lane_bytes, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("rep %d: %.1f us, %d launches (%s)\n", rep, ms * 1e3, n, hipGetErrorString(hipGetLastError()));
    }
    {   // inside the elimination: one more run with the slab stamps on (they cover every step; the last one stays)
        int one = 1; hipMemcpyToSymbol(HIP_SYMBOL(g_trace2_k), &one, sizeof(int));
        for (int b = 0; b < nl; ++b) hipMemcpy(reinterpret_cast<char*>(dH) + b * lane_bytes, H.data(), np * (size_t)np * 8, hipMemcpyHostToDevice);
        chol_inv_launch(dH, dM, dMt, dW, np, df, 0, nullptr, nullptr, nullptr, nl, lane_bytes, nullptr);
        hipDeviceSynchronize();
        long long t2[64]; hipMemcpyFromSymbol(t2, HIP_SYMBOL(g_trace2), sizeof(t2));
        printf("slab 1 of the last step (10 ns ticks): accumulators->LDS+barrier %lld | operand loads %lld | 16 pivots %lld | stores %lld | barrier %lld | rank-16 updates (to slab 2 start) %lld ; last image %lld\n",
               t2[1] - t2[0], t2[2] - t2[1], t2[3] - t2[2], t2[4] - t2[3], t2[5] - t2[4], t2[6] - t2[5], t2[8] - t2[7]);
    }
    long long tr[512]; hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_trace), sizeof(tr));
    for (int k : {1, 4, 8, 12}) { printf("k=%2d D block (10 ns ticks):", k); for (int s = 1; s < 6; ++s) printf(" %lld", tr[k * 16 + s] - tr[k * 16 + s - 1]); printf("  | total %lld\n", tr[k * 16 + 5] - tr[k * 16]); }
    for (int k : {1, 4, 8, 12}) printf("k=%2d: D start -> next D start %lld ticks\n", k, tr[(k + 1) * 16] - tr[k * 16]);
    return 0;
}
