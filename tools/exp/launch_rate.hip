// Experiment: host-side kernel launch throughput with T threads, one stream each (empty kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#include <vector>
#include <chrono>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
int main() {
    for (int T : {1, 2, 4, 8}) {
        const int n = 20000;
        std::vector<hipStream_t> st(T);
        for (auto& s : st) hipStreamCreate(&s);
        auto work = [&](int t) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[t], nullptr); hipStreamSynchronize(st[t]); };
        work(0);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %d: %.2f M launches/s total, %.2f us per launch per thread\n", T, T * n / el / 1e6, el / n * 1e6);
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
