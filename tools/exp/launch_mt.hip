// Host-side launch throughput with several threads, one stream each: how many kernel launches per second can T threads
// enqueue, and how long does the GPU take to drain them (tiny kernel, ~2 us).   hipcc -O3 --offload-arch=gfx950 launch_mt.hip -o launch_mt_exp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_tiny(double* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1e-9;
}
int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20000;
    for (int T : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st(T);
        std::vector<double*> buf(T);
        for (int t = 0; t < T; ++t) { hipStreamCreate(&st[t]); hipMalloc(&buf[t], 1 << 20); hipMemset(buf[t], 0, 1 << 20); }
        hipDeviceSynchronize();
        std::vector<double> enq(T), tot(T);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                auto a = std::chrono::steady_clock::now();
                for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(256), 0, st[t], buf[t], 16384);
                auto b = std::chrono::steady_clock::now();
                hipStreamSynchronize(st[t]);
                auto c = std::chrono::steady_clock::now();
                enq[t] = std::chrono::duration<double, std::micro>(b - a).count() / launches;
                tot[t] = std::chrono::duration<double, std::micro>(c - a).count() / launches;
            });
        for (auto& x : th) x.join();
        double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %d: enqueue %.2f us/launch/thread, enqueue+drain %.2f us/launch/thread, total %.0f launches/ms\n", T, enq[0], tot[0],
               1e3 * T * launches / wall);
        for (int t = 0; t < T; ++t) { hipFree(buf[t]); hipStreamDestroy(st[t]); }
    }
    return 0;
}
