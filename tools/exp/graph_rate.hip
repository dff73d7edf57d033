// Experiment: 200 dependent tiny kernels per iteration -- stream launches vs one hipGraph launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_tiny(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1e-9; }
int main() {
    const int n = 34000, K = 200, reps = 50;
    double* p; hipMalloc(&p, n * 8); hipMemset(p, 0, n * 8);
    hipStream_t st; hipStreamCreate(&st);
    auto run_stream = [&]() { for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_tiny, dim3((n + 255) / 256), dim3(256), 0, st, p, n); };
    run_stream(); hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) run_stream();
    hipStreamSynchronize(st);
    double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("stream launches: %.2f us per kernel\n", el / (reps * K) * 1e6);
    hipGraph_t g; hipGraphExec_t ge;
    t0 = std::chrono::steady_clock::now();
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    run_stream();
    hipStreamEndCapture(st, &g);
    double tc = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    t0 = std::chrono::steady_clock::now();
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    double ti = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("capture of %d kernels: %.1f us, instantiate: %.1f us\n", K, tc * 1e6, ti * 1e6);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("graph launch:    %.2f us per kernel (%s)\n", el / (reps * K) * 1e6, hipGetErrorString(hipGetLastError()));
    return 0;
}
