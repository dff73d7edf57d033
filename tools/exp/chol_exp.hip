// Experiment: time the individual Cholesky kernels (events) for N = 1024.
#include "../../multiband-rf-pulse-design_amd/csrc/chol.hip"
#include <vector>
using namespace mbfir;
int main() {
    const int np = 1024, nblk = 16;
    std::vector<double> H(np * (size_t)np);
    for (int i = 0; i < np; ++i) for (int j = 0; j < np; ++j) H[i * (size_t)np + j] = 1.0 / (1.0 + abs(i - j)) + (i == j ? 2.0 : 0.0);
    double *dH, *dH0, *dM, *dW; int* df;
    hipMalloc(&dH, np * np * 8); hipMalloc(&dH0, np * np * 8); hipMalloc(&dM, np * np * 8); hipMalloc(&dW, (np * np + 65 * np) * 8); hipMalloc(&df, 16);
    hipMemcpy(dH0, H.data(), np * np * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double* d0 = dW; double* Dfac = dW + np;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemcpy(dH, dH0, np * np * 8, hipMemcpyDeviceToDevice);
        hipMemset(dM, 0, np * np * 8);
        hipLaunchKernelGGL(k_diag_copy, dim3(4), dim3(256), 0, 0, dH, np, d0, dM);
        for (int k = 0; k < nblk; ++k) {
            const int npanel = nblk - k, nrem = nblk - k - 1;
            float a, b;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_chol_stepA, dim3(npanel + npanel * k), dim3(256), 0, 0, dH, dM, np, nblk, k, d0, 1e-13, Dfac, df);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&a, e0, e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_chol_stepB, dim3(nrem * (nrem + 1) / 2 + k + 1), dim3(256), 0, 0, dH, dM, np, nblk, k, Dfac);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&b, e0, e1);
            if (rep == 1) printf("k=%2d stepA(%3d blocks) %.1f us   stepB(%3d blocks) %.1f us\n", k, npanel + npanel * k, a * 1e3, nrem * (nrem + 1) / 2 + k + 1, b * 1e3);
        }
    }
    return 0;
}
