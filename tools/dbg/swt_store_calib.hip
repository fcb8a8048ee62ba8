// write-pattern calibration for the stationary packet table (n = 16384 samples per column, 64 leaf columns per node):
//   BURST 1: per step, 64 stores of 512 contiguous bytes, one per leaf column (what a sliding-window walk produces)
//   BURST 4 / 16: the same bytes, 4 / 16 consecutive steps of a column back to back (2 KiB / 8 KiB bursts per column)
//   BURST 256: one wavefront writes whole columns one after the other (128 KiB runs)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BURST> __global__ __launch_bounds__(64) void k_w(double *y, int nodes)
{
    const int n = 16384;
    double *base = y + (size_t)blockIdx.x * 64 * n + threadIdx.x;       // blockIdx.x = (signal, node)
    const double v = threadIdx.x;
    for (int m0 = 0; m0 < 256; m0 += BURST)
        for (int c = 0; c < 64; ++c)
#pragma unroll
            for (int j = 0; j < BURST; ++j) base[(size_t)c * n + (size_t)(m0 + j) * 64] = v + c;
}
int main()
{
    const size_t blocks = 4096, bytes = blocks * 64 * 16384 * 8;       // 32 GiB
    double *y;
    if (hipMalloc(&y, bytes) != hipSuccess) return 1;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_w<1>, dim3(blocks), dim3(64), 0, 0, y, 64);
        hipLaunchKernelGGL(k_w<4>, dim3(blocks), dim3(64), 0, 0, y, 64);
        hipLaunchKernelGGL(k_w<16>, dim3(blocks), dim3(64), 0, 0, y, 64);
        hipLaunchKernelGGL(k_w<256>, dim3(blocks), dim3(64), 0, 0, y, 64);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("done %zu bytes per kernel\n", bytes);
    return 0;
}
