// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access patterns of wx_lattice.hip (MI355X_MICROARCH.md: "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  One wavefront per
// 32 KiB block, 32 x 16-byte accesses per lane, every byte touched exactly once:
//   pattern 0: fully coalesced (instruction i covers the contiguous KiB i)
//   pattern 1: 8 complete 128-byte lines per instruction, lines 512 B apart (loads of k_lat_wpt, stores of k_lat_iwpt)
//   pattern 2: 8 complete 128-byte lines per instruction, lines 4 KiB apart (stores of k_lat_wpt, loads of k_lat_iwpt)
//   pattern 3: every lane owns whole 128-byte lines: instruction i writes 16 bytes of 64 different lines (8 consecutive
//              instructions complete them); pattern 4: the same with 256-byte runs per lane
// usage: pmc_calib            (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE; 65536 blocks = 2 GiB per kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int PAT> __device__ __forceinline__ size_t off(int i, int lane)
{
    if (PAT == 0) return (size_t)i * 128 + 2 * lane;                                   // elements (doubles)
    if (PAT == 1) return (size_t)512 * (i >> 2) + 16 * (i & 3) + 64 * (lane >> 3) + 2 * (lane & 7);
    if (PAT == 2) return (size_t)512 * (i & 7) + 16 * (i >> 3) + 64 * (lane >> 3) + 2 * (lane & 7);
    if (PAT == 3) return (size_t)1024 * (i >> 3) + 16 * lane + 2 * (i & 7);          // line = 4 x 64 lines of 16 doubles
    return (size_t)2048 * (i >> 4) + 32 * lane + 2 * (i & 15);                       // 256-byte runs
}
template <int PAT> __global__ __launch_bounds__(64) void k_read(const double *x, double *sink)
{
    const double *xs = x + (size_t)blockIdx.x * 4096;
    d2 v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = *(const d2 *)(xs + off<PAT>(i, threadIdx.x));
    double s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i].x * v[i].y;
    if (s == 123.456) sink[0] = s;
}
template <int PAT, bool NT> __global__ __launch_bounds__(64) void k_write(double *y)
{
    double *ys = y + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        d2 v; v.x = threadIdx.x + i; v.y = blockIdx.x;
        if (NT) __builtin_nontemporal_store(v, (d2 *)(ys + off<PAT>(i, threadIdx.x)));
        else *(d2 *)(ys + off<PAT>(i, threadIdx.x)) = v;
    }
}
int main()
{
    const size_t nb = 65536, bytes = nb * 32768;
    double *x, *y, *s;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess || hipMalloc(&s, 64) != hipSuccess) return 1;
    (void)hipMemset(x, 0, bytes);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_read<0>, dim3(nb), dim3(64), 0, 0, x, s);
        hipLaunchKernelGGL(k_read<1>, dim3(nb), dim3(64), 0, 0, x, s);
        hipLaunchKernelGGL(k_read<2>, dim3(nb), dim3(64), 0, 0, x, s);
        hipLaunchKernelGGL((k_write<0, false>), dim3(nb), dim3(64), 0, 0, y);
        hipLaunchKernelGGL((k_write<1, false>), dim3(nb), dim3(64), 0, 0, y);
        hipLaunchKernelGGL((k_write<2, false>), dim3(nb), dim3(64), 0, 0, y);
        hipLaunchKernelGGL((k_write<3, false>), dim3(nb), dim3(64), 0, 0, y);
        hipLaunchKernelGGL((k_write<4, false>), dim3(nb), dim3(64), 0, 0, y);
        hipLaunchKernelGGL((k_write<0, true>), dim3(nb), dim3(64), 0, 0, y);
        hipLaunchKernelGGL((k_write<2, true>), dim3(nb), dim3(64), 0, 0, y);
        hipLaunchKernelGGL((k_write<3, true>), dim3(nb), dim3(64), 0, 0, y);
        hipLaunchKernelGGL((k_write<4, true>), dim3(nb), dim3(64), 0, 0, y);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("done: every kernel moves %zu bytes\n", bytes);
    return 0;
}
