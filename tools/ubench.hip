// ubench.hip -- micro-benchmarks that calibrate the roofline terms used in DESIGN.md on the
// actual MI355X: FP64 FMA issue rate, LDS 16-byte read rate, HBM streaming copy.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ILP>
__global__ void k_fma64(double *out, double a, double b, int iters)
{
    double acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    if (s == 123.456) out[0] = s;
}

__global__ void k_lds128(double *out, int iters)
{
    extern __shared__ double2 sm[];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) { sm[i].x = i; sm[i].y = -i; }
    __syncthreads();
    double2 acc = {0, 0};
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double2 v = sm[(idx + k * 64) & 2047];
            acc.x += v.x; acc.y += v.y;
        }
        idx = (idx + 1) & 2047;
    }
    if (acc.x == 123.456) out[0] = acc.y;
}

__global__ void k_copy(const double4 *__restrict__ in, double4 *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_fill(double4 *__restrict__ out, size_t n)
{
    double4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}

__global__ void k_fill16(float4 *__restrict__ out, size_t n)
{
    float4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}
__global__ void k_fill16_nt(float4 *__restrict__ out, size_t n)
{
    float4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    { typedef float v4f __attribute__((ext_vector_type(4))); v4f vv = {1, 2, 3, 4}; __builtin_nontemporal_store(vv, (v4f *)&out[i]); }
}
// each block writes a contiguous 416 KiB region (like one packet table), 16 B per lane
__global__ void k_fill_chunks(float4 *__restrict__ out, size_t chunk16, size_t nchunks)
{
    float4 v = {1, 2, 3, 4};
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x)
        for (size_t i = threadIdx.x; i < chunk16; i += blockDim.x) out[c * chunk16 + i] = v;
}
// 32 bytes per lane as two 16-byte stores (lane stride 32 B), like the wpd kernel's V4 stores
__global__ void k_fill_chunks32(float4 *__restrict__ out, size_t chunk16, size_t nchunks)
{
    float4 v = {1, 2, 3, 4};
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x)
        for (size_t i = threadIdx.x; 2 * i + 1 < chunk16; i += blockDim.x) { out[c * chunk16 + 2 * i] = v; out[c * chunk16 + 2 * i + 1] = v; }
}
// same bytes, but every store instruction covers a contiguous 1 KiB per wave
__global__ void k_fill_chunks16x2(float4 *__restrict__ out, size_t chunk16, size_t nchunks)
{
    float4 v = {1, 2, 3, 4};
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x)
        for (size_t i = threadIdx.x; i + blockDim.x < chunk16; i += 2 * blockDim.x) { out[c * chunk16 + i] = v; out[c * chunk16 + i + blockDim.x] = v; }
}
__global__ void k_read16(const float4 *__restrict__ in, float *out, size_t n)
{
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}

template <typename F> float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    double *d;
    CK(hipMalloc(&d, 1 << 20));
    const int iters = 4096;
    for (int waves : {4, 8, 16, 32}) {
        const int threads = 256, blocks = 256 * waves / 4;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_fma64<8>, dim3(blocks), dim3(threads), 0, 0, d, 1.0000001, 1e-9, iters); }, 5);
        double flops = 2.0 * 8 * iters * (double)threads * blocks;
        printf("fma64 ILP8 %2d waves/CU: %.3f ms  %.1f TFLOP/s\n", waves, ms, flops / ms / 1e9);
    }
    for (int waves : {4, 8, 16}) {
        const int threads = 256, blocks = 256 * waves / 4;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_lds128, dim3(blocks), dim3(threads), 32768, 0, d, 2048); }, 5);
        double bytes = 16.0 * 8 * 2048 * (double)threads * blocks;
        printf("ds_read_b128 %2d waves/CU: %.3f ms  %.1f TB/s (%.1f B/clk/CU at 2.4 GHz)\n", waves, ms, bytes / ms / 1e9, bytes / ms / 1e9 * 1e3 / 256 / 2400);
    }
    size_t n = (size_t)1 << 27;   // 4 GiB of double4
    double4 *a, *b;
    CK(hipMalloc(&a, n * 32)); CK(hipMalloc(&b, n * 32));
    CK(hipMemset(a, 0, n * 32));
    float ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, a, b, n); }, 5);
    printf("copy 4 GiB -> 4 GiB: %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * n * 32 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_fill, dim3(256 * 16), dim3(256), 0, 0, b, n); }, 5);
    printf("fill 4 GiB: %.3f ms  %.2f TB/s (write only)\n", ms, 1.0 * n * 32 / ms / 1e9);
    size_t n16 = n * 2;
    for (int bl : {1024, 2048, 4096, 8192, 16384}) {
        ms = time_ms([&] { hipLaunchKernelGGL(k_fill16, dim3(bl), dim3(256), 0, 0, (float4 *)b, n16); }, 5);
        printf("fill16 %5d blocks: %.3f ms %.2f TB/s\n", bl, ms, n16 * 16.0 / ms / 1e9);
    }
    ms = time_ms([&] { hipLaunchKernelGGL(k_fill16_nt, dim3(4096), dim3(256), 0, 0, (float4 *)b, n16); }, 5);
    printf("fill16 nontemporal 4096 blocks: %.3f ms %.2f TB/s\n", ms, n16 * 16.0 / ms / 1e9);
    for (int bl : {512, 1024, 2048}) {
        size_t chunk16 = 425984 / 16, nch = n16 / chunk16;
        ms = time_ms([&] { hipLaunchKernelGGL(k_fill_chunks, dim3(bl), dim3(512), 0, 0, (float4 *)b, chunk16, nch); }, 5);
        printf("fill 416KiB chunks %4d blocks x512: %.3f ms %.2f TB/s\n", bl, ms, nch * chunk16 * 16.0 / ms / 1e9);
    }
    for (int bl : {512, 1024}) {
        size_t chunk16 = 425984 / 16, nch = n16 / chunk16;
        ms = time_ms([&] { hipLaunchKernelGGL(k_fill_chunks32, dim3(bl), dim3(512), 0, 0, (float4 *)b, chunk16, nch); }, 5);
        printf("fill chunks 2x16B/lane stride32 %4d blocks: %.3f ms %.2f TB/s\n", bl, ms, nch * chunk16 * 16.0 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_fill_chunks16x2, dim3(bl), dim3(512), 0, 0, (float4 *)b, chunk16, nch); }, 5);
        printf("fill chunks 2x16B/lane contiguous %4d blocks: %.3f ms %.2f TB/s\n", bl, ms, nch * chunk16 * 16.0 / ms / 1e9);
    }
    for (int bl : {2048, 8192}) {
        ms = time_ms([&] { hipLaunchKernelGGL(k_read16, dim3(bl), dim3(256), 0, 0, (const float4 *)a, (float *)d, n16); }, 5);
        printf("read16 %5d blocks: %.3f ms %.2f TB/s\n", bl, ms, n16 * 16.0 / ms / 1e9);
    }
    return 0;
}
