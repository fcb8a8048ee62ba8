// pkfma_probe: issue rate of v_fma_f64 / v_fma_f32 / v_pk_fma_f32 (vector and scalar-broadcast multiplier) on gfx950, per CU occupancy.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/dbg/pkfma_probe tools/dbg/pkfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int ILP> __global__ void k_f64(double *out, double a, double b, int iters)
{
    double acc[ILP];
    for (int i = 0; i < ILP; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = fma(acc[i], a, b);
    double s = 0;
    for (int i = 0; i < ILP; ++i) s += acc[i];
    if (s == 12345.678) out[0] = s;
}
template <int ILP> __global__ void k_f32(float *out, float a, float b, int iters)
{
    float acc[ILP];
    for (int i = 0; i < ILP; ++i) acc[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = fmaf(acc[i], a, b);
    float s = 0;
    for (int i = 0; i < ILP; ++i) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}
// x = c * y + x with a wave-uniform c (the lattice's shears): two independent accumulator sets so the chain is a real FMA chain
template <int ILP> __global__ void k_pk(float *out, float a, float b, int iters)
{
    f2 x[ILP], y[ILP];
    for (int i = 0; i < ILP; ++i) { x[i] = (f2){threadIdx.x * 1e-3f + i, 1.0f + i}; y[i] = (f2){0.5f + i, threadIdx.x * 2e-3f}; }
    const f2 ca = {a, a}, cb = {b, b};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            x[i] = __builtin_elementwise_fma(ca, y[i], x[i]);
            y[i] = __builtin_elementwise_fma(cb, x[i], y[i]);
        }
    f2 s = {0, 0};
    for (int i = 0; i < ILP; ++i) s += x[i] + y[i];
    if (s.x + s.y == 12345.678f) out[0] = s.x;
}
template <int ILP> __global__ void k_f64shear(double *out, double a, double b, int iters)
{
    double x[ILP], y[ILP];
    for (int i = 0; i < ILP; ++i) { x[i] = threadIdx.x * 1e-3 + i; y[i] = 0.5 + i; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            x[i] = fma(a, y[i], x[i]);
            y[i] = fma(b, x[i], y[i]);
        }
    double s = 0;
    for (int i = 0; i < ILP; ++i) s += x[i] + y[i];
    if (s == 12345.678) out[0] = s;
}
template <typename F> float time_ms(F f)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}
int main()
{
    void *d; hipMalloc(&d, 1024);
    const int iters = 4000, cus = 256;
    for (int waves : {4, 8, 12, 16}) {
        const int threads = 64, blocks = cus * waves;
        const double n = (double)blocks * threads * iters;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_f64<8>, dim3(blocks), dim3(threads), 0, 0, (double *)d, 1.0000001, 1e-9, iters); });
        printf("v_fma_f64     chains 8, %2d waves/CU: %.3f ms  %6.1f TFLOP/s  (%.2f cycles per wave-instruction per SIMD at 2.4 GHz)\n", waves, ms, n * 8 * 2 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 8.0 * waves / 4));
        ms = time_ms([&] { hipLaunchKernelGGL(k_f64shear<8>, dim3(blocks), dim3(threads), 0, 0, (double *)d, 1.0000001, -1e-9, iters); });
        printf("f64 shear pair chains 8, %2d waves/CU: %.3f ms  %6.1f TFLOP/s  (%.2f cycles per wave-instruction)\n", waves, ms, n * 16 * 2 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 16.0 * waves / 4));
        ms = time_ms([&] { hipLaunchKernelGGL(k_f32<8>, dim3(blocks), dim3(threads), 0, 0, (float *)d, 1.0000001f, 1e-9f, iters); });
        printf("v_fma_f32     chains 8, %2d waves/CU: %.3f ms  %6.1f TFLOP/s  (%.2f cycles per wave-instruction)\n", waves, ms, n * 8 * 2 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 8.0 * waves / 4));
        ms = time_ms([&] { hipLaunchKernelGGL(k_pk<8>, dim3(blocks), dim3(threads), 0, 0, (float *)d, 1.0000001f, -1e-9f, iters); });
        printf("v_pk_fma_f32 shear chains 8, %2d waves/CU: %.3f ms  %6.1f TFLOP/s  (%.2f cycles per wave-instruction)\n", waves, ms, n * 16 * 4 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 16.0 * waves / 4));
    }
    return 0;
}
