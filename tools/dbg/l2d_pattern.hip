// Memory pattern of the transposing 2-D pass (wx_lattice2d.hip) without arithmetic: what does the load / store shape alone
// sustain?  A workgroup of W wavefronts reads 16 W contiguous columns of a 512 x 512 Float32 image (32 KiB per wavefront)
// and writes, for each of the 512 rows of the transposed image, one run of 64 W bytes.
//   build: hipcc -O3 --offload-arch=gfx950 tools/dbg/l2d_pattern.hip -o tools/dbg/l2d_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int W, int REMAP, int NT>
__global__ __launch_bounds__(64 * W) void k_pat(const float *__restrict__ src, float *__restrict__ dst, int nimg)
{
    constexpr int NB = 32 / W;                           // workgroups per image
    int bx, img;
    if (REMAP == 1) {                                    // all workgroups of an image on one XCD (ids that share id % 8)
        const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
        img = (k / NB) * 8 + xcd;
        bx = k % NB;
    } else {
        bx = blockIdx.x % NB;
        img = blockIdx.x / NB;
    }
    if (img >= nimg) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *s = src + (size_t)img * 512 * 512 + (size_t)(bx * 16 * W + 16 * wave) * 512;
    float *d = dst + (size_t)img * 512 * 512 + bx * 16 * W;
    f4 r[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const f4 *p = (const f4 *)(s + 256 * q) + lane;
        r[q] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    constexpr int LPR = 4 * W;                           // lanes per row
    const int u = tid % LPR, r0 = tid / LPR;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        f4 *p = (f4 *)(d + (size_t)(16 * q + r0) * 512) + u;
        if (NT) __builtin_nontemporal_store(r[q], p);
        else *p = r[q];
    }
}

// MODE 0: plain copy (contiguous 1 KiB per instruction both ways); MODE 1: strided 64 W-byte runs on the LOAD side,
// contiguous stores; MODE 2: as k_pat but the rows of one store instruction are 32 apart instead of adjacent
template <int W, int MODE, int NT>
__global__ __launch_bounds__(64 * W) void k_alt(const float *__restrict__ src, float *__restrict__ dst, int nimg)
{
    constexpr int NB = 32 / W;
    const int bx = blockIdx.x % NB, img = blockIdx.x / NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *sc = src + (size_t)img * 512 * 512 + (size_t)(bx * 16 * W + 16 * wave) * 512;
    float *dc = dst + (size_t)img * 512 * 512 + (size_t)(bx * 16 * W + 16 * wave) * 512;
    const float *sr = src + (size_t)img * 512 * 512 + bx * 16 * W;
    float *dr = dst + (size_t)img * 512 * 512 + bx * 16 * W;
    constexpr int LPR = 4 * W;
    const int u = tid % LPR, r0 = tid / LPR;
    f4 r[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        // MODE 3: the gather a lattice wavefront wants: its own 16 signals (64 bytes per row), rows 64 a + 32 b + q
        const f4 *p = MODE == 3 ? (const f4 *)(sr + 16 * wave + (size_t)(64 * (lane & 7) + 32 * ((lane >> 3) & 1) + q) * 512) + (lane >> 4)
                    : MODE == 4 ? (const f4 *)(sr + 16 * wave + (size_t)(16 * q + (lane >> 2)) * 512) + (lane & 3)
                    : MODE == 1 ? (const f4 *)(sr + (size_t)(16 * q + r0) * 512) + u : (const f4 *)(sc + 256 * q) + lane;
        r[q] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        f4 *p = MODE == 2 ? (f4 *)(dr + (size_t)(q + 32 * r0) * 512) + u : (f4 *)(dc + 256 * q) + lane;
        if (NT) __builtin_nontemporal_store(r[q], p);
        else *p = r[q];
    }
}
template <int W, int MODE, int NT> void run_alt(const float *src, float *dst, int nimg, const char *name)
{
    const int grid = nimg * (32 / W);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_alt<W, MODE, NT>), dim3(grid), dim3(64 * W), 0, 0, src, dst, nimg);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_alt<W, MODE, NT>), dim3(grid), dim3(64 * W), 0, 0, src, dst, nimg);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-28s W %d mode %d nt %d: %.3f ms  %.2f TB/s\n", name, W, MODE, NT, ms, 2.0 * nimg * 512 * 512 * 4 / ms * 1e-9);
}

template <int W, int REMAP, int NT> void run(const float *src, float *dst, int nimg, const char *name)
{
    const int grid = nimg * (32 / W);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_pat<W, REMAP, NT>), dim3(grid), dim3(64 * W), 0, 0, src, dst, nimg);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_pat<W, REMAP, NT>), dim3(grid), dim3(64 * W), 0, 0, src, dst, nimg);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-28s W %d remap %d nt %d: %.3f ms  %.2f TB/s\n", name, W, REMAP, NT, ms, 2.0 * nimg * 512 * 512 * 4 / ms * 1e-9);
}

int main()
{
    const int nimg = 4096;
    float *a, *b;
    hipMalloc(&a, (size_t)nimg * 512 * 512 * 4);
    hipMalloc(&b, (size_t)nimg * 512 * 512 * 4);
    hipMemset(a, 0, (size_t)nimg * 512 * 512 * 4);
#define R(W, RM, NT) run<W, RM, NT>(a, b, nimg, "copy-transposed");
#define A(W, M, NT, nm) run_alt<W, M, NT>(a, b, nimg, nm);
    A(2, 0, 0, "plain copy") A(2, 0, 1, "plain copy") A(4, 0, 1, "plain copy")
    A(2, 1, 0, "strided loads") A(2, 1, 1, "strided loads") A(4, 1, 1, "strided loads") A(8, 1, 1, "strided loads")
    A(2, 3, 1, "wave gather 64B far rows") A(4, 3, 1, "wave gather 64B far rows") A(4, 3, 0, "wave gather 64B far rows") A(8, 3, 1, "wave gather 64B far rows") A(16, 3, 1, "wave gather 64B far rows")
    A(2, 4, 1, "wave gather 64B adj rows") A(4, 4, 1, "wave gather 64B adj rows") A(8, 4, 1, "wave gather 64B adj rows")
    A(2, 2, 0, "stores rows 32 apart") A(2, 2, 1, "stores rows 32 apart") A(4, 2, 1, "stores rows 32 apart")
    R(2, 0, 0) R(2, 1, 0) R(2, 0, 1) R(2, 1, 1)
    R(4, 0, 0) R(4, 1, 0) R(4, 0, 1) R(4, 1, 1)
    R(8, 0, 0) R(8, 1, 0) R(8, 0, 1) R(8, 1, 1)
    R(16, 0, 0) R(16, 1, 0) R(16, 0, 1) R(16, 1, 1)
    return 0;
}
