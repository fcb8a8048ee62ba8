// ring_probe.hip -- VERDICT r5 item 1: can the intermediate image of config 4's two transposing passes stay on chip?
// The memory shapes of wx_lattice2d.h without arithmetic (512 x 512 Float32 images, blocks of 64 columns = 128 KiB per
// workgroup of 4 wavefronts, 2 workgroups per CU):
//   pass 1: 128 KiB contiguous of the image  ->  128 KiB contiguous of the BLOCKED intermediate
//   pass 2: 8 pieces of 16 KiB of the intermediate (one per block)  ->  512 runs of 256 bytes, 2 KiB apart, of the result
// Variants:
//   two   : two launches over the whole batch, intermediate = a second 4 GiB array (what the library does)
//   fused : ONE persistent launch; tickets hand out pass-1 tasks of group s and pass-2 tasks of group s - D alternately, the
//           intermediate of a group of G images lives in slot (group mod K) of a ring of K G MiB; pass 2 of a group waits for
//           the group's pass-1 counter, pass 1 of group g for the pass-2 counter of group g - K (the slot's previous tenant).
//           PROTO 0: plain ring stores, agent release by one lane, agent acquire by the consumer (MI355X_MICROARCH.md, "Valid forms")
//           PROTO 1: sc1 ring stores and loads, no fences (the table's third row)
// Every variant is checked word for word against the two-launch result; pass 1 adds a per-run salt so that a stale ring line
// (L1 / another XCD's L2) cannot go unnoticed.
//   build: hipcc -O3 --offload-arch=gfx950 tools/dbg/ring_probe.hip -o tools/dbg/ring_probe
//   run:   tools/dbg/ring_probe [nimg=4096]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int IMG = 512 * 512, BLK = 512 * 64, NBLK = 8;

__device__ __forceinline__ f4 ld_nt(const float *p) { return __builtin_nontemporal_load((const f4 *)p); }
__device__ __forceinline__ void st_nt(float *p, f4 v) { __builtin_nontemporal_store(v, (f4 *)p); }
__device__ __forceinline__ f4 ld_sc1(const float *p)
{
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ f4 ld_sys(const float *p)
{
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc1(float *p, f4 v) { 
    // the wait state keeps a following VALU write of the data registers away from the store (the compiler's hazard recogniser does
    // not look inside inline asm)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void wait8(f4 &a, f4 &b, f4 &c, f4 &d, f4 &e, f4 &f, f4 &g, f4 &h)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "memory");
}

// RING: 0 = nt both sides (two-launch form), 1 = plain ring accesses, 2 = sc1 ring accesses, 3 = sc1 stores / plain loads (behind an
// acquire), 4 = sc1 stores / sc0 sc1 loads
template <int RING> __device__ __forceinline__ void pass1(const float *simg, float *zimg, int bx, float salt, int tid)
{
    const float *s = simg + bx * BLK + 4 * tid;
    float *z = zimg + bx * BLK + 4 * tid;
    f4 r[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) r[q] = ld_nt(s + 1024 * q);
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const f4 v = r[q] + salt;
        if (RING == 0) st_nt(z + 1024 * q, v);
        else if (RING == 1) *(f4 *)(z + 1024 * q) = v;
        else st_sc1(z + 1024 * q, v);
    }
}
template <int RING> __device__ __forceinline__ void pass2(const float *zimg, float *dimg, int bx, int tid)
{
    const float *z = zimg + 4096 * bx + 4 * tid;
    float *d = dimg + 64 * bx + 4 * (tid & 15) + 512 * (tid >> 4);
    f4 r[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const float *p = z + (q >> 2) * BLK + (q & 3) * 1024;
        if (RING == 0) r[q] = ld_nt(p);
        else if (RING == 1 || RING == 3) r[q] = *(const f4 *)p;
        else if (RING == 2) r[q] = ld_sc1(p);
        else r[q] = ld_sys(p);
    }
    if (RING == 2 || RING == 4) {
#pragma unroll
        for (int q = 0; q < 32; q += 8) wait8(r[q], r[q + 1], r[q + 2], r[q + 3], r[q + 4], r[q + 5], r[q + 6], r[q + 7]);
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) st_nt(d + 512 * 16 * q, r[q] * 2.0f);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_p1(const float *src, float *z, float salt)
{
    pass1<0>(src + (size_t)blockIdx.y * IMG, z + (size_t)blockIdx.y * IMG, blockIdx.x, salt, threadIdx.x);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_p2(const float *z, float *dst)
{
    pass2<0>(z + (size_t)blockIdx.y * IMG, dst + (size_t)blockIdx.y * IMG, blockIdx.x, threadIdx.x);
}

// ctl[0] = ticket; ctl[32 (1 + 2 g)] = pass-1 tasks of group g done; ctl[32 (2 + 2 g)] = pass-2 tasks of group g done
template <int PROTO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_fused(const float *src, float *ring, float *dst, int nimg,
                                                                                         int G, int D, int K, unsigned *ctl, float salt)
{
    __shared__ unsigned s_t;
    const int tid = threadIdx.x;
    const int NG = (nimg + G - 1) / G, TPG = G * NBLK;
    const unsigned total = (unsigned)(NG + D) * 2u * TPG;
    for (;;) {
        if (tid == 0) s_t = __hip_atomic_fetch_add(&ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned t = s_t;
        __syncthreads();
        if (t >= total) break;
        const int step = t / (2 * TPG), r = t % (2 * TPG), second = r & 1, k = r >> 1;
        const int g = second ? step - D : step;
        const int img = g * G + k / NBLK, bx = k % NBLK;
        const bool in_range = g >= 0 && g < NG;             // no "continue" below: one structured body per ticket
        const bool work = in_range && img < nimg;
        if (work) {
            float *zimg = ring + ((size_t)(g % K) * G + k / NBLK) * IMG;
            // what this task waits for: pass 2 the group's pass-1 counter, pass 1 the pass-2 counter of the slot's previous tenant
            const int wg = second ? g : g - K;
            if (wg >= 0) {
                if (tid == 0) {
                    const unsigned *c = &ctl[32 * (1 + (second ? 0 : 1) + 2 * wg)];
                    unsigned spins = 0;
                    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)TPG) {
                        __builtin_amdgcn_s_sleep(8);
                        ++spins;                              // watchdog: report instead of hanging the box
                        if ((spins & 1023u) == 0 && __hip_atomic_load(&ctl[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        if (spins > 300000u) {
                            __hip_atomic_fetch_add(&ctl[16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    if (PROTO == 0 || PROTO == 2) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                __syncthreads();
            }
            if (!second) pass1<PROTO + 1>(src + (size_t)img * IMG, zimg, bx, salt, tid);
            else pass2<PROTO + 1>(zimg, dst + (size_t)img * IMG, bx, tid);
            // every wave waits for its own accesses, then one lane signals for the workgroup
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (in_range && tid == 0) {                           // a ragged last group still counts its tasks
            if (PROTO == 0 && work && !second) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(&ctl[32 * (1 + second + 2 * g)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void k_cmp(const float *a, const float *b, size_t n, unsigned long long *bad)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(bad, c);
}
__global__ void k_fill(float *a, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (float)((i * 2654435761ull >> 7) & 0xffff) * 0.25f;
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

int main(int argc, char **argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int nimg = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t n = (size_t)nimg * IMG;
    float *src, *z, *dref, *dst, *ring;
    unsigned *ctl;
    unsigned long long *bad;
    const size_t ring_max = (size_t)512 << 20;
    CK(hipMalloc(&src, 4 * n)); CK(hipMalloc(&z, 4 * n)); CK(hipMalloc(&dref, 4 * n)); CK(hipMalloc(&dst, 4 * n));
    CK(hipMalloc(&ring, ring_max)); CK(hipMalloc(&ctl, 4 * 32 * (2 * 4096 + 16))); CK(hipMalloc(&bad, 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, src, n);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REP = 10;
    const double gb = 4.0 * 4 * n / 1e9;                  // bytes the four legs move
    // two launches
    float salt = 1.0f;
    for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL(k_p1, dim3(NBLK, nimg), dim3(256), 0, 0, src, z, salt);
        hipLaunchKernelGGL(k_p2, dim3(NBLK, nimg), dim3(256), 0, 0, z, dref);
    }
    CK(hipEventRecord(e0));
    for (int i = 0; i < REP; ++i) {
        hipLaunchKernelGGL(k_p1, dim3(NBLK, nimg), dim3(256), 0, 0, src, z, salt);
        hipLaunchKernelGGL(k_p2, dim3(NBLK, nimg), dim3(256), 0, 0, z, dref);
    }
    CK(hipEventRecord(e1));
    float ms = time_ms(e0, e1) / REP;
    printf("%-44s %7.3f ms  %5.2f TB/s over the 4 legs, %5.2f of 8 TB/s on in + out\n", "two launches, 4 GiB intermediate", ms, gb / ms, gb / 2 / ms / 8);
    CK(hipEventRecord(e0));
    for (int i = 0; i < REP; ++i) hipLaunchKernelGGL(k_p1, dim3(NBLK, nimg), dim3(256), 0, 0, src, z, salt);
    CK(hipEventRecord(e1));
    ms = time_ms(e0, e1) / REP;
    printf("%-44s %7.3f ms\n", "  pass 1 alone", ms);
    CK(hipEventRecord(e0));
    for (int i = 0; i < REP; ++i) hipLaunchKernelGGL(k_p2, dim3(NBLK, nimg), dim3(256), 0, 0, z, dref);
    CK(hipEventRecord(e1));
    ms = time_ms(e0, e1) / REP;
    printf("%-44s %7.3f ms\n", "  pass 2 alone", ms);
    // sub-batched launches through a ring (no persistence): groups of G images, one stream
    for (int G : {32, 128}) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < REP; ++i)
            for (int g0 = 0; g0 < nimg; g0 += G) {
                const int gi = std::min(G, nimg - g0);
                float *slot = ring + (size_t)((g0 / G) & 1) * G * IMG;
                hipLaunchKernelGGL(k_p1, dim3(NBLK, gi), dim3(256), 0, 0, src + (size_t)g0 * IMG, slot, salt);
                hipLaunchKernelGGL(k_p2, dim3(NBLK, gi), dim3(256), 0, 0, slot, dst + (size_t)g0 * IMG);
            }
        CK(hipEventRecord(e1));
        ms = time_ms(e0, e1) / REP;
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(k_cmp, dim3(4096), dim3(256), 0, 0, dst, dref, n, bad);
        unsigned long long hb;
        CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        char name[96];
        snprintf(name, sizeof name, "sub-batched launches, G=%d (ring %d MiB)", G, 2 * G);
        printf("%-44s %7.3f ms  %5.2f TB/s, %5.2f  mismatches %llu\n", name, ms, gb / ms, gb / 2 / ms / 8, hb);
    }
    // persistent fused
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount;
    const int grid = 2 * cus;
    struct Cfg { int G, D, K; };
    const std::vector<Cfg> cfgs = {{4096, 1, 2}, {64, 2, 4}, {32, 3, 5}, {16, 4, 8}, {8, 8, 16}, {8, 8, 12}, {4, 16, 20}, {4, 8, 12},
                                   {4, 8, 10}, {2, 16, 20}, {2, 32, 40}, {1, 32, 40}, {1, 64, 80}};
    for (int proto = argc > 2 ? atoi(argv[2]) : 0; proto < 4; ++proto)
        for (const Cfg &c : cfgs) {
            const int G = std::min(c.G, nimg), NG = (nimg + G - 1) / G;
            if ((size_t)c.K * G * IMG * 4 > ring_max && G != nimg) continue;
            float *rg = G == nimg ? z : ring;
            const int K = G == nimg ? 1 : c.K;
            float tot = 0;
            unsigned long long hb = 0;
            for (int i = 0; i < REP + 2; ++i) {
                salt = i == REP + 1 ? 1.0f : 2.0f + i;        // the checked run uses the reference's salt, the runs before it others
                CK(hipMemsetAsync(ctl, 0, 4 * 32 * (2 * (NG + 2) + 2), 0));
                CK(hipEventRecord(e0));
                if (proto == 0) hipLaunchKernelGGL(k_fused<0>, dim3(grid), dim3(256), 0, 0, src, rg, dst, nimg, G, c.D, K, ctl, salt);
                else if (proto == 1) hipLaunchKernelGGL(k_fused<1>, dim3(grid), dim3(256), 0, 0, src, rg, dst, nimg, G, c.D, K, ctl, salt);
                else if (proto == 2) hipLaunchKernelGGL(k_fused<2>, dim3(grid), dim3(256), 0, 0, src, rg, dst, nimg, G, c.D, K, ctl, salt);
                else hipLaunchKernelGGL(k_fused<3>, dim3(grid), dim3(256), 0, 0, src, rg, dst, nimg, G, c.D, K, ctl, salt);
                CK(hipEventRecord(e1));
                const float t = time_ms(e0, e1);
                if (i >= 2) tot += t;
                if (i == REP + 1) {
                    CK(hipMemset(bad, 0, 8));
                    hipLaunchKernelGGL(k_cmp, dim3(4096), dim3(256), 0, 0, dst, dref, n, bad);
                    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
                    CK(hipMemset(dst, 0, 4 * n));
                }
            }
            ms = tot / REP;
            char name[96];
            snprintf(name, sizeof name, "fused proto %d  G=%d D=%d K=%d (ring %zu MiB)", proto, G, c.D, K, (size_t)K * G);
            unsigned wd = 0;
            CK(hipMemcpy(&wd, ctl + 16, 4, hipMemcpyDeviceToHost));
            printf("%-44s %7.3f ms  %5.2f TB/s, %5.2f  mismatches %llu  watchdog %u\n", name, ms, gb / ms, gb / 2 / ms / 8, hb, wd);
        }
    return 0;
}
