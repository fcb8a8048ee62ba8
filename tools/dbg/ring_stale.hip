// ring_stale.hip -- companion of ring_probe.hip: does a consumer whose XCD's L2 (and CU's L1) HOLDS an older copy of a ring line see the
// new bytes?  The ring of the fused 2-D kernel is rewritten in place every K groups, by workgroups on any XCD, so this is the case that
// matters and the one a streaming benchmark cannot provoke on purpose (its lines leave L2 between two uses).
// 16 workgroups (two per XCD).  Round i: every workgroup reads the whole buffer (4 KiB: stays in every L1; 64 KiB: self-evicts from L1) (so all eight L2s hold round i - 1's bytes),
// grid barrier, workgroup i mod 16 rewrites the buffer with value i through sc1 stores, waits, signals; every workgroup then reads
// the buffer again and counts words that are not i.  Load flavours: 0 plain without acquire (positive control: must show stale
// words), 1 plain behind an agent acquire, 2 sc1 loads, 3 sc0 sc1 loads.
//   build: hipcc -O3 --offload-arch=gfx950 tools/dbg/ring_stale.hip -o tools/dbg/ring_stale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int NWG = 16;

__device__ __forceinline__ f4 ld_sc1(const float *p)
{
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ f4 ld_plain(const float *p)
{
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ f4 ld_sys(const float *p)
{
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc1(float *p, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory"); }

__device__ __forceinline__ void grid_wait(unsigned *c, unsigned target)
{
    if (threadIdx.x == 0)
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
}

template <int LD> __device__ __forceinline__ unsigned read_count(const float *buf, float want, int tid, int WORDS)
{
    unsigned bad = 0;
    for (int k = 0; k < WORDS / 4 / 256; ++k) {
        const float *p = buf + 4 * (tid + 256 * k);
        f4 v;
        if (LD <= 1) v = ld_plain(p);
        else if (LD == 2) v = ld_sc1(p);
        else v = ld_sys(p);
        bad += (v.x != want) + (v.y != want) + (v.z != want) + (v.w != want);
    }
    return bad;
}

template <int LD> __global__ __launch_bounds__(256) void k_stale(float *buf, unsigned *ctl, int rounds, unsigned long long *stale, int WORDS)
{
    const int tid = threadIdx.x, wg = blockIdx.x;
    unsigned long long bad = 0;
    for (int i = 1; i <= rounds; ++i) {
        // everybody holds round i - 1's bytes in its caches
        (void)read_count<LD>(buf, (float)(i - 1), tid, WORDS);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grid_wait(&ctl[0], (unsigned)(NWG * i));
        if (wg == i % NWG) {
            const f4 v = {(float)i, (float)i, (float)i, (float)i};
            for (int k = 0; k < WORDS / 4 / 256; ++k) st_sc1(buf + 4 * (tid + 256 * k), v);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&ctl[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        grid_wait(&ctl[32], (unsigned)i);
        if (LD == 1) {
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
        bad += read_count<LD>(buf, (float)i, tid, WORDS);
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&ctl[64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grid_wait(&ctl[64], (unsigned)(NWG * i));          // nobody is still reading when the next round's writer starts
    }
    if (bad) atomicAdd(stale, bad);
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    float *buf;
    unsigned *ctl;
    unsigned long long *stale, h;
    CK(hipMalloc(&buf, 4 * 16384)); CK(hipMalloc(&ctl, 4 * 128)); CK(hipMalloc(&stale, 8));
    const int rounds = 2000;
    const char *names[4] = {"plain loads, no acquire (control)", "plain loads behind an agent acquire", "sc1 loads", "sc0 sc1 loads"};
    for (int WORDS : {1024, 16384})
    for (int ld = 0; ld < 4; ++ld) {
        CK(hipMemset(buf, 0, 4 * WORDS)); CK(hipMemset(ctl, 0, 4 * 128)); CK(hipMemset(stale, 0, 8));
        if (ld == 0) hipLaunchKernelGGL(k_stale<0>, dim3(NWG), dim3(256), 0, 0, buf, ctl, rounds, stale, WORDS);
        else if (ld == 1) hipLaunchKernelGGL(k_stale<1>, dim3(NWG), dim3(256), 0, 0, buf, ctl, rounds, stale, WORDS);
        else if (ld == 2) hipLaunchKernelGGL(k_stale<2>, dim3(NWG), dim3(256), 0, 0, buf, ctl, rounds, stale, WORDS);
        else hipLaunchKernelGGL(k_stale<3>, dim3(NWG), dim3(256), 0, 0, buf, ctl, rounds, stale, WORDS);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&h, stale, 8, hipMemcpyDeviceToHost));
        printf("%6d-byte buffer  %-40s stale words %llu of %llu\n", 4 * WORDS, names[ld], h, (unsigned long long)rounds * NWG * WORDS);
    }
    return 0;
}
