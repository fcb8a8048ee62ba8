// hostpath_probe: what bounds the host-array path of the C ABI.  Measures, on the GPU box:
//   first-touch of a fresh pageable array by T threads (plain / MADV_HUGEPAGE / MADV_POPULATE_WRITE), threaded memcpy into
//   faulted memory, hipHostRegister, D2H into pinned / registered / pageable memory, H2D from pageable / pinned.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/dbg/hostpath_probe tools/dbg/hostpath_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par(int T, size_t n, void (*f)(char *, size_t, void *), char *base, void *arg)
{
    std::vector<std::thread> th;
    const size_t sl = ((n / T) + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; ++t) {
        const size_t o = (size_t)t * sl;
        if (o >= n) break;
        const size_t l = o + sl <= n ? sl : n - o;
        th.emplace_back([=] { f(base + o, l, arg); });
    }
    for (auto &t : th) t.join();
}
static char *fresh(size_t n) { return (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); }
int main(int argc, char **argv)
{
    const size_t N = (size_t)(argc > 1 ? atof(argv[1]) : 3.5) << 30;
    printf("bytes %.2f GiB, hw threads %u\n", N / 1073741824.0, std::thread::hardware_concurrency());
    for (int T : {1, 8, 16, 32, 64}) {
        for (int mode = 0; mode < 4; ++mode) {
            char *p = fresh(N);
            if (mode == 1 || mode == 3) madvise(p, N, MADV_HUGEPAGE);
            const double t0 = now();
            if (mode < 2)
                par(T, N, [](char *b, size_t l, void *) { for (size_t i = 0; i < l; i += 4096) b[i] = 0; }, p, nullptr);
            else
                par(T, N, [](char *b, size_t l, void *) { if (madvise(b, l, MADV_POPULATE_WRITE)) perror("populate"); }, p, nullptr);
            const double t1 = now();
            printf("first touch T=%2d %-28s %7.1f ms  %6.1f GB/s\n", T,
                   mode == 0 ? "write per 4K page" : mode == 1 ? "hugepage + write per page" : mode == 2 ? "MADV_POPULATE_WRITE" : "hugepage + POPULATE_WRITE",
                   (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
            munmap(p, N);
        }
    }
    // memcpy into faulted memory from a 32 MiB (cache-cold-ish) source ring
    {
        char *dst = fresh(N);
        par(32, N, [](char *b, size_t l, void *) { memset(b, 1, l); }, dst, nullptr);
        char *src = fresh(N);
        par(32, N, [](char *b, size_t l, void *) { memset(b, 2, l); }, src, nullptr);
        for (int T : {1, 8, 16, 32, 64}) {
            struct A { char *src, *dst; } a = {src, dst};
            const double t0 = now();
            par(T, N, [](char *b, size_t l, void *arg) { A *a = (A *)arg; memcpy(b, a->src + (b - a->dst), l); }, dst, &a);
            const double t1 = now();
            printf("memcpy faulted->faulted T=%2d %7.1f ms  %6.1f GB/s\n", T, (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
        }
        void *d = nullptr;
        if (hipMalloc(&d, N) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
        hipMemset(d, 3, N);
        hipDeviceSynchronize();
        double t0 = now();
        hipMemcpy(dst, d, N, hipMemcpyDeviceToHost);
        double t1 = now();
        printf("D2H into faulted pageable        %7.1f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
        t0 = now();
        hipMemcpy(d, src, N, hipMemcpyHostToDevice);
        t1 = now();
        printf("H2D from faulted pageable        %7.1f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
        t0 = now();
        hipError_t e = hipHostRegister(dst, N, hipHostRegisterDefault);
        t1 = now();
        printf("hipHostRegister(faulted)  %s  %7.1f ms  %6.1f GB/s\n", hipGetErrorString(e), (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
        if (e == hipSuccess) {
            t0 = now();
            hipMemcpy(dst, d, N, hipMemcpyDeviceToHost);
            t1 = now();
            printf("D2H into registered              %7.1f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
            t0 = now();
            hipMemcpy(d, dst, N, hipMemcpyHostToDevice);
            t1 = now();
            printf("H2D from registered              %7.1f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
            t0 = now();
            hipHostUnregister(dst);
            t1 = now();
            printf("hipHostUnregister                %7.1f ms\n", (t1 - t0) * 1e3);
        }
        char *fr = fresh(N);
        t0 = now();
        e = hipHostRegister(fr, N, hipHostRegisterDefault);
        t1 = now();
        printf("hipHostRegister(fresh)    %s  %7.1f ms  %6.1f GB/s\n", hipGetErrorString(e), (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
        if (e == hipSuccess) hipHostUnregister(fr);
        munmap(fr, N);
        void *pin = nullptr;
        t0 = now();
        e = hipHostMalloc(&pin, N, hipHostMallocDefault);
        t1 = now();
        printf("hipHostMalloc             %s  %7.1f ms  %6.1f GB/s\n", hipGetErrorString(e), (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
        if (e == hipSuccess) {
            for (int rep = 0; rep < 2; ++rep) {
                t0 = now();
                hipMemcpy(pin, d, N, hipMemcpyDeviceToHost);
                t1 = now();
                printf("D2H into hipHostMalloc           %7.1f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
            }
            t0 = now();
            hipMemcpy(d, pin, N, hipMemcpyHostToDevice);
            t1 = now();
            printf("H2D from hipHostMalloc           %7.1f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
            // two streams, both directions at once
            hipStream_t s1, s2;
            hipStreamCreate(&s1); hipStreamCreate(&s2);
            void *d2 = nullptr; hipMalloc(&d2, N / 2);
            t0 = now();
            hipMemcpyAsync(pin, d, N / 2, hipMemcpyDeviceToHost, s1);
            hipMemcpyAsync(d2, (char *)pin + N / 2, N / 2, hipMemcpyHostToDevice, s2);
            hipDeviceSynchronize();
            t1 = now();
            printf("D2H + H2D concurrently (N/2 each) %7.1f ms  %6.1f GB/s total\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
            t0 = now();
            hipHostFree(pin);
            t1 = now();
            printf("hipHostFree                      %7.1f ms\n", (t1 - t0) * 1e3);
        }
        t0 = now();
        hipFree(d);
        t1 = now();
        printf("hipFree(3.5 GiB)                 %7.1f ms\n", (t1 - t0) * 1e3);
        t0 = now();
        hipMalloc(&d, N);
        t1 = now();
        printf("hipMalloc(3.5 GiB)               %7.1f ms\n", (t1 - t0) * 1e3);
    }
    return 0;
}
