// prints what lane i receives for the DPP controls used by wx_lattice.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __global__ void k(int *out) {
    int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(0, lane, CTRL, 0xF, 0xF, true);
}
template <int CTRL> void run(const char *name) {
    int *d; hipMalloc(&d, 64 * 4); int h[64];
    hipLaunchKernelGGL(k<CTRL>, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%s:", name); for (int i = 0; i < 64; ++i) printf(" %d", h[i]); printf("\n");
    hipFree(d);
}
int main() {
    run<0x134>("wave_rol1 0x134"); run<0x13C>("wave_ror1 0x13C"); run<0x12F>("row_ror15 0x12F"); run<0x121>("row_ror1 0x121");
    run<0x130>("wave_shl1 0x130"); run<0x138>("wave_shr1 0x138");
    return 0;
}
