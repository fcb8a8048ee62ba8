// wx_lattice_shw.hip -- launcher of the lattice wpd kernel for 2048- and 1024-sample signals (k_lat_wpd_sh_f64 in
// wx_lattice_dev.h; its own translation unit so that the instantiations compile beside the others)
#include "wx_lattice_dev.h"

// 0 = not applicable, 1 = launched, < 0 = error
int wx_lattice_wpd_sh_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    const int SH = n == 2048 ? 1 : 2;
    const int64_t per = (int64_t)1 << SH;
    if (L < 1 || L + SH > 12 || filt.F < 2 || batch < per || batch > 0x7fffffff) return 0;
    if (SH == 2 && L < 1) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    if ((n * (L + 1)) * (per - 1) + 4096 > 0x7fffffff) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, false, &cw.c)) return 0;
    {
        WxLat tmp;
        if (!wx_lattice_factor(filt, 1, false, &tmp)) return 0;
        const long double g = tmp.g0;
        long double acc = 1;
        for (int l = 0; l <= 12; ++l) { cw.gl[l] = (double)acc; acc *= g; }
    }
    const int64_t nwave = (batch + per - 1) / per;
    const int last_sig = (int)(batch - per);
#define WX_GOSW(NSS)                                                                                                 \
    case NSS:                                                                                                        \
        if (SH == 1)                                                                                                 \
            hipLaunchKernelGGL((k_lat_wpd_sh_f64<NSS, 2, 1>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, cw); \
        else                                                                                                         \
            hipLaunchKernelGGL((k_lat_wpd_sh_f64<NSS, 2, 2>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, cw); \
        break;
    switch (wx_lat_stages(filt.F)) {
        WX_GOSW(1) WX_GOSW(2) WX_GOSW(4) WX_GOSW(6) WX_GOSW(8) WX_GOSW(10)
    default: return 0;
    }
#undef WX_GOSW
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpd launch (interleaved signals)", __FILE__, __LINE__);
    return 1;
}
