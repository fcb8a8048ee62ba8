// wx_lattice_sg.h -- launcher body of the general interleaved lattice kernels for signals of 512, 256, 128 and 64 samples
// (k_lat_wpt_g_f64, k_lat_iwpt_g_f64 in wx_lattice_dev.h: 8 .. 64 signals per wavefront), four filter lengths per translation unit:
// NS0 .. NS0 + 3 rotation stages (wx_lattice_sg.hip: 2 .. 8 taps)
#pragma once
#include "wx_lattice_dev.h"

// 0 = not applicable, 1 = launched, < 0 = error
template <int NS0>
static int wx_lattice_launch_g_T(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, int64_t in_stride,
                                 const WxFilt &filt, hipStream_t st)
{
    int SH = 0;
    while (((int64_t)4096 >> SH) > n) ++SH;
    if (SH < 3 || SH > 6 || ((int64_t)4096 >> SH) != n) return 0;
    const int64_t per = (int64_t)1 << SH;
    if (L < 1 || L + SH < 6 || L + SH > 12 || (filt.F & 1) || filt.F < 2 * NS0 || filt.F > 2 * (NS0 + 3) || batch < per || batch > 0x7fffffff) return 0;
    if ((batch & (per - 1)) && x == y) return 0;             // the tail wavefront re-does signals: out of place only
    if (in_stride < n || in_stride * (per - 1) + 4096 > 0x7fffffff || (in_stride & 1)) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, inverse, &cw.c)) return 0;
    for (int l = 0; l <= 12; ++l) cw.gl[l] = 0.0;
    cw.gl[L] = cw.c.g0;
    cw.gl[0] = 1.0;
    cw.tail_bsig = 0;
    const int64_t nwave = (batch + per - 1) / per;
    const int last_sig = (int)(batch - per);
    const unsigned is32 = (unsigned)in_stride;
#define WX_GOG(NSS, SHH)                                                                                             \
    if (wx_lat_built(NSS) && wx_lat_stages(filt.F) == NSS && SH == SHH) {                                                                            \
        if (inverse)                                                                                                 \
            hipLaunchKernelGGL((k_lat_iwpt_g_f64<NSS, 2, SHH>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, is32, cw); \
        else                                                                                                         \
            hipLaunchKernelGGL((k_lat_wpt_g_f64<NSS, 2, SHH>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, cw); \
    }
#define WX_GOG4(SHH) WX_GOG(NS0, SHH) WX_GOG(NS0 + 1, SHH) WX_GOG(NS0 + 2, SHH) WX_GOG(NS0 + 3, SHH)
    WX_GOG4(3) WX_GOG4(4) WX_GOG4(5) WX_GOG4(6)
#undef WX_GOG4
#undef WX_GOG
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpt launch (short signals)", __FILE__, __LINE__);
    return 1;
}
