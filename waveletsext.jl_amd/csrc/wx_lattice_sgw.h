// wx_lattice_sgw.h -- launcher body of the general interleaved lattice wpd kernel for signals of 512, 256, 128 and 64 samples
// (k_lat_wpd_g_f64 in wx_lattice_dev.h), four filter lengths per translation unit (wx_lattice_sgw.hip: 2 .. 8 taps, wx_lattice_sgw_b.hip: 10 .. 16)
#pragma once
#include "wx_lattice_dev.h"

template <int NS0>
static int wx_lattice_wpd_g_T(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    int SH = 0;
    while (((int64_t)4096 >> SH) > n) ++SH;
    if (SH < 3 || SH > 6 || ((int64_t)4096 >> SH) != n) return 0;
    const int64_t per = (int64_t)1 << SH;
    if (L < 1 || L + SH > 12 || (filt.F & 1) || filt.F < 2 * NS0 || filt.F > 2 * (NS0 + 3) || batch < per || batch > 0x7fffffff) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    if ((n * (L + 1)) * (per - 1) + 4096 > 0x7fffffff) return 0;
    if ((n * (L + 1)) & 1) return 0;
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
#define WX_GOGW(NSS, SHH)                                                                                            \
    if (wx_lat_built(NSS) && wx_lat_stages(filt.F) == NSS && SH == SHH)                                                                              \
        hipLaunchKernelGGL((k_lat_wpd_g_f64<NSS, 2, SHH>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, cw);
#define WX_GOGW4(SHH) WX_GOGW(NS0, SHH) WX_GOGW(NS0 + 1, SHH) WX_GOGW(NS0 + 2, SHH) WX_GOGW(NS0 + 3, SHH)
    WX_GOGW4(3) WX_GOGW4(4) WX_GOGW4(5) WX_GOGW4(6)
#undef WX_GOGW4
#undef WX_GOGW
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpd launch (short signals)", __FILE__, __LINE__);
    return 1;
}


// Float32 signals of 256, 128 and 64 samples: the same kernels with Float32 at the two ends (Float64 registers)
template <int NS0>
static int wx_lattice_wpd_g32_T(const float *x, float *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    int SH = 0;
    while (((int64_t)4096 >> SH) > n) ++SH;
    if (SH < 4 || SH > 6 || ((int64_t)4096 >> SH) != n) return 0;
    const int64_t per = (int64_t)1 << SH;
    if (L < 1 || L + SH > 12 || (filt.F & 1) || filt.F < 2 * NS0 || filt.F > 2 * (NS0 + 3) || batch < per || batch > 0x7fffffff || x == y) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    if ((n * (L + 1)) * (per - 1) + 4096 > 0x7fffffff) return 0;
    if ((n * (L + 1)) & 3) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, false, &cw.c)) return 0;
    {
        WxLat tmp;
        if (!wx_lattice_factor(filt, 1, false, &tmp)) return 0;
        const long double g = tmp.g0;
        long double acc = 1;
        for (int l = 0; l <= 12; ++l) { cw.gl[l] = (double)acc; acc *= g; }
    }
    cw.tail_bsig = 0;
    const int64_t nwave = (batch + per - 1) / per;
    const int last_sig = (int)(batch - per);
#define WX_GOGW(NSS, SHH)                                                                                            \
    if (wx_lat_built(NSS) && wx_lat_stages(filt.F) == NSS && SH == SHH)                                                                              \
        hipLaunchKernelGGL((k_lat_wpd_g_f64<NSS, 2, SHH, float>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, cw);
#define WX_GOGW4(SHH) WX_GOGW(NS0, SHH) WX_GOGW(NS0 + 1, SHH) WX_GOGW(NS0 + 2, SHH) WX_GOGW(NS0 + 3, SHH)
    WX_GOGW4(4) WX_GOGW4(5) WX_GOGW4(6)
#undef WX_GOGW4
#undef WX_GOGW
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpd launch (short Float32 signals)", __FILE__, __LINE__);
    return 1;
}

// (Float32 arithmetic on pairs of signals -- k_lat_wpd_g_f64<.., float, true> -- measured slower here: 64 samples 0.39 against 0.36 ms per 1 GiB
// table, 128 samples at depth 7 0.43 against 0.37: not instantiated)
