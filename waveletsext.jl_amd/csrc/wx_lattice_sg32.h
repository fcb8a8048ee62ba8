// wx_lattice_sg32.h -- launcher of the interleaved lattice kernels (k_lat_wpt_g_f64 / k_lat_iwpt_g_f64, wx_lattice_dev.h) for Float32
// signals of 2048 ... 64 samples, full trees: 2^SH signals per wavefront, Float32 at the two ends (the loads widen, the stores round
// once), the rotations in Float64.  Reference: Wavelets.jl's wpt / iwpt with a level on an AbstractArray{T} (the reference is generic
// in T: dwt/dwt_one_level.jl:79-83) as called by wptall / iwptall (dwt/dwt_all.jl:152-166, 210-225), and the column pass of the 2-D
// transforms (DWT.jl:500-548).
// Included by wx_lattice_sg32_{1..6}.hip with WX_G32_SH = 1 .. 6 and WX_G32_FN = the launcher's name.
#include "wx_lattice_dev.h"

// 0 = not applicable (the caller goes on to the tree-driven or the fused LDS kernels), 1 = launched, < 0 = error
int WX_G32_FN(bool inverse, const float *x, float *y, int64_t n, int L, int64_t batch, int64_t in_stride, const WxFilt &filt, hipStream_t st)
{
    constexpr int SH = WX_G32_SH;
    constexpr int64_t per = (int64_t)1 << SH;
    if (n != (4096 >> SH) || L < 1 || L + SH < 6 || L + SH > 12 || filt.F < 2 || filt.F > 8 || batch < per || batch > 0x7fffffff) return 0;
    if ((batch & (per - 1)) && x == y) return 0;             // the tail wavefront re-does signals: out of place only
    if (in_stride < n || (in_stride & 3) || in_stride * (per - 1) + 4096 > 0x7fffffff) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, inverse, &cw.c)) return 0;
    for (int l = 0; l <= 12; ++l) cw.gl[l] = 0.0;
    cw.gl[L] = cw.c.g0;
    cw.gl[0] = 1.0;
    const unsigned nwave = (unsigned)((batch + per - 1) / per);
    const int last_sig = (int)(batch - per);
#define WX_GOG(NSS)                                                                                                                  \
    case NSS:                                                                                                                        \
        if (inverse)                                                                                                                 \
            hipLaunchKernelGGL((k_lat_iwpt_g_f64<NSS, 2, SH, float>), dim3(nwave), dim3(64), 0, st, x, y, L, last_sig, (unsigned)in_stride, cw); \
        else                                                                                                                         \
            hipLaunchKernelGGL((k_lat_wpt_g_f64<NSS, 2, SH, float>), dim3(nwave), dim3(64), 0, st, x, y, L, last_sig, cw);           \
        break;
    switch (filt.F / 2) {
        WX_GOG(1) WX_GOG(2) WX_GOG(3) WX_GOG(4)
    default: return 0;
    }
#undef WX_GOG
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpt launch (short Float32 signals)", __FILE__, __LINE__);
    return 1;
}
