// wx_lattice_sh.hip -- launcher of the lattice kernels for 2048- and 1024-sample signals (2 / 4 signals interleaved in one
// wavefront: k_lat_wpt_sh_f64, k_lat_iwpt_sh_f64 in wx_lattice_dev.h)
#include "wx_lattice_dev.h"

// 2^SH signals of 4096 >> SH samples per wavefront (forward wpt)
int wx_lattice_launch_sh(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, int64_t in_stride,
                         const WxFilt &filt, hipStream_t st)
{
    const int SH = n == 2048 ? 1 : 2;
    const int64_t per = (int64_t)1 << SH;
    if (L + SH < 6 || L + SH > 12 || filt.F < 2 || batch < per) return 0;
    if ((batch & (per - 1)) && x == y) return 0;             // the tail wavefront re-does signals: out of place only
    if (in_stride < n || in_stride * (per - 1) + 4096 > 0x7fffffff || (in_stride & 1)) return 0;
    const unsigned is32 = (unsigned)in_stride;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, inverse, &cw.c)) return 0;
    for (int l = 0; l <= 12; ++l) cw.gl[l] = 0.0;
    cw.gl[L] = cw.c.g0;                                       // only the leaves are written / read
    cw.gl[0] = 1.0;                                           // lat_emit<0, 16 SH>: plain transposition
    const int64_t nwave = (batch + per - 1) / per;
    if (batch > 0x7fffffff) return 0;
    const int last_sig = (int)(batch - per);
#define WX_GOS(NSS)                                                                                                  \
    case NSS:                                                                                                        \
        if (inverse && SH == 1)                                                                                      \
            hipLaunchKernelGGL((k_lat_iwpt_sh_f64<NSS, 2, 1>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, is32, cw); \
        else if (inverse)                                                                                            \
            hipLaunchKernelGGL((k_lat_iwpt_sh_f64<NSS, 2, 2>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, is32, cw); \
        else if (SH == 1)                                                                                            \
            hipLaunchKernelGGL((k_lat_wpt_sh_f64<NSS, 2, 1>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, cw); \
        else                                                                                                         \
            hipLaunchKernelGGL((k_lat_wpt_sh_f64<NSS, 2, 2>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, cw); \
        break;
    switch (wx_lat_stages(filt.F)) {
        WX_GOS(1) WX_GOS(2) WX_GOS(4) WX_GOS(6) WX_GOS(8) WX_GOS(10)
    default: return 0;
    }
#undef WX_GOS
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpt launch (interleaved signals)", __FILE__, __LINE__);
    return 1;
}

