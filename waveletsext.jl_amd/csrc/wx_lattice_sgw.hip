// wx_lattice_sgw.hip -- the interleaved lattice wpd kernel for signals of 512 ... 64 samples, filters of 2 ... 8 taps (wx_lattice_sgw.h;
// 10 ... 16 taps: wx_lattice_sgw_b.hip), and the Float32 form for 64 / 128 samples
#include "wx_lattice_sgw.h"

int wx_lattice_wpd_g_b_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st);

int wx_lattice_wpd_g_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    if (filt.F > 8) return wx_lattice_wpd_g_b_f64(x, y, n, L, batch, filt, st);
    return wx_lattice_wpd_g_T<1>(x, y, n, L, batch, filt, st);
}

// Float32 signals of 256, 128 and 64 samples (round 5: the fused LDS kernel ran them at 0.47 / 0.35 / 0.21 of the HBM roofline of the table's
// bytes, the Float64 kernels above at 0.62): the same kernels with Float32 at the two ends
int wx_lattice_wpd_g_f32(const float *x, float *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    int SH = 0;
    while (((int64_t)4096 >> SH) > n) ++SH;
    if (SH < 4 || SH > 6 || ((int64_t)4096 >> SH) != n) return 0;
    const int64_t per = (int64_t)1 << SH;
    if (L < 1 || L + SH > 12 || filt.F < 2 || filt.F > 8 || (filt.F & 1) || batch < per || batch > 0x7fffffff || x == y) return 0;
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
    if (filt.F / 2 == NSS && SH == SHH)                                                                              \
        hipLaunchKernelGGL((k_lat_wpd_g_f64<NSS, 2, SHH, float>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, cw);
    WX_GOGW(1, 4) WX_GOGW(2, 4) WX_GOGW(3, 4) WX_GOGW(4, 4)
    WX_GOGW(1, 5) WX_GOGW(2, 5) WX_GOGW(3, 5) WX_GOGW(4, 5)
    WX_GOGW(1, 6) WX_GOGW(2, 6) WX_GOGW(3, 6) WX_GOGW(4, 6)
#undef WX_GOGW
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpd launch (short Float32 signals)", __FILE__, __LINE__);
    return 1;
}
