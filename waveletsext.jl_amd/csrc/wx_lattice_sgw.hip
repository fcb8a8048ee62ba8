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
int wx_lattice_wpd_g_b_f32(const float *x, float *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st);
int wx_lattice_wpd_g_f32(const float *x, float *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    if (filt.F > 8) return wx_lattice_wpd_g_b_f32(x, y, n, L, batch, filt, st);
    return wx_lattice_wpd_g32_T<1>(x, y, n, L, batch, filt, st);
}
