// wx_lattice_sgw_b.hip -- the interleaved lattice wpd kernel for signals of 512 ... 64 samples, filters of 10 ... 16 taps (wx_lattice_sgw.h)
#include "wx_lattice_sgw.h"

int wx_lattice_wpd_g_b_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    return wx_lattice_wpd_g_T<5>(x, y, n, L, batch, filt, st);
}

int wx_lattice_wpd_g_b_f32(const float *x, float *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    return wx_lattice_wpd_g32_T<5>(x, y, n, L, batch, filt, st);
}
