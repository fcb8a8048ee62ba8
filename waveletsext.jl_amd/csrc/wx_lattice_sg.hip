// wx_lattice_sg.hip -- the interleaved lattice kernels for signals of 512 ... 64 samples, filters of 2 ... 8 taps (wx_lattice_sg.h);
// longer filters: wx_lattice_sg_b.hip
#include "wx_lattice_sg.h"

int wx_lattice_launch_g_b(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, int64_t in_stride, const WxFilt &filt, hipStream_t st);

int wx_lattice_launch_g(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, int64_t in_stride,
                        const WxFilt &filt, hipStream_t st)
{
    if (filt.F > 8) return wx_lattice_launch_g_b(inverse, x, y, n, L, batch, in_stride, filt, st);
    return wx_lattice_launch_g_T<1>(inverse, x, y, n, L, batch, in_stride, filt, st);
}
