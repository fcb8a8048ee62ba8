// 64 x 64 Float64 images, full quad trees in one pass (wx_lattice_2d64.h): the forward kernels
#include "wx_lattice_2d64.h"
int wx_lattice_2d64_fwd_f64(const double *x, double *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    return wx_lattice_2d64_launch<double, 8, false>(x, y, L, batch, 4096, filt, st);
}
