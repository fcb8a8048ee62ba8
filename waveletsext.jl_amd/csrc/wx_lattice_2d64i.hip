// 64 x 64 Float64 images, full quad trees in one pass (wx_lattice_2d64.h): the inverse kernels
#include "wx_lattice_2d64.h"
int wx_lattice_2d64_inv_f64(const double *x, double *y, int L, int64_t batch, int64_t in_img, const WxFilt &filt, hipStream_t st)
{
    return wx_lattice_2d64_launch<double, 8, true>(x, y, L, batch, in_img, filt, st);
}
