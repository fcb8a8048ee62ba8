// 64 x 64 images along any quad tree in one pass (wx_lattice_2d64t.h): the inverse kernels, Float64 and Float32 (two images per wavefront)
#include "wx_lattice_2d64t.h"
int wx_lattice_2d64t_inv_f64(const double *x, double *y, int L, int64_t batch, int64_t in_img, const WxFilt &filt, const uint8_t *dstatus, int64_t nstatus,
                             hipStream_t st)
{
    return wx_lattice_2d64t_launch<double, 8, true>(x, y, L, batch, in_img, filt, dstatus, nstatus, st);
}
int wx_lattice_2d64t_inv_f32(const float *x, float *y, int L, int64_t batch, int64_t in_img, const WxFilt &filt, const uint8_t *dstatus, int64_t nstatus,
                             hipStream_t st)
{
    return wx_lattice_2d64t_launch<float, 8, true>(x, y, L, batch, in_img, filt, dstatus, nstatus, st);
}
