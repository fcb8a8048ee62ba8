// 64 x 64 images along any quad tree in one pass (wx_lattice_2d64t.h): the forward kernels, Float64 and Float32 (two images per wavefront)
#include "wx_lattice_2d64t.h"
int wx_lattice_2d64t_fwd_f64(const double *x, double *y, int L, int64_t batch, const WxFilt &filt, const uint8_t *dstatus, int64_t nstatus, hipStream_t st)
{
    return wx_lattice_2d64t_launch<double, 8, false>(x, y, L, batch, 4096, filt, dstatus, nstatus, st);
}
int wx_lattice_2d64t_fwd_f32(const float *x, float *y, int L, int64_t batch, const WxFilt &filt, const uint8_t *dstatus, int64_t nstatus, hipStream_t st)
{
    return wx_lattice_2d64t_launch<float, 8, false>(x, y, L, batch, 4096, filt, dstatus, nstatus, st);
}
