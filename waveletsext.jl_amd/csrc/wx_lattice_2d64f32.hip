// 64 x 64 Float32 images, full quad trees in one pass, two images per wavefront (wx_lattice_2d64.h): the forward kernels
#include "wx_lattice_2d64.h"
int wx_lattice_2d64_fwd_f32(const float *x, float *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    return wx_lattice_2d64_launch<float, 8, false>(x, y, L, batch, 4096, filt, st);
}
