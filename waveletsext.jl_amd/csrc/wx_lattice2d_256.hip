// the kernels of the 256 x 256 geometry (depth 5, two images per register column) of wx_lattice2d.h
#include "wx_lattice2d.h"

int wx_lattice2d_launch_256(const float *src, float *dst, int64_t m, int L, int64_t batch, const WxFilt &filt, bool inverse, int pass, hipStream_t st)
{
    return wx_lattice2d_launch<1>(src, dst, m, L, batch, filt, inverse, pass, st);
}
