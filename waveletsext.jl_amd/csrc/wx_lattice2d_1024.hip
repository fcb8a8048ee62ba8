// the kernels of the 1024 x 1024 geometry (depth 7, eight columns per wavefront) of wx_lattice2d.h
#include "wx_lattice2d.h"

int wx_lattice2d_launch_1024(const float *src, float *dst, int64_t m, int L, int64_t batch, const WxFilt &filt, bool inverse, int pass, hipStream_t st)
{
    return wx_lattice2d_launch<2>(src, dst, m, L, batch, filt, inverse, pass, st);
}
int wx_lattice2d_fused_1024(const float *src, float *dst, float *ring, unsigned *ctl, int64_t m, int L, int64_t batch, const WxFilt &filt, bool inverse,
                          hipStream_t st)
{
    return wx_lattice2d_fused_launch<2>(src, dst, ring, ctl, m, L, batch, filt, inverse, st);
}
int64_t wx_lattice2d_ring_elems_1024(int64_t batch) { return wx_lattice2d_ring_elems_t<2>(batch); }
