// the kernels of the 256 x 256 geometry (depth 5, two images per register column) of wx_lattice2d.h
#include "wx_lattice2d.h"

int wx_lattice2d_launch_256(const float *src, float *dst, int64_t m, int L, int64_t batch, const WxFilt &filt, bool inverse, int pass, hipStream_t st)
{
    return wx_lattice2d_launch<1>(src, dst, m, L, batch, filt, inverse, pass, st);
}
int wx_lattice2d_fused_256(const float *src, float *dst, float *ring, unsigned *ctl, int64_t m, int L, int64_t batch, const WxFilt &filt, bool inverse,
                          hipStream_t st)
{
    return wx_lattice2d_fused_launch<1>(src, dst, ring, ctl, m, L, batch, filt, inverse, st);
}
int64_t wx_lattice2d_ring_elems_256(int64_t batch) { return wx_lattice2d_ring_elems_t<1>(batch); }
