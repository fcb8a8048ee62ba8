// wx_lattice_sg.hip -- the interleaved lattice kernels for signals of 512 ... 64 samples, filters of 2 ... 8 taps (wx_lattice_sg.h).
// Full trees with 10 ... 16 taps on these lengths go through the masked tree kernels as a tree of ones (wx_lattice_tree_s.h: the same speed --
// 256 samples, db8, depth 8: 0.55 / 0.52 against 0.56 / 0.53 ms per GiB -- so a second set of these kernels is not built)
#include "wx_lattice_sg.h"

int wx_lattice_launch_g(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, int64_t in_stride,
                        const WxFilt &filt, hipStream_t st)
{
    if (filt.F > 8) return 0;
    return wx_lattice_launch_g_T<1>(inverse, x, y, n, L, batch, in_stride, filt, st);
}
