// wx_lattice2d.hip -- 2-D full-tree packets of Float32 images on the transposing lattice kernel (wx_lattice2d.h): dispatch, and
// the kernels of the 512 x 512 geometry (depth 6).  Reference: 2-D wpt / iwpt by level, DWT.jl:500-548, 662-710.
#include "wx_lattice2d.h"

int wx_lattice2d_launch_256(const float *, float *, int64_t, int64_t, const WxFilt &, bool, int, hipStream_t);    // wx_lattice2d_256.hip
int wx_lattice2d_launch_1024(const float *, float *, int64_t, int64_t, const WxFilt &, bool, int, hipStream_t);   // wx_lattice2d_1024.hip

bool wx_lattice2d_ok(int64_t m, int64_t n, int L, const WxFilt &filt, size_t esz)
{
    static const bool off = getenv("WX_LATTICE2D") && atoi(getenv("WX_LATTICE2D")) == 0;
    const bool shape = (m == 512 && n == 512 && L == 6) || (m == 256 && n == 256 && L == 5) || (m == 1024 && n == 1024 && L == 7);
    return !off && esz == 4 && shape && filt.F >= 2 && (filt.F & 1) == 0 && filt.F / 2 <= WX_L2_MAXS;
}

// one transposing pass over `batch` images of side `m` (512: depth 6, 256: depth 5, 1024: depth 7): 0 = not applicable,
// 1 = launched, < 0 = error.  pass 0: natural image in, transposed image out (what one application of the kernel is);
// pass 1 / 2: the first / second pass of a transform, with the intermediate image in the blocked layout of wx_lattice2d.h (the
// caller's scratch buffer, never seen outside the library).
int wx_lattice2d_colT_f32(const float *src, float *dst, int64_t m, int64_t batch, const WxFilt &filt, bool inverse, int pass, hipStream_t st)
{
    if (m == 512) return wx_lattice2d_launch<0>(src, dst, m, batch, filt, inverse, pass, st);
    if (m == 256) return wx_lattice2d_launch_256(src, dst, m, batch, filt, inverse, pass, st);
    if (m == 1024) return wx_lattice2d_launch_1024(src, dst, m, batch, filt, inverse, pass, st);
    return 0;
}
