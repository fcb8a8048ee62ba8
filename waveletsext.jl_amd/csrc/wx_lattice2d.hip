// wx_lattice2d.hip -- 2-D full-tree packets of Float32 images on the transposing lattice kernel (wx_lattice2d.h): dispatch, and
// the kernels of the 512 x 512 geometry (depth 6).  Reference: 2-D wpt / iwpt by level, DWT.jl:500-548, 662-710.
#include "wx_lattice2d.h"

int wx_lattice2d_launch_256(const float *, float *, int64_t, int, int64_t, const WxFilt &, bool, int, hipStream_t);    // wx_lattice2d_256.hip
int wx_lattice2d_launch_1024(const float *, float *, int64_t, int, int64_t, const WxFilt &, bool, int, hipStream_t);   // wx_lattice2d_1024.hip

int wx_lattice2d_fused_256(const float *, float *, float *, unsigned *, int64_t, int, int64_t, const WxFilt &, bool, hipStream_t);
int wx_lattice2d_fused_1024(const float *, float *, float *, unsigned *, int64_t, int, int64_t, const WxFilt &, bool, hipStream_t);
int64_t wx_lattice2d_ring_elems_256(int64_t batch);
int64_t wx_lattice2d_ring_elems_1024(int64_t batch);

bool wx_lattice2d_ok(int64_t m, int64_t n, int L, const WxFilt &filt, size_t esz)
{
    static const bool off = wx_getenv("WX_LATTICE2D") && atoi(wx_getenv("WX_LATTICE2D")) == 0;
    // the lattice levels (6 / 5 / 7: down to nodes of 8 x 8) and, since round 4, up to three more levels as one 8 x 8 matrix per node in the
    // store of the forward pass / the load of the inverse pass: depths 6 .. 9, 5 .. 8, 7 .. 10 -- the full depth, which is the
    // reference's default L, included
    static const bool deep = !(wx_getenv("WX_LATTICE2D_DEEP") && atoi(wx_getenv("WX_LATTICE2D_DEEP")) == 0);
    const int ex = deep ? 3 : 0;
    const bool shape = (m == 512 && n == 512 && L >= 6 && L <= 6 + ex) || (m == 256 && n == 256 && L >= 5 && L <= 5 + ex) ||
                       (m == 1024 && n == 1024 && L >= 7 && L <= 7 + ex);
    return !off && esz == 4 && shape && filt.F >= 2 && (filt.F & 1) == 0 && filt.F / 2 <= WX_L2_MAXS;
}

// one transposing pass over `batch` images of side `m` (512: depth 6, 256: depth 5, 1024: depth 7): 0 = not applicable,
// 1 = launched, < 0 = error.  pass 0: natural image in, transposed image out (what one application of the kernel is);
// pass 1 / 2: the first / second pass of a transform, with the intermediate image in the blocked layout of wx_lattice2d.h (the
// caller's scratch buffer, never seen outside the library).
int wx_lattice2d_colT_f32(const float *src, float *dst, int64_t m, int L, int64_t batch, const WxFilt &filt, bool inverse, int pass, hipStream_t st)
{
    if (m == 512) return wx_lattice2d_launch<0>(src, dst, m, L, batch, filt, inverse, pass, st);
    if (m == 256) return wx_lattice2d_launch_256(src, dst, m, L, batch, filt, inverse, pass, st);
    if (m == 1024) return wx_lattice2d_launch_1024(src, dst, m, L, batch, filt, inverse, pass, st);
    return 0;
}

// both passes of a transform in one persistent launch, the intermediate image in a ring of wx_lattice2d_ring_elems(m, batch) elements that
// stays in the Infinity Cache (k_lat2d_fused_f32 in wx_lattice2d.h); ctl: wx_lattice2d_ctl_bytes() bytes of scratch.  0 = not applicable,
// 1 = launched, < 0 = error
int wx_lattice2d_fused_f32(const float *src, float *dst, float *ring, unsigned *ctl, int64_t m, int L, int64_t batch, const WxFilt &filt, bool inverse,
                           hipStream_t st)
{
    if (m == 512) return wx_lattice2d_fused_launch<0>(src, dst, ring, ctl, m, L, batch, filt, inverse, st);
    if (m == 256) return wx_lattice2d_fused_256(src, dst, ring, ctl, m, L, batch, filt, inverse, st);
    if (m == 1024) return wx_lattice2d_fused_1024(src, dst, ring, ctl, m, L, batch, filt, inverse, st);
    return 0;
}
int64_t wx_lattice2d_ring_elems(int64_t m, int64_t batch)
{
    if (m == 512) return wx_lattice2d_ring_elems_t<0>(batch);
    if (m == 256) return wx_lattice2d_ring_elems_256(batch);
    if (m == 1024) return wx_lattice2d_ring_elems_1024(batch);
    return 0;
}
size_t wx_lattice2d_ctl_bytes() { return WX_L2F_CTL_BYTES; }
bool wx_lattice2d_fused_on()
{
    static const bool off = wx_getenv("WX_L2D_FUSED") && atoi(wx_getenv("WX_L2D_FUSED")) == 0;
    return !off;
}
