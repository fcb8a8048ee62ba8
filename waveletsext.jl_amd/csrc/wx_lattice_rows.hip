// wx_lattice_rows.hip -- which images take the row pass on the lattice kernels (wx_lattice_rows.h; the kernels are built per column count and
// direction in wx_lattice_rows_{3,4,5}{f,i}.hip).  Reference: the row half of the 2-D wpt / iwpt by level, DWT.jl:500-548, 662-710.
#include "wx_common.h"
#include "wx_kernels.h"
#include <cstdlib>

#define WX_DECL_ROWS(SD)                                                                                                             \
    int wx_lattice_rows_##SD##_f64(const double *, double *, int64_t, int64_t, int64_t, int, int64_t, const WxFilt &, hipStream_t);
WX_DECL_ROWS(3f) WX_DECL_ROWS(3i) WX_DECL_ROWS(4f) WX_DECL_ROWS(4i) WX_DECL_ROWS(5f) WX_DECL_ROWS(5i)
#undef WX_DECL_ROWS

// Built since round 6: Float64 images of 128, 256 and (from 6 levels on) 512 columns.  1024 columns (0.20 -> 0.21 of the roofline, deep trees only)
// and Float32 (128 columns 0.22 -> 0.28) were more than half of the family's compile time -- 34 of the library's 113 CPU-minutes -- and take the
// LDS strips of k_rows_fused again.
// A wavefront takes 2^SH rows: runs of 2^SH elements.  Measured per GiB of square Float64 images, column pass + row pass, db4, against the LDS
// strips of k_rows_fused (profiles/r05_floor2d_rows.txt): 128 columns 0.82 / 1.10 ms, 256 columns 0.82 / 1.05 ms at full depth (0.80 / 0.86 at
// depth 3); 512 columns (64-byte runs; the XCD-aware mapping brought them from 1.32 to 1.04 ms, workgroups of two wavefronts on adjacent row
// groups to 0.97) 0.97 / 1.13 ms at full depth but 0.91 / 0.85 at depth 3: taken from 6 levels on; 1024 columns (32-byte runs, workgroups of
// four wavefronts: 1.71 -> 1.27 ms) 1.27 / 1.34 ms at full depth, 0.95 / 0.93 at depth 3: taken from 8 levels on.  Float32: 128 columns 0.96 / 1.22 ms
// at full depth, no gain at depth 3 (the launcher refuses fewer than 5 levels); 256 ... 1024 columns at depth >= 5 ... 7 belong to wx_lattice2d.h.
// Knob WX_LATROWS_MINSH (diagnostics): the smallest SH taken at every depth (6 = none).
static int wx_rows_minsh(int L)
{
    static const int env = wx_getenv("WX_LATROWS_MINSH") ? atoi(wx_getenv("WX_LATROWS_MINSH")) : -1;
    return env >= 0 ? env : (L >= 6 ? 3 : 4);
}

int wx_lattice_rows(bool inverse, const double *x, double *y, int64_t in_img, int64_t out_img, int64_t m, int64_t n, int L, int64_t batch,
                    const WxFilt &filt, hipStream_t st)
{
    const int SH = n == 512 ? 3 : (n == 256 ? 4 : (n == 128 ? 5 : -1));
    if (SH < 0 || SH < wx_rows_minsh(L)) return 0;
    if (SH == 3) return inverse ? wx_lattice_rows_3i_f64(x, y, in_img, out_img, m, L, batch, filt, st) : wx_lattice_rows_3f_f64(x, y, in_img, out_img, m, L, batch, filt, st);
    if (SH == 4) return inverse ? wx_lattice_rows_4i_f64(x, y, in_img, out_img, m, L, batch, filt, st) : wx_lattice_rows_4f_f64(x, y, in_img, out_img, m, L, batch, filt, st);
    return inverse ? wx_lattice_rows_5i_f64(x, y, in_img, out_img, m, L, batch, filt, st) : wx_lattice_rows_5f_f64(x, y, in_img, out_img, m, L, batch, filt, st);
}
int wx_lattice_rows(bool, const float *, float *, int64_t, int64_t, int64_t, int64_t, int, int64_t, const WxFilt &, hipStream_t)
{
    return 0;                                                 // not built since round 6 (see above): the LDS strips
}
