// wx_lattice_tree_s.h -- launcher of the tree-driven lattice kernels of wx_lattice_tree_sc.h for SHORT signals: 512, 256, 128, 64 samples
// (SH = 3 .. 6: 8 .. 64 signals per wavefront), Float64 and Float32 (Float64 registers: the loads widen, the stores round once; 128 and 256
// samples: Float32 arithmetic on pairs of signals), filters up to 16 taps, wpt / iwpt
// along any tree -- pyramids (dwt / idwt), best bases, random trees -- and iwpd by tree.  Reference: Wavelets.jl's wpt / iwpt with a
// tree::BitVector as called by wptall / iwptall (dwt/dwt_all.jl:152-166, 210-225), dwtall / idwtall (dwt/dwt_all.jl:39-110: the tree of
// maketree(:dwt)), iwpd (DWT.jl:340-351).
// Until round 5 these signals took the kernels that keep a signal in LDS or in one lane (wx_smalltree.hip, wx_lanetree.h: 0.22-0.47 of the
// HBM roofline against 0.65-0.70 for the full trees of the same lengths, profiles/r05_floor.txt).
// Included by wx_lattice_trees_{3..6}{f,i}.hip with WX_LAT_TREES_SH, WX_LAT_TREES_INV and WX_LAT_TREES_FN(type suffix).
#include "wx_lattice_dev.h"
#include "wx_host.h"
#include "wx_lattice_tree_sc.h"

// 0 = not applicable (the caller goes on), 1 = launched, < 0 = error.  inverse: leaves of signal b, depth l at x + b in_stride + l col_stride
// (col_stride = 0: dense leaves, n: a packet table); signal b of the output at y + b out_stride
template <typename IO, int SH, bool INV, int NSMAX, bool FP32A = false>
static int wx_lattice_trees_launch(const IO *x, IO *y, int64_t n, int L, int64_t batch, int64_t in_stride, int64_t col_stride, const WxFilt &filt,
                                   const uint8_t *dstatus, int64_t nstatus, hipStream_t st, int64_t out_stride)
{
    constexpr int64_t per = (int64_t)1 << SH;
    if (n != (4096 >> SH) || L < 1 || L + SH > 12 || filt.F < 2 || (filt.F & 1) || filt.F > 2 * NSMAX || batch < per || batch > 0x7fffffff || !dstatus) return 0;
    WxPairPlan pp;
    if (FP32A) {
        if (!wx_lat_pair_plan(batch, SH, x == y, &pp)) return 0;
    } else {
        pp.nwave = (unsigned)((batch + per - 1) / per); pp.tail_sig = (int)(batch - per); pp.tail_bsig = 0;
    }
    if (!FP32A && (batch & (per - 1)) && x == y) return 0;   // the tail wavefront re-does signals: out of place only
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    if (in_stride < n || (in_stride & 3) || (col_stride & 3) || in_stride * (2 * per - 1) + 13 * col_stride + 4096 > 0x7fffffff) return 0;
    const int64_t ostr = out_stride ? out_stride : n;
    if (ostr < n || (ostr & 3) || ostr * (2 * per - 1) + 4096 > 0x7fffffff) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, INV, &cw.c)) return 0;
    {
        WxLat one;
        if (!wx_lattice_factor(filt, 1, false, &one)) return 0;
        const long double g = one.g0;                           // product of the cosines of one level
        long double acc = 1;
        for (int l = 0; l <= 12; ++l) { cw.gl[l] = (double)acc; acc *= INV ? 1 / g : g; }
    }
    cw.tail_bsig = pp.tail_bsig;
    WxScratch scr(st);
    WxLatTreeSc *tsc = (WxLatTreeSc *)scr.alloc(sizeof(WxLatTreeSc));
    if (!tsc) return WX_EHIP;
    if (hipMemsetAsync(tsc->dep, 0, sizeof(tsc->dep), st) != hipSuccess) return wx_set_error(WX_EHIP, "lattice tree tables");
    hipLaunchKernelGGL((k_lat_treesc_prep<SH>), dim3(8), dim3(256), 0, st, dstatus, nstatus, L, tsc);
    hipLaunchKernelGGL(k_lat_treesc_prep2, dim3(1), dim3(64), 0, st, tsc, SH);
    const WxLatTreeSc *ctsc = tsc;
    const unsigned nw = pp.nwave;
    const int lsig = pp.tail_sig;
    WxThreshArg ta{nullptr, 0, 0, 0, 1.0};
    (void)ta;
#define WX_GOS(NSS)                                                                                                                  \
    case NSS:                                                                                                                        \
        if constexpr (INV)                                                                                                           \
            hipLaunchKernelGGL((k_lat_iwpt_treesc_f64<NSS, 2, SH, false, IO, FP32A>), dim3(nw), dim3(64), 0, st, x, y, L, lsig,      \
                               (unsigned)in_stride, (unsigned)col_stride, (unsigned)ostr, cw, ctsc, ta);                             \
        else                                                                                                                         \
            hipLaunchKernelGGL((k_lat_wpt_treesc_f64<NSS, 2, SH, IO, FP32A>), dim3(nw), dim3(64), 0, st, x, y, L, lsig,              \
                               (unsigned)in_stride, (unsigned)ostr, cw, ctsc);                                                       \
        break;
    switch (wx_lat_stages(filt.F)) {
        WX_GOS(1) WX_GOS(2) WX_GOS(4)
    default:
        if constexpr (NSMAX > 4) {
            switch (wx_lat_stages(filt.F)) {
                WX_GOS(6) WX_GOS(8)
            default: return 0;
            }
        } else
            return 0;
    }
#undef WX_GOS
    const hipError_t es = hipGetLastError();
    if (es != hipSuccess) return wx_set_hip_error(es, "lattice tree launch (short signals)", __FILE__, __LINE__);
    return 1;
}

#ifdef WX_LAT_TREES_SH
int WX_LAT_TREES_FN(f64)(const double *x, double *y, int64_t n, int L, int64_t batch, int64_t in_stride, int64_t col_stride, const WxFilt &filt,
                         const uint8_t *dstatus, int64_t nstatus, hipStream_t st, int64_t out_stride)
{
    return wx_lattice_trees_launch<double, WX_LAT_TREES_SH, WX_LAT_TREES_INV, 8>(x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, st, out_stride);
}
int WX_LAT_TREES_FN(f32)(const float *x, float *y, int64_t n, int L, int64_t batch, int64_t in_stride, int64_t col_stride, const WxFilt &filt,
                         const uint8_t *dstatus, int64_t nstatus, hipStream_t st, int64_t out_stride)
{
    // Float32 arithmetic on pairs of signals (lat_f2v, as wx_lattice_tree32.h does from 1024 samples up) where it measured faster: 128 and 256
    // samples, both directions (random trees 0.47-0.60 -> 0.43-0.45 ms per GiB, pyramids 0.45-0.53 -> 0.41-0.44); 64 samples (inverse 0.42 -> 0.52)
    // and 512 samples (inverse 0.44-0.52 -> 0.56) keep Float64 registers.  Filters up to 8 taps, dense leaves.  WX_TREES32_PAIRS = 0 / 1: never / always.
    static const int pk = wx_getenv("WX_TREES32_PAIRS") ? atoi(wx_getenv("WX_TREES32_PAIRS")) : -1;
    const bool pairs = pk >= 0 ? pk != 0 : (WX_LAT_TREES_SH == 4 || WX_LAT_TREES_SH == 5);
    if (pairs && col_stride == 0 && filt.F <= 8) {
        const int r = wx_lattice_trees_launch<float, WX_LAT_TREES_SH, WX_LAT_TREES_INV, 4, true>(x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, st, out_stride);
        if (r) return r;
    }
    return wx_lattice_trees_launch<float, WX_LAT_TREES_SH, WX_LAT_TREES_INV, 8>(x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, st, out_stride);
}
#endif
