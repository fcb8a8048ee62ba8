#ifndef WX_LAT_TREE_FIRST_FORM
#define WX_LAT_TREE_FIRST_FORM 0
#endif
// wx_lattice_tree.h -- launcher of the tree-driven lattice kernels (k_lat_wpt_tree_f64, k_lat_iwpt_tree_f64,
// k_lat_tree_prep in wx_lattice_dev.h): wpt / iwpt along a tree and iwpd by tree for Float64 signals of 4096, 2048 and
// 1024 samples.  Reference: Wavelets.jl's wpt / iwpt with a tree::BitVector as called by wptall / iwptall
// (dwt/dwt_all.jl:152-166, 210-225), iwpd (DWT.jl:340-351), LDB (LDB.jl:303, 409) and denoise (Denoising.jl:527).
#include "wx_lattice_dev.h"
#include "wx_host.h"
#include "wx_lattice_tree_sc.h"

// included by wx_lattice_tree{0,1,2}{f,i}.hip with WX_LAT_TREE_SH = 0, 1, 2 (signal length 4096 >> SH), WX_LAT_TREE_INV = 0 / 1
// (direction) and WX_LAT_TREE_FN = the launcher's name: one translation unit per length and direction so that the 60 kernels
// compile in parallel

// 0 = not applicable (the caller takes the fused LDS kernels), 1 = launched, < 0 = error.
// inverse: leaves of signal b, depth l at x + b in_stride + l col_stride (col_stride = 0: dense leaves, n: packet table);
// signal b of the output at y + b out_stride (0: n), of the forward's input at x + b in_stride
int WX_LAT_TREE_FN(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, int64_t in_stride, int64_t col_stride,
                   const WxFilt &filt, const uint8_t *dstatus, int64_t nstatus, const WxThreshArg *thr, hipStream_t st, int64_t out_stride)
{
    constexpr int SH = WX_LAT_TREE_SH;
    constexpr int64_t per = (int64_t)1 << SH;
    if (inverse != (WX_LAT_TREE_INV != 0)) return 0;
    if (n != (4096 >> SH) || L < 1 || L + SH > 12 || filt.F < 2 || batch < per || batch > 0x7fffffff || !dstatus) return 0;
    if ((batch & (per - 1)) && x == y) return 0;             // the tail wavefront re-does signals: out of place only
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    if (in_stride < n || (in_stride & 3) || (col_stride & 3) || in_stride * (per - 1) + 13 * col_stride + 4096 > 0x7fffffff) return 0;
    const int64_t ostr = out_stride ? out_stride : n;
    if (ostr < n || (ostr & 3) || ostr * (per - 1) + 4096 > 0x7fffffff) return 0;
    const bool strided = ostr != n || (!inverse && in_stride != n);      // the per-depth form below takes dense signals only
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, inverse, &cw.c)) return 0;
    {
        WxLat one;
        if (!wx_lattice_factor(filt, 1, false, &one)) return 0;
        const long double g = one.g0;                           // product of the cosines of one level
        long double acc = 1;
        for (int l = 0; l <= 12; ++l) { cw.gl[l] = (double)acc; acc *= inverse ? 1 / g : g; }
    }
    WxScratch scr(st);
    // the threshold of denoise() rides on the leaves the inverse takes in; a pyramid's head (idwt: the 64 samples rebuilt by
    // wx_dwttail.hip stand in for positions 0 .. 63 of every signal) is taken by the kernels of wx_lattice_tree_sc.h only
    if (thr && (thr->t || thr->head) && !inverse) return 0;
    WxThreshArg ta{nullptr, 0, 0, 0, 1.0};
    if (thr && thr->t) ta = *thr;
    if (thr) ta.head = thr->head;
    (void)ta;
    // every level under the tree's masks, one permutation through LDS (wx_lattice_tree_sc.h).  The first form of round 3 -- leaves
    // leave right after their level, the kernel returns after the last one -- is kept behind WX_TREE_SC=0 for dense signals: with the
    // tables made in LDS the masked form is at least as fast on every tree measured (depth-4 pyramid, 65536 x 4096 db4: 0.77 / 0.83 ms
    // against 0.85 / 0.82; deep random trees 0.86 / 0.85 against 1.15 / 1.33).
    static const int sc_env = wx_getenv("WX_TREE_SC") ? atoi(wx_getenv("WX_TREE_SC")) : -1;
    if (sc_env != 0 || strided || ta.head || !WX_LAT_TREE_FIRST_FORM) {
        WxLatTreeSc *tsc = (WxLatTreeSc *)scr.alloc(sizeof(WxLatTreeSc));
        if (!tsc) return WX_EHIP;
        if (hipMemsetAsync(tsc->dep, 0, sizeof(tsc->dep), st) != hipSuccess) return wx_set_error(WX_EHIP, "lattice tree tables");
        hipLaunchKernelGGL((k_lat_treesc_prep<SH>), dim3(8), dim3(256), 0, st, dstatus, nstatus, L, tsc);
        hipLaunchKernelGGL(k_lat_treesc_prep2, dim3(1), dim3(64), 0, st, tsc);
        const WxLatTreeSc *ctsc = tsc;
        const unsigned nw = (unsigned)((batch + per - 1) / per);
        const int lsig = (int)(batch - per);
#if WX_LAT_TREE_INV
#define WX_GOS(NSS)                                                                                                  \
    case NSS:                                                                                                        \
        hipLaunchKernelGGL((k_lat_iwpt_treesc_f64<NSS, 2, SH, false>), dim3(nw), dim3(64), 0, st, x, y, L, lsig,      \
                           (unsigned)in_stride, (unsigned)col_stride, (unsigned)ostr, cw, ctsc, ta);                 \
        break;
#define WX_GOST(NSS)                                                                                                 \
    case NSS:                                                                                                        \
        hipLaunchKernelGGL((k_lat_iwpt_treesc_f64<NSS, 2, SH, true>), dim3(nw), dim3(64), 0, st, x, y, L, lsig,       \
                           (unsigned)in_stride, (unsigned)col_stride, (unsigned)ostr, cw, ctsc, ta);                 \
        break;
        if (ta.t) {
            switch (wx_lat_stages(filt.F)) {
                WX_GOST(1) WX_GOST(2) WX_GOST(4)
            default: return 0;
            }
        } else
#undef WX_GOST
#else
#define WX_GOS(NSS)                                                                                                  \
    case NSS:                                                                                                        \
        hipLaunchKernelGGL((k_lat_wpt_treesc_f64<NSS, 2, SH>), dim3(nw), dim3(64), 0, st, x, y, L, lsig,             \
                           (unsigned)in_stride, (unsigned)ostr, cw, ctsc);                                           \
        break;
#endif
        switch (wx_lat_stages(filt.F)) {
            WX_GOS(1) WX_GOS(2) WX_GOS(4) WX_GOS(6) WX_GOS(8) WX_GOS(10)
        default: return 0;
        }
#undef WX_GOS
        const hipError_t es = hipGetLastError();
        if (es != hipSuccess) return wx_set_hip_error(es, "lattice tree launch", __FILE__, __LINE__);
        return 1;
    }
    // The first form is not built any more (round 6: it was reachable through the knob only and was 60 % of these translation units'
    // code: -DWX_LAT_TREE_FIRST_FORM=1 brings it back)
#if !WX_LAT_TREE_FIRST_FORM
    return 0;
#else
    if (ta.head) return 0;
    WxLatTreeTab *tab = (WxLatTreeTab *)scr.alloc(sizeof(WxLatTreeTab));
    if (!tab) return WX_EHIP;
    // (experiment: WX_TREE_DBG_CUT = l leaves the emissions / absorptions deeper than l out -- wrong results, the time of the rest)
    static const int dbg_cut = wx_getenv("WX_TREE_DBG_CUT") ? atoi(wx_getenv("WX_TREE_DBG_CUT")) : 99;
    hipLaunchKernelGGL((k_lat_tree_prep<SH>), dim3(13), dim3(64), 0, st, dstatus, nstatus, L, dbg_cut, tab);
    const int64_t nwave = (batch + per - 1) / per;
    const int last_sig = (int)(batch - per);
    const WxLatTreeTab *ctab = tab;
#if WX_LAT_TREE_INV
#define WX_GOT(NSS)                                                                                                  \
    case NSS:                                                                                                        \
        hipLaunchKernelGGL((k_lat_iwpt_tree_f64<NSS, 2, SH, false>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, \
                           (unsigned)in_stride, (unsigned)col_stride, cw, ctab, ta);                                 \
        break;
    // with the threshold of denoise(): Haar, db2, db4 only (every instantiation of this kernel is half a minute of compile
    // time); other filters keep the fused LDS kernel for that call
#define WX_GOTT(NSS)                                                                                                 \
    case NSS:                                                                                                        \
        hipLaunchKernelGGL((k_lat_iwpt_tree_f64<NSS, 2, SH, true>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, \
                           (unsigned)in_stride, (unsigned)col_stride, cw, ctab, ta);                                 \
        break;
    if (ta.t) {
        switch (wx_lat_stages(filt.F)) {
            WX_GOTT(1) WX_GOTT(2) WX_GOTT(4)
        default: return 0;
        }
        const hipError_t et = hipGetLastError();
        if (et != hipSuccess) return wx_set_hip_error(et, "lattice tree launch", __FILE__, __LINE__);
        return 1;
    }
#undef WX_GOTT
#else
#define WX_GOT(NSS)                                                                                                  \
    case NSS:                                                                                                        \
        hipLaunchKernelGGL((k_lat_wpt_tree_f64<NSS, 2, SH>), dim3((unsigned)nwave), dim3(64), 0, st, x, y, L, last_sig, cw, ctab); \
        break;
#endif
    switch (wx_lat_stages(filt.F)) {
        WX_GOT(1) WX_GOT(2) WX_GOT(4) WX_GOT(6) WX_GOT(8) WX_GOT(10)
    default: return 0;
    }
#undef WX_GOT
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice tree launch", __FILE__, __LINE__);
    return 1;
#endif
}
