// wx_lattice_tree32.h -- launcher of the tree-driven lattice kernels (wx_lattice_tree_sc.h) for Float32 signals of 4096, 2048 and
// 1024 samples: every level under the tree's masks and one permutation through LDS, like Float64.  Round 5: Float32 ARITHMETIC on
// pairs of signals (lat_f2v, FP32A = true): a wavefront takes 2 x 2^SH signals under the one tree of the call (until round 4 the
// rotations ran in Float64 registers: FP64 issue at half the bytes, 0.39-0.54 of the HBM peak).  Reference: Wavelets.jl's wpt / iwpt with a tree::BitVector on an
// AbstractArray{T} (the reference is generic in T: dwt/dwt_one_level.jl:79-83) as called by wptall / iwptall
// (dwt/dwt_all.jl:152-166, 210-225).
#include "wx_lattice_dev.h"
#include "wx_host.h"
#include "wx_lattice_tree_sc.h"

// included by wx_lattice_tree32_{0,1,2}{f,i}.hip with WX_LAT_TREE_SH = 0, 1, 2 (signal length 4096 >> SH), WX_LAT_TREE_INV = 0 / 1
// and WX_LAT_TREE_FN = the launcher's name.  0 = not applicable (the caller takes the fused LDS kernels), 1 = launched, < 0 = error.
int WX_LAT_TREE_FN(bool inverse, const float *x, float *y, int64_t n, int L, int64_t batch, int64_t in_stride, const WxFilt &filt,
                   const uint8_t *dstatus, int64_t nstatus, const WxThreshArg *thr, hipStream_t st, int64_t out_stride)
{
    constexpr int SH = WX_LAT_TREE_SH;
    constexpr int64_t per = (int64_t)1 << SH;
    if (inverse != (WX_LAT_TREE_INV != 0)) return 0;
    if (n != (4096 >> SH) || L < 1 || L + SH > 12 || filt.F < 2 || !dstatus) return 0;
    WxPairPlan pp;
    if (!wx_lat_pair_plan(batch, SH, x == y, &pp)) return 0;   // a remainder below 2^SH signals re-does signals: out of place only
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return 0;
    if (in_stride < n || (in_stride & 3) || in_stride * (2 * per - 1) + 4096 > 0x7fffffff) return 0;
    const int64_t ostr = out_stride ? out_stride : n;
    if (ostr < n || (ostr & 3) || ostr * (2 * per - 1) + 4096 > 0x7fffffff) return 0;
    if (thr && thr->t) return 0;                             // the threshold of denoise() rides on the Float64 kernels only
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
    WxThreshArg ta{nullptr, 0, 0, 0, 1.0};
    if (thr) ta.head = thr->head;
    (void)ta;
    WxLatTreeSc *tsc = (WxLatTreeSc *)scr.alloc(sizeof(WxLatTreeSc));
    if (!tsc) return WX_EHIP;
    if (hipMemsetAsync(tsc->dep, 0, sizeof(tsc->dep), st) != hipSuccess) return wx_set_error(WX_EHIP, "lattice tree tables");
    hipLaunchKernelGGL((k_lat_treesc_prep<SH>), dim3(8), dim3(256), 0, st, dstatus, nstatus, L, tsc);
    hipLaunchKernelGGL(k_lat_treesc_prep2, dim3(1), dim3(64), 0, st, tsc);
    const WxLatTreeSc *ctsc = tsc;
    const unsigned nw = pp.nwave;
    const int lsig = pp.tail_sig;
    cw.tail_bsig = pp.tail_bsig;
#if WX_LAT_TREE_INV
#define WX_GOS(NSS)                                                                                                      \
    case NSS:                                                                                                            \
        hipLaunchKernelGGL((k_lat_iwpt_treesc_f64<NSS, 2, SH, false, float, true>), dim3(nw), dim3(64), 0, st, x, y, L, lsig,   \
                           (unsigned)in_stride, 0u, (unsigned)ostr, cw, ctsc, ta);                                       \
        break;
#else
#define WX_GOS(NSS)                                                                                                      \
    case NSS:                                                                                                            \
        hipLaunchKernelGGL((k_lat_wpt_treesc_f64<NSS, 2, SH, float, true>), dim3(nw), dim3(64), 0, st, x, y, L, lsig,           \
                           (unsigned)in_stride, (unsigned)ostr, cw, ctsc);                                               \
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
