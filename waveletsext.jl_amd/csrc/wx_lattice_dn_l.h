// wx_lattice_dn_l.h -- launcher of the one-pass denoise kernels (wx_lattice_dn.h); included by wx_lattice_dn{0 .. 6}.hip with WX_DN_SH = 0 ... 6
// (signals of 4096 >> SH samples) and WX_DN_FN = the launcher's name: one translation unit per length so that the kernels compile in parallel.
// Reference: denoiseall(x, :sig, wt; L, dnt, estnoise = noisest, smooth) Denoising.jl:651-712.
#include "wx_lattice_dn.h"
// waves per SIMD: 3 (0.85 / 0.95 / 0.87 ms per GiB of 1024 / 2048 / 4096-sample signals when measured) against 2 (0.94 / 1.07 / 0.98): the kernel
// waits on its exchanges and scalar loads 40 % of a wavefront's life and a third wavefront fills part of it; the registers the noise estimate
// does not read wait in LDS meanwhile (wx_lattice_dn.h)
#ifndef WX_DN_WPE
#define WX_DN_WPE 3
#endif
#include <cstring>
#include <vector>

// 0 = not applicable (the caller runs the separate kernels), 1 = launched, < 0 = error
// coefs != 0: x holds the coefficients of dwtall(., wt, L) (denoiseall(:dwt)), else the signals (denoiseall(:sig))
int WX_DN_FN(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, int th_kind, double scale, int undersmooth,
             double *sigma, int coefs, hipStream_t st)
{
    constexpr int SH = WX_DN_SH;
    constexpr int64_t per = (int64_t)1 << SH;
    if (n != (4096 >> SH) || L < 1 || L + SH > 12 || filt.F < 2 || batch < per || batch > 0x7fffffff) return 0;
    if ((batch & (per - 1)) && x == y) return 0;             // the tail wavefront re-does signals: out of place only
    // coefficients in: 1.03 ms per GiB at 4096 samples against 0.83 for k_mad_count + the inverse with the threshold on its loads (the estimate in the
    // last layout spans 32 lanes there); from 1024 samples down the one kernel wins (0.98 / 0.96 / 0.67 ms at 1024 / 256 / 64 against 1.13 / 1.46 / 1.39)
    if (coefs && SH < 2) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    // built for 2, 4 and 8 rotation stages: a shorter filter runs on the next of them bit-identically (wx_lattice_factor leaves the missing stages at
    // p = kappa = 0: identities) -- Haar on 2, db3 on 4, db5 ... db7 on 8; filters of up to 16 taps
    const int ns0 = wx_lat_stages(filt.F);
    const int ns = ns0 <= 2 ? 2 : (ns0 <= 4 ? 4 : (ns0 <= 8 ? 8 : 0));
    if (!ns) return 0;
    WxLatW cw2[2];
    memset(cw2, 0, sizeof cw2);
    WxLatW &cwf = cw2[0], &cwi = cw2[1];
    if (!wx_lattice_factor(filt, L, false, &cwf.c) || !wx_lattice_factor(filt, L, true, &cwi.c)) return 0;
    {
        WxLat one;
        if (!wx_lattice_factor(filt, 1, false, &one)) return 0;
        const long double g = one.g0;                           // product of the cosines of one level
        long double af = 1, ai = 1;
        for (int l = 0; l <= 12; ++l) { cwf.gl[l] = (double)af; cwi.gl[l] = (double)ai; af *= g; ai *= 1 / g; }
    }
    cwf.tail_bsig = cwi.tail_bsig = 0;
    WxScratch scr(st);
    // the pyramid of depth L as node flags in heap order: node (d, 0) = index 2^d is split for d < L
    std::vector<uint8_t> tree((size_t)n - 1, 0);
    for (int d = 0; d < L; ++d) tree[((size_t)1 << d) - 1] = 1;
    const uint8_t *dstatus = (const uint8_t *)scr.upload(tree.data(), tree.size());
    WxLatTreeSc *tsc = (WxLatTreeSc *)scr.alloc(sizeof(WxLatTreeSc));
    if (!dstatus || !tsc) return WX_EHIP;
    if (hipMemsetAsync(tsc->dep, 0, sizeof(tsc->dep), st) != hipSuccess) return wx_set_error(WX_EHIP, "lattice tree tables");
    hipLaunchKernelGGL((k_lat_treesc_prep<SH>), dim3(8), dim3(256), 0, st, dstatus, (int64_t)tree.size(), L, tsc);
    hipLaunchKernelGGL(k_lat_treesc_prep2, dim3(1), dim3(64), 0, st, tsc, SH);
    const WxLatTreeSc *ctsc = tsc;
    const WxLatW *cws = (const WxLatW *)wx_const_upload(cw2, sizeof cw2, st, true);
    if (!cws) return WX_EHIP;
    const unsigned nw = (unsigned)((batch + per - 1) / per);
    const int lsig = (int)(batch - per);
    WxDnArg dn;
    dn.kind = th_kind;
    dn.zmask = undersmooth ? (((1u << L) - 1u) << SH) : 0u;
    dn.zval = undersmooth ? 0u : 1u;
    dn.scale = scale;
    dn.sigma = sigma;
    switch (ns) {
#if WX_DN_SH >= 2
#define WX_DN_GO(NSS)                                                                                                          \
    case NSS:                                                                                                                  \
        if (coefs)                                                                                                             \
            hipLaunchKernelGGL((k_lat_denoise_dwt_f64<NSS, 2, SH>), dim3(nw), dim3(64), 0, st, x, y, lsig, (unsigned)n, (unsigned)n, cws, ctsc, dn); \
        else                                                                                                                   \
            hipLaunchKernelGGL((k_lat_denoise_f64<NSS, WX_DN_WPE, SH>), dim3(nw), dim3(64), 0, st, x, y, lsig, (unsigned)n, (unsigned)n, cws, ctsc, dn); \
        break;
#else               // 4096 / 2048 samples: the coefficients-in kernel is not built (declined above)
#define WX_DN_GO(NSS)                                                                                                          \
    case NSS:                                                                                                                  \
        hipLaunchKernelGGL((k_lat_denoise_f64<NSS, WX_DN_WPE, SH>), dim3(nw), dim3(64), 0, st, x, y, lsig, (unsigned)n, (unsigned)n, cws, ctsc, dn); \
        break;
#endif
        WX_DN_GO(2) WX_DN_GO(4) WX_DN_GO(8)
#undef WX_DN_GO
    default: return 0;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "one-pass denoise launch", __FILE__, __LINE__);
    return 1;
}
