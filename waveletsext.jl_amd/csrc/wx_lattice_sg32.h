// wx_lattice_sg32.h -- launcher of the interleaved lattice kernels (k_lat_wpt_g_f64 / k_lat_iwpt_g_f64, wx_lattice_dev.h) for Float32
// signals of 4096 ... 64 samples, full trees.  Round 5: Float32 ARITHMETIC on pairs of signals (lat_f2v): a wavefront takes 2 x 2^SH
// signals, every rotation is one v_pk_fma_f32 on (signal A, signal B) -- the instruction stream of the Float64 kernel for the same
// number of bytes (until round 4 the rotations ran in Float64 and FP64 issue bound these kernels at half the
// bytes).  The reference rounds to Float32 at every accumulate (dwt/dwt_one_level.jl:97-103): Float32 arithmetic is its semantics.  Reference: Wavelets.jl's wpt / iwpt with a level on an AbstractArray{T} (the reference is generic
// in T: dwt/dwt_one_level.jl:79-83) as called by wptall / iwptall (dwt/dwt_all.jl:152-166, 210-225), and the column pass of the 2-D
// transforms (DWT.jl:500-548).
// Included by wx_lattice_sg32_{1..6}.hip with WX_G32_SH = 1 .. 6 and WX_G32_FN = the launcher's name.
#include "wx_lattice_dev.h"

// 0 = not applicable (the caller goes on to the tree-driven or the fused LDS kernels), 1 = launched, < 0 = error
int WX_G32_FN(bool inverse, const float *x, float *y, int64_t n, int L, int64_t batch, int64_t in_stride, const WxFilt &filt, hipStream_t st)
{
    constexpr int SH = WX_G32_SH;
    // pairs of signals in Float32 arithmetic where the rotations dominate: from 256 samples up and from 7 levels on.  64- and 128-sample
    // signals and shallow trees keep Float64 registers -- few levels are not bound by FP64 issue, and the pair form pays for its two
    // 8-byte accesses per slot pair (measured per GiB: n = 64 0.48 / 0.48 ms against 0.55 / 0.57; n = 1024, L = 4 0.43 / 0.45 against
    // 0.51-0.57; n = 1024, L = 10 0.61 / 0.62 against 0.54 / 0.55: profiles/r04_floor.txt, r05_floor.txt)
    constexpr bool PAIRS_OK = SH <= 4;
    const bool PAIRS = PAIRS_OK && L >= 7;
    const int64_t per = (int64_t)(PAIRS ? 2 : 1) << SH;      // signals per wavefront: two sets of 2^SH
#ifndef WX_G32_NSMAX
#define WX_G32_NSMAX 4
#endif
    if (n != (4096 >> SH) || L < 1 || L + SH < 6 || L + SH > 12 || filt.F < 2 || filt.F > 2 * WX_G32_NSMAX) return 0;
    if (SH == 0 && !PAIRS) return 0;                         // 4096 samples, L = 6: the dedicated kernel of wx_lattice_f32.hip
    WxPairPlan pp;
    if (PAIRS) {
        if (!wx_lat_pair_plan(batch, SH, x == y, &pp)) return 0;   // a remainder below 2^SH signals re-does signals: out of place only
    } else {
        if (batch < per || batch > 0x7fffffff) return 0;
        if ((batch & (per - 1)) && x == y) return 0;         // the tail wavefront re-does signals: out of place only
        pp.nwave = (unsigned)((batch + per - 1) / per);
        pp.tail_sig = (int)(batch - per);
        pp.tail_bsig = 0;
    }
    if (in_stride < n || (in_stride & 3) || in_stride * (per - 1) + 4096 > 0x7fffffff) return 0;
    if (in_stride * (per / 2) > 0x7fffffff) return 0;         // offset of the second signal set (32-bit element offsets)
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, inverse, &cw.c)) return 0;
    for (int l = 0; l <= 12; ++l) cw.gl[l] = 0.0;
    cw.gl[L] = cw.c.g0;
    cw.gl[0] = 1.0;
    const unsigned nwave = pp.nwave;
    const int last_sig = pp.tail_sig;
    cw.tail_bsig = pp.tail_bsig;
// wavefronts per SIMD: three (168 registers: 1-12 spill for up to 8 taps) measured no faster than two (0.57 / 0.58 against 0.55 / 0.53 ms
// per GiB of 4096-sample signals): the kernels are bound by vector issue, which more residency does not add to
#define WX_G32_WPE(NSS) 2
#define WX_GOG(NSS)                                                                                                                  \
    case NSS:                                                                                                                        \
        if (PAIRS && inverse)                                                                                                        \
            hipLaunchKernelGGL((k_lat_iwpt_g_f64<NSS, WX_G32_WPE(NSS), SH, float, PAIRS_OK>), dim3(nwave), dim3(64), 0, st, x, y, L, last_sig, (unsigned)in_stride, cw); \
        else if (PAIRS)                                                                                                              \
            hipLaunchKernelGGL((k_lat_wpt_g_f64<NSS, WX_G32_WPE(NSS), SH, float, PAIRS_OK>), dim3(nwave), dim3(64), 0, st, x, y, L, last_sig, cw); \
        else if constexpr (SH != 0) {                        /* 4096 samples without pairs never get here: not instantiated */      \
            if (inverse)                                                                                                             \
                hipLaunchKernelGGL((k_lat_iwpt_g_f64<NSS, 2, SH, float, false>), dim3(nwave), dim3(64), 0, st, x, y, L, last_sig, (unsigned)in_stride, cw); \
            else                                                                                                                     \
                hipLaunchKernelGGL((k_lat_wpt_g_f64<NSS, 2, SH, float, false>), dim3(nwave), dim3(64), 0, st, x, y, L, last_sig, cw); \
        }                                                                                                                            \
        break;
    switch (wx_lat_stages(filt.F)) {
        WX_GOG(1) WX_GOG(2) WX_GOG(4)
#if WX_G32_NSMAX > 4
        WX_GOG(6) WX_GOG(8)
#endif
#if WX_G32_NSMAX > 8
        WX_GOG(10)
#endif
    default: return 0;
    }
#undef WX_GOG
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpt launch (short Float32 signals)", __FILE__, __LINE__);
    return 1;
}
