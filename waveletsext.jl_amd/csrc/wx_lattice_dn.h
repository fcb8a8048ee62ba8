// wx_lattice_dn.h -- denoiseall(x, :sig, wt; L, dnt, estnoise = noisest, smooth) in ONE pass over the signals (Float64, 4096 ... 64
// samples): pyramid analysis -> per-signal noise estimate -> threshold -> pyramid synthesis without the coefficients leaving the registers.
//
// Reference: Denoising.jl:651-712 (denoiseall: dwtall -> noisest per signal -> threshold! -> idwtall), Denoising.jl:483-599 (denoise),
// noisest Denoising.jl:214-232 (Wavelets.Threshold.noisest: mad!(finest detail coefficients) / 0.6745, Statistics.median! = middle of the two
// central order statistics), Wavelets.Threshold.threshold! (HardTH / SoftTH / SemiSoftTH / SteinTH), VisuShrink t = sqrt(2 log n).
//
// The separate kernels of this pipeline (k_lat_wpt_treesc_f64 -> k_mad* -> k_lat_iwpt_treesc_f64 with the threshold on its loads) move
// 4.5 x the signals' bytes and take 1.8 ms per GiB of signals, 0.7 ... 0.85 ms of it the order statistics (profiles/r06_denoise_onepass.md).
// Here a wavefront keeps its 4096 coefficients (2^SH signals of 4096 >> SH samples) in the in-place lattice layouts of wx_lattice_tree_sc.h:
//   * the ROOT level leaves the finest details -- final values, every level normalises its own gains -- in the registers whose index has
//     the root's bit set; the two medians of the noise estimate are EXACT ORDER STATISTICS found by counting: a pivot p per signal,
//     c = #{v < p} summed over the signal's lanes (a ballot and a population count per register when the signal is the whole
//     wavefront, per-lane counters and a row reduction when it is a row of 16 lanes), the bracket [lo, hi) with #{v < lo} <= k < #{v < hi}
//     halves in value until one element is left in it (middle of lo and hi; the order-preserving integer image of the doubles takes over
//     when values stop separating: at most 64 more steps whatever the data); a pivot that makes no progress twice -- many equal values
//     at the median: sparse signals -- is replaced by the successor of the smallest candidate, which removes all its copies at once.
//     The deviations |v - median| are formed on the fly for the second median.  Same order statistics, the same a/2 + b/2 and the same
//     rounded subtraction as the selection of k_mad (wx_denoise.hip) and the oracle's sort; NaN anywhere in the details gives NaN.
//   * the remaining levels of the pyramid run under the tree's masks exactly as in k_lat_wpt_treesc_f64, down to single coefficients (the
//     separate pipeline hands levels 7 ... 12 to the lane-local tail kernels of wx_dwttail.hip; here they are masked levels whose idle
//     register classes are skipped);
//   * in the last layout (lane = index bits 5 .. 0, so one signal per lane) every coefficient is thresholded with its signal's t = sigma
//     x dnt.t -- but the coarsest scaling coefficients when smooth = :undersmooth (index bits SH .. SH + L - 1 all zero);
//   * the synthesis levels run back through the same layouts and the signals leave through lat_emit: the packet-order permutation of the
//     two separate transforms never happens.
// Bytes: x in, the denoised x out.
#pragma once
#include "wx_lattice_dev.h"
#include "wx_host.h"
#include "wx_lattice_tree_sc.h"
#include "wx_select_count.h"

#ifndef WX_DN_PARK2
#define WX_DN_PARK2 5
#endif

struct WxDnArg {
    int kind;                 // Wavelets.Threshold rule, see wx_thresh
    unsigned zmask, zval;     // a coefficient with (index & zmask) == zval is left alone (undersmooth: the coarsest scaling coefficients)
    double scale;             // t = sigma * scale
    double *sigma;            // optional: the noise estimates, one per signal
};

namespace {

// HardTH, SoftTH and SemiSoftTH (wx_thresh, wx_common.h) as ONE branch-free rule with wave-uniform parameters: out = v where the rule keeps the
// coefficient (Hard: not |v| <= t; SemiSoft: |v| > 2 t; Soft: never), else sign(v) max(A |v| - B, 0) with (A, B) = (0, 1) Hard, (1, t) Soft,
// (2, 2 t) SemiSoft -- A |v| is exact, so the fused multiply-add rounds once like the rule's own subtraction.  (A switch around four copies of the
// 64 updates spilled 198 registers, a branch per register 22, Hard against the rest 40.  SteinTH keeps the separate kernels.)
__device__ __forceinline__ void dn_threshold(double (&c)[64], double t, int kind, int lane, unsigned zmask, unsigned zval)
{
    const unsigned lm = (unsigned)lane & zmask;
    const bool hard = kind == 0;
    const double A = kind == 0 ? 0.0 : (kind == 1 ? 1.0 : 2.0);
    const double B = kind == 0 ? 1.0 : (kind == 1 ? t : 2.0 * t);
    const double K = kind == 0 ? t : (kind == 1 ? __builtin_inf() : 2.0 * t);
    lat_for<64>([&](auto Rc) {
        constexpr int r = Rc;
        const unsigned im = lm | ((unsigned)(r << 6) & zmask);
        const double v = c[r], av = fabs(v);
        const double lin = __builtin_fma(A, av, -B);
        const double sg = v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : v);
        const double sh = lin < 0.0 ? 0.0 : sg * lin;
        const bool keep = hard ? !(av <= K) : (av > K);
        const double o = keep ? v : sh;
        c[r] = im == zval ? v : o;
    });
}

#define WX_DN_FWD(KK, HH, REG, BIT, MK, ANY)                                                   \
    if constexpr (BIT == SH) {                                                                  \
        lat_level<KK, HH, NS, false>(REG, *cfp);                                                \
        _Pragma("unroll") for (int r = 0; r < 64; ++r) REG[r] = lat_mul(REG[r], ((r >> KK) & 1) ? ginv_ : g);    \
        if constexpr (!SELC) {                                                                  \
            noise(REG);                                                                         \
            fresh();                                                                            \
        }                                                                                       \
    } else if constexpr (BIT > SH) {                                                            \
        if (ANY) lat_level_hm<KK, HH, NS, false>(REG, *cfp, MK, g, ginv_);                      \
    }
#define WX_DN_INV(KK, HH, REG, BIT, MK, ANY)                                                   \
    if constexpr (BIT == SH) {                                                                  \
        _Pragma("unroll") for (int r = 0; r < 64; ++r) REG[r] = lat_mul(REG[r], ((r >> KK) & 1) ? gd : ga);     \
        lat_level<KK, HH, NS, true>(REG, ci);                                                   \
    } else if constexpr (BIT > SH) {                                                            \
        if (ANY) lat_level_hm<KK, HH, NS, true>(REG, ci, MK, ga, gd);                           \
    }

// SH = 0 ... 6: 2^SH signals of 4096 >> SH samples per wavefront; `tab` = the masks of the pyramid of depth L (k_lat_treesc_prep<SH>)
template <int NS, int WPE, int SH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_denoise_f64(
    const double *__restrict__ x, double *__restrict__ y, int last_sig, unsigned in_stride, unsigned out_stride, const WxLatW *__restrict__ cws,
    const WxLatTreeSc *__restrict__ tab, WxDnArg dn)
{
    static_assert(SH >= 0 && SH <= 6, "4096 .. 64 samples");
    __shared__ __attribute__((aligned(16))) double lds[WX_LAT_LDS];       // the window of the exchanges (absorb / emit / layout changes)
    __shared__ double tsm[1 << SH];
    __shared__ double lds2[64 * (WX_DN_PARK2 > 0 ? WX_DN_PARK2 : 1)];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    const int sig0 = min((int)blockIdx.x << SH, last_sig);
    const double *xs = x + (int64_t)sig0 * in_stride;
    double *ys = y + (int64_t)sig0 * out_stride;
    // the two coefficient sets (cws[0] analysis, cws[1] synthesis) stay in constant memory and are read where a phase needs them: as two
    // by-value arguments they cost 74 ... 100 spilled scalar registers
    typedef const WxLatW __attribute__((address_space(4))) *dn_cst;
    dn_cst cwp = (dn_cst)(uintptr_t)cws;
    asm volatile("" : "+s"(cwp));
    const WxLatW &cwf = *(const WxLatW *)cwp;
    const WxLat *cfp = &cwf.c;
    // analysis: gl[1] = g, g2 = g^-2 (the a-slot of a split node leaves as a / g, the d-slot as d g); synthesis: gl[1] = 1 / g, g2 = g^2
    double g = cwf.gl[1], ginv_ = cwf.c.g2 * cwf.gl[1];
    // behind the noise estimate the constants are read again: kept in scalar registers across its loops they were 86 spilled registers and
    // 2900 lane reads in the levels that follow
    auto fresh = [&]() {
        asm volatile("" : "+s"(cwp));
        const WxLatW &w = *(const WxLatW *)cwp;
        cfp = &w.c;
        g = w.gl[1];
        ginv_ = w.c.g2 * w.gl[1];
    };
    // where the noise estimate runs: in the layout of the root level (A for 4096 / 2048 samples, B for 1024 / 512, C for 64; CB = signal bits held by
    // the register index = the classes of dn_noisest, GW = lanes per signal), or -- 256 / 128 samples, where layout B would have 4 / 8 classes
    // with a pivot each (1.23 / 1.64 ms per GiB) -- in the LAST layout just before the threshold: there the root's bit is lane bit SH, the finest
    // details of a signal are every register of the 2 / 1 lanes with that bit set and the signal's low lane bits (they are final since the root level)
    constexpr bool SELC = SH == 4 || SH == 5;
    constexpr int CB = SELC ? -1 : (SH == 1 ? 1 : (SH == 3 ? 1 : 0)), GW = SELC ? (32 >> SH) : (SH <= 1 ? 64 : (SH <= 5 ? 16 : 1)), ST = SELC ? (2 << SH) : 1;
    auto noise = [&](double (&regs)[64]) {
        double sg[dn_nc(CB)];
        const bool act = SELC ? ((lane >> SH) & 1) != 0 : true;
#ifdef WX_DN_NOSEL             // timing experiment of profiles/r06_denoise_onepass.md section 3: the estimate replaced by a register read (wrong results)
        for (int q = 0; q < dn_nc(CB); ++q) sg[q] = regs[2 + q];
#else
        // 4096 ... 512 samples at three wavefronts per SIMD: 17 of the 32 registers the estimate does not read (root bit clear) wait in the exchange
        // window, which is idle here -- left to the register allocator they went to scratch and came back (21 doubles per lane: the kernel moved
        // 1.33 x its algorithmic bytes through HBM, profiles/r06_denoise_onepass.md)
        constexpr int NST = (!SELC && SH <= 3) ? 17 + WX_DN_PARK2 : 0;             // 17 fill the window, WX_DN_PARK2 more an array of their own
        const unsigned park = lds0 + 8u * (unsigned)lane;
        const unsigned park2 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds2 + 8u * (unsigned)lane;
        lat_for<NST>([&](auto Jc) {
            constexpr int j = Jc, r = ((j >> CB) << (CB + 1)) | (j & ((1 << CB) - 1));
            if constexpr (j < 17) lds_wr<8 * 64 * j>(park, regs[r]);
            else lds_wr<8 * 64 * (j - 17)>(park2, regs[r]);
        });
#pragma unroll
        for (int j = 0; j < NST; ++j) asm volatile("" : "=v"(regs[((j >> CB) << (CB + 1)) | (j & ((1 << CB) - 1))]));      // dead until read back
        dn_noisest<CB, GW, ST>(regs, sg, act);
        lat_for<NST>([&](auto Jc) {
            constexpr int j = Jc, r = ((j >> CB) << (CB + 1)) | (j & ((1 << CB) - 1));
            if constexpr (j < 17) regs[r] = lds_rd<8 * 64 * j, double>(park);
            else regs[r] = lds_rd<8 * 64 * (j - 17), double>(park2);
        });
        if constexpr (NST > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < NST; ++j) asm volatile("" : "+v"(regs[((j >> CB) << (CB + 1)) | (j & ((1 << CB) - 1))]));  // no use above the wait
        }
#endif
#pragma unroll
        for (int q = 0; q < dn_nc(CB); ++q) {
            // signal of the wavefront: index bits SH - 1 .. 0; one lane of its group writes
            const int s = SELC ? (lane & ((1 << SH) - 1)) : (SH <= 1 ? q : (SH <= 5 ? ((q << 2) | (lane >> 4)) : lane));
            const bool first = SELC ? (lane >> SH) == 1 : (lane & (GW - 1)) == 0;
            if (first) {
                tsm[s] = sg[q] * dn.scale;
                if (dn.sigma) dn.sigma[sig0 + s] = sg[q];
            }
        }
    };
    double c[64];
    if constexpr (SH >= 6)
        lat_absorb<6, 16 * SH>(c, lds0, xs, lane, cwf, in_stride, 0, 0, 0, 0xffffffffu);
    else {
        double bb[64];
        if constexpr (SH < 2) {
            double a[64];
            lat_absorb<0, 16 * SH>(a, lds0, xs, lane, cwf, in_stride, 0, 0, 0, 0xffffffffu);
            WX_DN_FWD(0, 6, a, 0, tab->mA, true)
            WX_DN_FWD(1, 6, a, 1, tab->mA, tab->anyA)
            lat_t2(a, bb, lds0, lane);
        } else
            lat_absorb<2, 16 * SH>(bb, lds0, xs, lane, cwf, in_stride, 0, 0, 0, 0xffffffffu);
        WX_DN_FWD(0, 4, bb, 2, tab->mB + 0, tab->anyB[0])
        WX_DN_FWD(1, 4, bb, 3, tab->mB + 32, tab->anyB[1])
        WX_DN_FWD(2, 4, bb, 4, tab->mB + 64, tab->anyB[2])
        WX_DN_FWD(3, 4, bb, 5, tab->mB + 96, tab->anyB[3])
        if (tab->deepB) {
            WX_DN_FWD(4, 4, bb, 6, tab->mB + 128, tab->anyB[4])
            WX_DN_FWD(5, 4, bb, 7, tab->mB + 160, tab->anyB[5])
        }
        lat_t3(bb, c, lds0, lane);
    }
    const unsigned long long *mk = tab->mC;
    if (!tab->deepB) {
        if (tab->anyC[0]) lat_level_cm<0, NS, false>(c, *cfp, mk + 0, g, ginv_);
        if constexpr (SH >= 6) {                                 // 64 samples: this was the root level (its mask is every lane)
            noise(c);
            fresh();
        }
        if (tab->anyC[1]) lat_level_cm<1, NS, false>(c, *cfp, mk + 32, g, ginv_);
    }
    if (tab->anyC[2]) lat_level_cm<2, NS, false>(c, *cfp, mk + 64, g, ginv_);
    if (tab->anyC[3]) lat_level_cm<3, NS, false>(c, *cfp, mk + 96, g, ginv_);
    if (tab->anyC[4]) lat_level_cm<4, NS, false>(c, *cfp, mk + 128, g, ginv_);
    if (tab->anyC[5]) lat_level_cm<5, NS, false>(c, *cfp, mk + 160, g, ginv_);
    if constexpr (SELC) noise(c);
    // threshold: layout C, lane = index bits 5 .. 0 -> the lane's signal is its low SH bits
    lat_sync();
    {
        const double t = tsm[lane & ((1 << SH) - 1)];
        dn_threshold(c, t, dn.kind, lane, dn.zmask, dn.zval);
    }
    dn_cst cip = (dn_cst)(uintptr_t)(cws + 1);
    asm volatile("" : "+s"(cip));
    const WxLatW &cwi = *(const WxLatW *)cip;
    const WxLat &ci = cwi.c;
    const double ga = cwi.gl[1], gd = cwi.c.g2 * cwi.gl[1];
    if (tab->anyC[5]) lat_level_cm<5, NS, true>(c, ci, mk + 160, ga, gd);
    if (tab->anyC[4]) lat_level_cm<4, NS, true>(c, ci, mk + 128, ga, gd);
    if (tab->anyC[3]) lat_level_cm<3, NS, true>(c, ci, mk + 96, ga, gd);
    if (tab->anyC[2]) lat_level_cm<2, NS, true>(c, ci, mk + 64, ga, gd);
    if (!tab->deepB) {
        if (tab->anyC[1]) lat_level_cm<1, NS, true>(c, ci, mk + 32, ga, gd);
        if (tab->anyC[0]) lat_level_cm<0, NS, true>(c, ci, mk + 0, ga, gd);
    }
    if constexpr (SH >= 6)
        lat_emit<6, 16 * SH>(c, lds0, ys, lane, cwi, out_stride, 0, 0, 0xffffffffu);
    else {
        double bb[64];
        lat_t3i(c, bb, lds0, lane);
        if (tab->deepB) {
            WX_DN_INV(5, 4, bb, 7, tab->mB + 160, tab->anyB[5])
            WX_DN_INV(4, 4, bb, 6, tab->mB + 128, tab->anyB[4])
        }
        WX_DN_INV(3, 4, bb, 5, tab->mB + 96, tab->anyB[3])
        WX_DN_INV(2, 4, bb, 4, tab->mB + 64, tab->anyB[2])
        WX_DN_INV(1, 4, bb, 3, tab->mB + 32, tab->anyB[1])
        WX_DN_INV(0, 4, bb, 2, tab->mB + 0, tab->anyB[0])
        if constexpr (SH >= 2) {
            lat_emit<2, 16 * SH>(bb, lds0, ys, lane, cwi, out_stride, 0, 0, 0xffffffffu);
        } else {
            double a[64];
            lat_t2i(bb, a, lds0, lane);
            WX_DN_INV(1, 6, a, 1, tab->mA, tab->anyA)
            WX_DN_INV(0, 6, a, 0, tab->mA, true)
            lat_emit<0, 16 * SH>(a, lds0, ys, lane, cwi, out_stride, 0, 0, 0xffffffffu);
        }
    }
}
#undef WX_DN_FWD
#undef WX_DN_INV

// denoiseall(xw, :dwt, wt; L, dnt, smooth): the coefficients of the pyramid come in (packet order), the denoised signals go out -- the front end
// of k_lat_iwpt_treesc_f64 (KiB loads -> packet-order image in LDS -> table-addressed reads into the last layout), then the noise estimate in that
// layout (the finest details of a signal are every register of the 32 >> SH lanes with lane bit SH set and the signal's low lane bits; 64 samples: the
// odd registers of the signal's lane), the threshold, and the synthesis of k_lat_denoise_f64.  Replaces k_mad* + k_lat_iwpt_treesc_f64 with the
// threshold on its loads + the tail kernel of the deep levels: 2 x the signals' bytes instead of 2.5 x and three launches.
template <int NS, int WPE, int SH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_denoise_dwt_f64(
    const double *__restrict__ xw, double *__restrict__ y, int last_sig, unsigned in_stride, unsigned out_stride, const WxLatW *__restrict__ cws,
    const WxLatTreeSc *__restrict__ tab, WxDnArg dn)
{
    static_assert(SH >= 0 && SH <= 6, "4096 .. 64 samples");
    __shared__ __attribute__((aligned(16))) double lds[2048];             // the packet-order image (two halves of 16 KiB in turn)
    __shared__ double tsm[1 << SH];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    const int sig0 = min((int)blockIdx.x << SH, last_sig);
    double *ys = y + (int64_t)sig0 * out_stride;
    typedef const WxLatW __attribute__((address_space(4))) *dn_cst;
    dn_cst cip = (dn_cst)(uintptr_t)(cws + 1);
    asm volatile("" : "+s"(cip));
    const WxLatW &cwi = *(const WxLatW *)cip;
    const WxThreshArg none{nullptr, 0, 0, 0, 1.0};
    constexpr int CB = SH == 6 ? 0 : -1, GW = SH == 6 ? 1 : (32 >> SH), ST = SH == 6 ? 1 : (2 << SH);
    lat_treesc_inv_h<NS, SH, false, double, false>(
        xw + (int64_t)sig0 * in_stride, sig0, lds0, lane, in_stride, 0u, (unsigned)(in_stride << SH), 1u << SH, cwi, tab, none,
        [&](double (&regs)[64]) { lat_emit<(SH < 2 ? 0 : (SH < 6 ? 2 : 6)), 16 * SH>(regs, lds0, ys, lane, cwi, out_stride, 0, 0, 0xffffffffu); },
        [&](double (&c)[64]) {
            double sg[1];
            const bool act = SH == 6 ? true : ((lane >> SH) & 1) != 0;
            dn_noisest<CB, GW, ST>(c, sg, act);
            const int s = lane & ((1 << SH) - 1);
            if (SH == 6 || (lane >> SH) == 1) {
                tsm[s] = sg[0] * dn.scale;
                if (dn.sigma) dn.sigma[sig0 + s] = sg[0];
            }
            lat_sync();
            dn_threshold(c, tsm[s], dn.kind, lane, dn.zmask, dn.zval);
        });
}

}  // namespace
