// wx_lattice_dn.h -- denoiseall(x, :sig, wt; L, dnt, estnoise = noisest, smooth) in ONE pass over the signals (Float64, 4096 ... 1024
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

struct WxDnArg {
    int kind;                 // Wavelets.Threshold rule, see wx_thresh
    unsigned zmask, zval;     // a coefficient with (index & zmask) == zval is left alone (undersmooth: the coarsest scaling coefficients)
    double scale;             // t = sigma * scale
    double *sigma;            // optional: the noise estimates, one per signal
};

namespace {

// order-preserving image of the doubles (NaN aside): a < b <=> key(a) < key(b), -0 just below +0
__device__ __forceinline__ unsigned long long dn_key(double d)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dn_unkey(unsigned long long k)
{
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}
// the smallest double above a (finite a)
__device__ __forceinline__ double dn_next_up(double a)
{
    if (a == 0.0) return __longlong_as_double(1ll);
    return dn_unkey(dn_key(a) + 1ull);
}
// the value of lane ^ 32 beside the lane's own, as (value of the lower half's lane, value of the upper half's lane): v_permlane32_swap, no LDS
// (with ds_bpermute in its loops the 256-sample kernel spilled 19 ... 168 registers)
__device__ __forceinline__ void dn_x32(unsigned v, unsigned &lower, unsigned &upper)
{
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    lower = r[0];
    upper = r[1];
}
__device__ __forceinline__ void dn_x32(double v, double &lower, double &upper)
{
    unsigned l0, u0, l1, u1;
    dn_x32((unsigned)__double2loint(v), l0, u0);
    dn_x32((unsigned)__double2hiint(v), l1, u1);
    lower = __hiloint2double((int)l1, (int)l0);
    upper = __hiloint2double((int)u1, (int)u0);
}
// a signal's lanes: GW of them, ST apart (ST = 1: consecutive lanes)
constexpr int dn_log2(int v) { return v <= 1 ? 0 : 1 + dn_log2(v >> 1); }
template <int O> __device__ __forceinline__ double dn_xor(double v)
{
    if constexpr (O == 32) { double a, b; dn_x32(v, a, b); return ((unsigned)__lane_id() & 32u) ? a : b; }
    else return __shfl_xor(v, O, 64);
}
template <int GW, int ST = 1> __device__ __forceinline__ double dn_gmin(double v)
{
    lat_for<dn_log2(GW)>([&](auto Ic) { constexpr int o = (ST * GW / 2) >> Ic; const double u = dn_xor<o>(v); v = u < v ? u : v; });
    return v;
}
template <int GW, int ST = 1> __device__ __forceinline__ double dn_gmax(double v)
{
    lat_for<dn_log2(GW)>([&](auto Ic) { constexpr int o = (ST * GW / 2) >> Ic; const double u = dn_xor<o>(v); v = u > v ? u : v; });
    return v;
}
template <int GW, int ST = 1> __device__ __forceinline__ int dn_gsum(int v)
{
    if constexpr (GW == 16 && ST == 1) {
        // every lane of a 16-lane row gets the row's sum: four DPP adds (quad_perm 1 0 3 2, quad_perm 2 3 0 1, row_half_mirror, row_mirror)
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
    } else {
        lat_for<dn_log2(GW)>([&](auto Ic) {
            constexpr int o = (ST * GW / 2) >> Ic;
            if constexpr (o == 32) { unsigned a, b; dn_x32((unsigned)v, a, b); v = (int)(a + b); }
            else v += __shfl_xor(v, o, 64);
        });
    }
    return v;
}
// c += #{lanes of the group with pred}: the whole wavefront -> ballot + population count (scalar), else a per-lane counter (dn_gsum later)
template <int GW> __device__ __forceinline__ void dn_acc(int &c, bool pred)
{
    if constexpr (GW == 64) c += (int)__popcll(__builtin_amdgcn_ballot_w64(pred));
    else c += pred ? 1 : 0;
}
template <int GW, int ST = 1> __device__ __forceinline__ int dn_fin(int c)
{
    if constexpr (GW == 64) return c;
    else return dn_gsum<GW, ST>(c);
}
__device__ __forceinline__ bool dn_any(bool v) { return __builtin_amdgcn_ballot_w64(v) != 0; }

// the detail registers of the layout the root level ran in: index bit CB of the register set, class (= signal bits held by the register
// index) q = r mod 2^CB
template <int CB, typename F> __device__ __forceinline__ void dn_each(F &&f)
{
    lat_for<64>([&](auto Rc) {
        constexpr int r = Rc;
        if constexpr (CB < 0) f(Rc, std::integral_constant<int, 0>{});                    // every register, one class (dn_noisest in the last layout)
        else if constexpr ((r >> CB) & 1) f(Rc, std::integral_constant<int, (r & ((1 << CB) - 1))>{});
    });
}
constexpr int dn_nc(int cb) { return cb < 0 ? 1 : 1 << cb; }

// median (Statistics.median!: a/2 + b/2 of the order statistics k and k + 1, cnt even) of v = e (DEV = false) or |e - ctr| (DEV = true) per class
// and lane group; [blo, bhi): #{v < blo} = 0, #{v < bhi} = cnt
template <int CB, int GW, int ST, bool DEV>
__device__ __forceinline__ void dn_median(const double (&e)[64], const double (&ctr)[dn_nc(CB)], const double (&blo)[dn_nc(CB)],
                                          const double (&bhi)[dn_nc(CB)], int cnt, bool act, double (&med)[dn_nc(CB)])
{
    constexpr int NC = dn_nc(CB);
    const int k = cnt / 2 - 1;
    // the deviations are formed again in every pass: hoisted out of the loops (they do not change) they are 32 / 64 more live doubles -- the
    // centre goes through an opaque copy per pass
    double cc[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) cc[q] = ctr[q];
    auto opaque = [&]() {
        if constexpr (DEV) {
#pragma unroll
            for (int q = 0; q < NC; ++q) asm volatile("" : "+v"(cc[q]));
        }
    };
    auto val = [&](auto Rc, auto Qc) -> double {
        constexpr int r = Rc, q = Qc;
        if constexpr (DEV) return fabs(e[r] - cc[q]);
        else return e[r];
    };
    // smallest v >= lo of every group
    auto min_ge = [&](const double (&lo)[NC], double (&a)[NC]) {
#pragma unroll
        for (int q = 0; q < NC; ++q) a[q] = __builtin_inf();
        opaque();
        dn_each<CB>([&](auto Rc, auto Qc) {
            constexpr int q = Qc;
            const double v = val(Rc, Qc);
            const double w = v >= lo[q] ? v : __builtin_inf();
            a[q] = w < a[q] ? w : a[q];
        });
#pragma unroll
        for (int q = 0; q < NC; ++q) a[q] = dn_gmin<GW, ST>(a[q]);
    };
    double lo[NC], hi[NC];
    int clo[NC], chi[NC], stall[NC];
    bool done[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) { lo[q] = blo[q]; hi[q] = bhi[q]; clo[q] = 0; chi[q] = cnt; stall[q] = 0; done[q] = !act; }
    for (int it = 0; it < 192; ++it) {
        bool forced[NC], anyst = false;
#pragma unroll
        for (int q = 0; q < NC; ++q) { forced[q] = !done[q] && stall[q] >= 2; anyst = anyst || forced[q]; }
        if (dn_any(anyst)) {
            double a[NC];
            min_ge(lo, a);
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if (forced[q]) { lo[q] = a[q]; stall[q] = 0; }
        }
        double p[NC];
        bool live = false, odd = false, vsp[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            p[q] = hi[q];
            vsp[q] = false;
            if (!done[q]) {
                if (chi[q] - clo[q] <= 1) done[q] = true;
                else {
                    p[q] = forced[q] ? dn_next_up(lo[q]) : lo[q] * 0.5 + hi[q] * 0.5;
                    vsp[q] = !(p[q] > lo[q] && p[q] < hi[q]) || (it >= 48 && !forced[q]);      // the value's middle does not separate (any more)
                    odd = odd || vsp[q];
                }
            }
        }
        if (dn_any(odd)) {
            // rare: the middle of the order-preserving integer images; no double between lo and hi -> order statistic k is lo itself
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if (vsp[q]) {
                    const unsigned long long kl = dn_key(lo[q]), kh = dn_key(hi[q]);
                    p[q] = dn_unkey(kl + ((kh - kl) >> 1));
                    if (!(p[q] > lo[q] && p[q] < hi[q])) { done[q] = true; p[q] = hi[q]; }
                }
        }
#pragma unroll
        for (int q = 0; q < NC; ++q) live = live || !done[q];
        if (!dn_any(live)) break;
        int c[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) c[q] = 0;
        opaque();
        dn_each<CB>([&](auto Rc, auto Qc) {
            constexpr int q = Qc;
            dn_acc<GW>(c[q], val(Rc, Qc) < p[q]);
        });
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int cq = dn_fin<GW, ST>(c[q]);
            if (!done[q]) {
                stall[q] = (cq == clo[q] || cq == chi[q]) ? stall[q] + 1 : 0;
                if (cq <= k) { lo[q] = p[q]; clo[q] = cq; }
                else { hi[q] = p[q]; chi[q] = cq; }
            }
        }
    }
    // order statistic k = the smallest v >= lo; k + 1 = the same value when at least k + 2 elements are <= it, else the next larger one
    double a[NC], nx[NC];
    int le[NC];
    min_ge(lo, a);
#pragma unroll
    for (int q = 0; q < NC; ++q) { nx[q] = __builtin_inf(); le[q] = 0; }
    opaque();
    dn_each<CB>([&](auto Rc, auto Qc) {
        constexpr int q = Qc;
        const double v = val(Rc, Qc);
        dn_acc<GW>(le[q], v <= a[q]);
        const double w = v > a[q] ? v : __builtin_inf();
        nx[q] = w < nx[q] ? w : nx[q];
    });
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const double b = dn_fin<GW, ST>(le[q]) >= k + 2 ? a[q] : dn_gmin<GW, ST>(nx[q]);
        med[q] = a[q] / 2 + b / 2;
    }
}

// noise estimates of the signals whose finest details sit in the registers with bit CB set (CB < 0: in every register of the lanes with `act`):
// sig[q] for class q of this lane's group (GW lanes, ST apart)
template <int CB, int GW, int ST = 1>
__device__ __forceinline__ void dn_noisest(const double (&e)[64], double (&sig)[dn_nc(CB)], bool act = true)
{
    constexpr int NC = dn_nc(CB);
    constexpr int cnt = (CB < 0 ? 64 : (32 >> CB)) * GW;
    double vmin[NC], vmax[NC], zero[NC], med[NC], dhi[NC], mad[NC];
    int bad[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) { vmin[q] = __builtin_inf(); vmax[q] = -__builtin_inf(); bad[q] = 0; zero[q] = 0.0; }
    dn_each<CB>([&](auto Rc, auto Qc) {
        constexpr int r = Rc, q = Qc;
        const double v = e[r];
        vmin[q] = v < vmin[q] ? v : vmin[q];
        vmax[q] = v > vmax[q] ? v : vmax[q];
        dn_acc<GW>(bad[q], v != v);
    });
    double hi0[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        vmin[q] = dn_gmin<GW, ST>(vmin[q]); vmax[q] = dn_gmax<GW, ST>(vmax[q]); bad[q] = dn_fin<GW, ST>(bad[q]);
        hi0[q] = dn_next_up(vmax[q]);
    }
    dn_median<CB, GW, ST, false>(e, zero, vmin, hi0, cnt, act, med);
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const double d0 = fabs(vmin[q] - med[q]), d1 = fabs(vmax[q] - med[q]);
        dhi[q] = dn_next_up(d0 > d1 ? d0 : d1);
    }
    dn_median<CB, GW, ST, true>(e, med, zero, dhi, cnt, act, mad);
#pragma unroll
    for (int q = 0; q < NC; ++q) sig[q] = bad[q] ? __builtin_nan("") : mad[q] / 0.6745;
}

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
#ifdef WX_DN_NOSEL
        for (int q = 0; q < dn_nc(CB); ++q) sg[q] = regs[2 + q];
#else
        dn_noisest<CB, GW, ST>(regs, sg, act);
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
#ifndef WX_DN_NOTHR
    {
        const double t = tsm[lane & ((1 << SH) - 1)];
        dn_threshold(c, t, dn.kind, lane, dn.zmask, dn.zval);
    }
#endif
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

}  // namespace
