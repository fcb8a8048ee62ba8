// wx_lattice_lo.hip -- full packet trees of depth 1 .. 5 of 4096-sample Float64 signals on the lattice kernels.
//
// k_lat_wpt_f64 / k_lat_iwpt_f64 (wx_lattice_dev.h) always pass through the three register layouts, so they start at depth 6.
// A shallower tree ends in layout A (levels 1, 2) or B (levels 3 .. 5): the signal enters layout A as in k_lat_wpd_f64,
// runs its L levels, and leaves through the static bit routing of
// lat_emit<LAY, L> -- the same routine that writes row L of a packet table in k_lat_wpd_f64 (complete 128-byte lines).  The
// inverse is the mirror: lat_absorb<LAY, L>, the inverse levels, lat_emit<0, 0>.
// Reference: wpt / iwpt of Wavelets.jl as called by wptall / iwptall (dwt/dwt_all.jl:152-225) with an integer depth L.
#include "wx_lattice_dev.h"

namespace {

template <int NS, int WPE, int L>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_wpt_lo_f64(
    const double *__restrict__ x, double *__restrict__ y, WxLatW cw)
{
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    const int64_t off = (int64_t)blockIdx.x * 4096;
    const double *xs = x + off;
    double *ys = y + off;
    const WxLat &cf = cw.c;
    double a[64];
    {   // natural order -> layout A (the load of k_lat_wpd_f64)
        lat_d2 r[32];
        const unsigned xo = 64u * (lane >> 3) + 2u * (lane & 7);
        lat_for<32>([&](auto Q) {
            constexpr int hi3 = Q / 4, f = Q % 4;
            r[Q] = lat_ld2(lat_sbase(xs + 512 * hi3 + 16 * f) + xo);
        });
        const unsigned wa = lds0 + 8u * (17u * (lane >> 3) + 2u * (lane & 7)), ra = lds0 + 8u * 17u * lane;
        lat_for<4>([&](auto Fq) {
            constexpr int f = Fq;
            lat_for<8>([&](auto Hq) {
                constexpr int hi3 = Hq;
                lds_wr<8 * (136 * hi3)>(wa, r[4 * hi3 + f].x);
                lds_wr<8 * (136 * hi3 + 1)>(wa, r[4 * hi3 + f].y);
            });
            double t[16];
            lat_for<16>([&](auto M) {
                constexpr int m = M;
                t[m] = lds_rd<8 * m>(ra);
            });
            lat_wait16<0>(t);
            lat_for<16>([&](auto M) {
                constexpr int m = M;
                a[16 * f + m] = t[m];
            });
        });
    }
    // the depth is a template parameter: one straight-line kernel per depth (with the depth at run time -- early returns
    // after conditional emits, or a switch -- the register allocator keeps both layouts alive and spills 270 .. 420 registers)
    lat_level<0, 6, NS, false>(a, cf);
    if constexpr (L == 1) {
        lat_emit<0, 1>(a, lds0, ys, lane, cw);
    } else {
        lat_level<1, 6, NS, false>(a, cf);
        if constexpr (L == 2) {
            lat_emit<0, 2>(a, lds0, ys, lane, cw);
        } else {
            double bb[64];
            lat_t2(a, bb, lds0, lane);
            lat_level<0, 4, NS, false>(bb, cf);
            if constexpr (L >= 4) lat_level<1, 4, NS, false>(bb, cf);
            if constexpr (L >= 5) lat_level<2, 4, NS, false>(bb, cf);
            lat_emit<2, L>(bb, lds0, ys, lane, cw);
        }
    }
}

// leaves of signal s at xw + s in_stride (a dense array or row L of packet tables)
template <int NS, int WPE, int L>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_iwpt_lo_f64(
    const double *__restrict__ xw, double *__restrict__ y, int64_t in_stride, WxLatW cw)
{
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    const double *xs = xw + (int64_t)blockIdx.x * in_stride;
    double *ys = y + (int64_t)blockIdx.x * 4096;
    const WxLat &cf = cw.c;
    double a[64];
    if constexpr (L >= 3) {
        double bb[64];
        lat_absorb<2, L>(bb, lds0, xs, lane, cw);
        if constexpr (L > 4) lat_level<2, 4, NS, true>(bb, cf);
        if constexpr (L > 3) lat_level<1, 4, NS, true>(bb, cf);
        lat_level<0, 4, NS, true>(bb, cf);
        lat_t2i(bb, a, lds0, lane);
        lat_level<1, 6, NS, true>(a, cf);
    } else if constexpr (L == 2) {
        lat_absorb<0, 2>(a, lds0, xs, lane, cw);
        lat_level<1, 6, NS, true>(a, cf);
    } else {
        lat_absorb<0, 1>(a, lds0, xs, lane, cw);
    }
    lat_level<0, 6, NS, true>(a, cf);
    lat_emit<0, 0>(a, lds0, ys, lane, cw);
}

}  // namespace

// 0 = not applicable, 1 = launched, < 0 = error
int wx_lattice_launch_lo(bool inverse, const double *x, double *y, int L, int64_t batch, int64_t in_stride, const WxFilt &filt,
                         hipStream_t st)
{
    if (L < 1 || L > 5 || filt.F < 2 || batch <= 0 || batch > 0x7fffffff) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    if (inverse && (in_stride < 4096 || (in_stride & 3))) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, inverse, &cw.c)) return 0;
    for (int l = 0; l <= 12; ++l) cw.gl[l] = 0.0;
    cw.gl[L] = cw.c.g0;
    cw.gl[0] = 1.0;
#define WX_GOLL(NSS, LL)                                                                                             \
    if (wx_lat_stages(filt.F) == NSS && L == LL) {                                                                              \
        if (inverse)                                                                                                 \
            hipLaunchKernelGGL((k_lat_iwpt_lo_f64<NSS, 2, LL>), dim3((unsigned)batch), dim3(64), 0, st, x, y, in_stride, cw); \
        else                                                                                                         \
            hipLaunchKernelGGL((k_lat_wpt_lo_f64<NSS, 2, LL>), dim3((unsigned)batch), dim3(64), 0, st, x, y, cw);     \
    }
#define WX_GOL(NSS) WX_GOLL(NSS, 1) WX_GOLL(NSS, 2) WX_GOLL(NSS, 3) WX_GOLL(NSS, 4) WX_GOLL(NSS, 5)
    if (filt.F / 2 < 1 || filt.F / 2 > WX_LAT_MAXS) return 0;
    WX_GOL(1) WX_GOL(2) WX_GOL(4) WX_GOL(6) WX_GOL(8) WX_GOL(10)
#undef WX_GOLL
#undef WX_GOL
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpt launch (shallow trees)", __FILE__, __LINE__);
    return 1;
}
