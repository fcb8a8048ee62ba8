// wx_lattice_2d64w.hip -- 2-D wpd of 64 x 64 images in ONE pass (DWT.jl:164-209: slice 0 = the image, slice l = the full quad tree of depth l):
// the image is read once, every slice written once -- the algorithmic traffic.  An image is the 4096 slots of one wavefront (Float32: two
// images, lat_f2v), as in wx_lattice_2d64.h.  Slice l = (l row levels) o (l column levels): the column levels accumulate in layout A (one
// more level per slice, the registers `a` stay alive across the slices); for every slice the exchanges T2, T3 copy them to layout C, the l row
// levels run there and lat_emit<6, 256 + l> routes the slice out.  L (L + 1) / 2 row levels instead of L: the price of not re-reading a slice.
// 64 + 64 live 8-byte registers and the exchange temporaries: one wavefront per SIMD (512 registers); the kernel is a stream of
// (L + 1) x 32 KiB stores per 32 KiB load, which is what bounds it.
#include "wx_lattice_dev.h"

namespace {

template <int NS, typename IO>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_lat2d64_wpd(
    const IO *__restrict__ x, IO *__restrict__ y, int L, int last_img, WxLatW cw)
{
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    typedef typename std::conditional<std::is_same<IO, float>::value, lat_f2v, double>::type V;
    constexpr bool PAIR = lat_vtraits<V>::pair != 0;
    const bool lastw = PAIR && blockIdx.x == gridDim.x - 1;
    const int img0 = PAIR ? (lastw ? last_img : (int)(blockIdx.x << 1)) : (int)blockIdx.x;
    const unsigned bimg = (unsigned)(lastw ? cw.tail_bsig : 1);
    const unsigned ts = 4096u * (unsigned)(L + 1);                            // elements of one image's table
    const unsigned bofs_in = PAIR ? bimg * 4096u : 0xffffffffu, bofs = PAIR ? bimg * ts : 0xffffffffu;
    const IO *xs = x + (int64_t)img0 * 4096;
    IO *ys = y + (int64_t)img0 * ts;
    const typename std::conditional<PAIR, WxLatF, const WxLat &>::type cf = lat_cfsel<NS, PAIR>(cw.c);
    V a[64];
    lat_absorb<0, 0>(a, lds0, xs, lane, cw, 4096u, 0, 0, 0, bofs_in);
    lat_emit<0, 0>(a, lds0, ys, lane, cw, 4096u, 0, 0, bofs);                 // slice 0 = the image
#define WX_W_SLICE(LV)                                                                          \
    if (L >= LV) {                                                                              \
        lat_level<LV - 1, 0, NS, false>(a, cf);                                                 \
        V c[64];                                                                                \
        { V bb[64]; lat_t2(a, bb, lds0, lane); lat_t3(bb, c, lds0, lane); }                     \
        lat_level<0, 0, NS, false>(c, cf);                                                      \
        if constexpr (LV > 1) lat_level<1, 0, NS, false>(c, cf);                                \
        if constexpr (LV > 2) lat_level<2, 0, NS, false>(c, cf);                                \
        if constexpr (LV > 3) lat_level<3, 0, NS, false>(c, cf);                                \
        if constexpr (LV > 4) lat_level<4, 0, NS, false>(c, cf);                                \
        if constexpr (LV > 5) lat_level<5, 0, NS, false>(c, cf);                                \
        lat_emit<6, 256 + LV>(c, lds0, ys + 4096 * LV, lane, cw, 4096u, 0, 0, bofs);            \
    }
    WX_W_SLICE(1) WX_W_SLICE(2) WX_W_SLICE(3) WX_W_SLICE(4) WX_W_SLICE(5) WX_W_SLICE(6)
#undef WX_W_SLICE
}

template <typename IO>
int wx_lattice_2d64_wpd_T(const IO *x, IO *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    constexpr bool PAIR = std::is_same<IO, float>::value;
    if (L < 1 || L > 6 || filt.F < 2 || (filt.F & 1) || filt.F > 8 || batch < 1 || batch > 0x3fffffff || (const void *)x == (const void *)y) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, false, &cw.c)) return 0;
    {
        WxLat one;
        if (!wx_lattice_factor(filt, 1, false, &one)) return 0;
        const long double g = one.g0;                           // product of the cosines of one level: slice l carries g^(2 l)
        long double acc = 1;
        for (int l = 0; l <= 12; ++l) { cw.gl[l] = (double)acc; acc *= g; }
    }
    WxPairPlan pp;
    if (PAIR) {
        if (!wx_lat_pair_plan(batch, 0, false, &pp)) return 0;
    } else {
        pp.nwave = (unsigned)batch; pp.tail_sig = (int)(batch - 1); pp.tail_bsig = 0;
    }
    cw.tail_bsig = pp.tail_bsig;
#define WX_GOW(NSS)                                                                                                                  \
    case NSS: hipLaunchKernelGGL((k_lat2d64_wpd<NSS, IO>), dim3(pp.nwave), dim3(64), 0, st, x, y, L, pp.tail_sig, cw); break;
    switch (wx_lat_stages(filt.F)) {
        WX_GOW(1) WX_GOW(2) WX_GOW(3) WX_GOW(4)
    default: return 0;
    }
#undef WX_GOW
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice 2-D wpd launch (64 x 64 images)", __FILE__, __LINE__);
    return 1;
}

}  // namespace

// 0 = not applicable (the caller goes level by level), 1 = launched, < 0 = error
int wx_lattice_2d64_wpd_f64(const double *x, double *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    return wx_lattice_2d64_wpd_T<double>(x, y, L, batch, filt, st);
}
int wx_lattice_2d64_wpd_f32(const float *x, float *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    return wx_lattice_2d64_wpd_T<float>(x, y, L, batch, filt, st);
}
