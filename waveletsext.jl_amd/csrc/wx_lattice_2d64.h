#pragma once
// wx_lattice_2d64.h -- 2-D full-tree packets of 64 x 64 images, ONE pass: an image is the 4096 slots of one wavefront (Float32: two images
// per wavefront, one v_pk_fma_f32 per rotation on the pair, as in wx_lattice_sg32.h).  Reference: 2-D wpt / iwpt by level
// (DWT.jl:500-548, 662-710 over dwt/dwt_one_level.jl:319-354, 401-436); a full tree of depth L is separable: L levels
// down every column, then L levels along every row.
//
// The image is column-major, so slot p = row + 64 col.  Layout A of wx_lattice_dev.h (register = p[5:0], lane = p[11:6]) holds one
// COLUMN per lane: the column levels are the lane-local levels lat_level<K, 0> on register bits 0 .. L-1 (64-sample periodic sequences,
// no neighbour lanes).  The exchanges T2, T3 of the 1-D kernel lead to layout C (register = p[11:6], lane = p[5:0]): one ROW per lane,
// and the row levels are the same lane-local levels again.  lat_emit routes the result with the bit map of level code 256 + L
// (lat_obit: each half of the address is a 64-sample packet order of its own) and applies the 2 L gains.  The inverse runs backwards.
// Until round 5 these images took the generic two-pass path (a column kernel and a row kernel through an intermediate image in HBM:
// 0.24-0.27 of the roofline of the one-pass byte count).
#include "wx_lattice_dev.h"

template <int NS, int WPE, typename IO, bool INV>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat2d64_f64(
    const IO *__restrict__ x, IO *__restrict__ y, int L, int last_img, unsigned in_img, WxLatW cw)
{
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    typedef typename std::conditional<std::is_same<IO, float>::value, lat_f2v, double>::type V;
    constexpr bool PAIR = lat_vtraits<V>::pair != 0;
    // pair kernels: last_img is the tail wavefront's first image, cw.tail_bsig the distance of its second (wx_lat_pair_plan with SH = 0)
    const bool lastw = PAIR && blockIdx.x == gridDim.x - 1;
    const int img0 = PAIR ? (lastw ? last_img : (int)(blockIdx.x << 1)) : (int)blockIdx.x;
    // in_img: elements between the images of the input (the inverse reads the deepest slice of a packet table: iwpd of a full tree)
    const unsigned bimg = (unsigned)(lastw ? cw.tail_bsig : 1);
    const unsigned bofs = PAIR ? bimg * 4096u : 0xffffffffu, bofs_in = PAIR ? bimg * in_img : 0xffffffffu;
    const IO *xs = x + (int64_t)img0 * in_img;
    IO *ys = y + (int64_t)img0 * 4096;
    const typename std::conditional<PAIR, WxLatF, const WxLat &>::type cf = lat_cfsel<NS, PAIR>(cw.c);
    if constexpr (!INV) {
        V c[64];
        {
            V a[64], bb[64];
            lat_absorb<0, 0>(a, lds0, xs, lane, cw, 4096u, 0, 0, 0, bofs_in);
            lat_level<0, 0, NS, false>(a, cf);
            if (L > 1) lat_level<1, 0, NS, false>(a, cf);
            if (L > 2) lat_level<2, 0, NS, false>(a, cf);
            if (L > 3) lat_level<3, 0, NS, false>(a, cf);
            if (L > 4) lat_level<4, 0, NS, false>(a, cf);
            if (L > 5) lat_level<5, 0, NS, false>(a, cf);
            lat_t2(a, bb, lds0, lane);
            lat_t3(bb, c, lds0, lane);
        }
        lat_level<0, 0, NS, false>(c, cf);
        if (L > 1) lat_level<1, 0, NS, false>(c, cf);
        if (L > 2) lat_level<2, 0, NS, false>(c, cf);
        if (L > 3) lat_level<3, 0, NS, false>(c, cf);
        if (L > 4) lat_level<4, 0, NS, false>(c, cf);
        if (L > 5) lat_level<5, 0, NS, false>(c, cf);
        switch (L) {
        case 1: lat_emit<6, 256 + 1>(c, lds0, ys, lane, cw, 4096u, 0, 0, bofs); break;
        case 2: lat_emit<6, 256 + 2>(c, lds0, ys, lane, cw, 4096u, 0, 0, bofs); break;
        case 3: lat_emit<6, 256 + 3>(c, lds0, ys, lane, cw, 4096u, 0, 0, bofs); break;
        case 4: lat_emit<6, 256 + 4>(c, lds0, ys, lane, cw, 4096u, 0, 0, bofs); break;
        case 5: lat_emit<6, 256 + 5>(c, lds0, ys, lane, cw, 4096u, 0, 0, bofs); break;
        default: lat_emit<6, 256 + 6>(c, lds0, ys, lane, cw, 4096u, 0, 0, bofs); break;
        }
    } else {
        V a[64];
        {
            V c[64], bb[64];
            switch (L) {
            case 1: lat_absorb<6, 256 + 1>(c, lds0, xs, lane, cw, 4096u, 0, 0, 0, bofs_in); break;
            case 2: lat_absorb<6, 256 + 2>(c, lds0, xs, lane, cw, 4096u, 0, 0, 0, bofs_in); break;
            case 3: lat_absorb<6, 256 + 3>(c, lds0, xs, lane, cw, 4096u, 0, 0, 0, bofs_in); break;
            case 4: lat_absorb<6, 256 + 4>(c, lds0, xs, lane, cw, 4096u, 0, 0, 0, bofs_in); break;
            case 5: lat_absorb<6, 256 + 5>(c, lds0, xs, lane, cw, 4096u, 0, 0, 0, bofs_in); break;
            default: lat_absorb<6, 256 + 6>(c, lds0, xs, lane, cw, 4096u, 0, 0, 0, bofs_in); break;
            }
            if (L > 5) lat_level<5, 0, NS, true>(c, cf);
            if (L > 4) lat_level<4, 0, NS, true>(c, cf);
            if (L > 3) lat_level<3, 0, NS, true>(c, cf);
            if (L > 2) lat_level<2, 0, NS, true>(c, cf);
            if (L > 1) lat_level<1, 0, NS, true>(c, cf);
            lat_level<0, 0, NS, true>(c, cf);
            lat_t3i(c, bb, lds0, lane);
            lat_t2i(bb, a, lds0, lane);
        }
        if (L > 5) lat_level<5, 0, NS, true>(a, cf);
        if (L > 4) lat_level<4, 0, NS, true>(a, cf);
        if (L > 3) lat_level<3, 0, NS, true>(a, cf);
        if (L > 2) lat_level<2, 0, NS, true>(a, cf);
        if (L > 1) lat_level<1, 0, NS, true>(a, cf);
        lat_level<0, 0, NS, true>(a, cf);
        lat_emit<0, 0>(a, lds0, ys, lane, cw, 4096u, 0, 0, bofs);
    }
}

// 0 = not applicable (the caller goes on to the two-pass path), 1 = launched, < 0 = error
template <typename IO, int NSMAX, bool INV>
static int wx_lattice_2d64_launch(const IO *x, IO *y, int L, int64_t batch, int64_t in_img, const WxFilt &filt, hipStream_t st)
{
    constexpr bool PAIR = std::is_same<IO, float>::value;
    if (L < 1 || L > 6 || filt.F < 2 || (filt.F & 1) || filt.F > 2 * NSMAX || batch < 1 || batch > 0x3fffffff) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return 0;
    if (in_img < 4096 || (in_img & 3) || in_img > 0x3fffffff) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, 2 * L, INV, &cw.c)) return 0;   // 2 L levels lie between an image and a coefficient: g0 = g^(2 L)
    for (int l = 0; l <= 12; ++l) cw.gl[l] = 0.0;
    cw.gl[2 * L] = cw.c.g0;
    cw.gl[0] = 1.0;
    WxPairPlan pp;
    if (PAIR) {
        if (!wx_lat_pair_plan(batch, 0, x == y, &pp)) return 0;
    } else {
        pp.nwave = (unsigned)batch; pp.tail_sig = (int)(batch - 1); pp.tail_bsig = 0;
    }
    cw.tail_bsig = pp.tail_bsig;
    // Float64: 166-168 registers, three wavefronts per SIMD without a spill; the Float32 pair kernels need the 256 of two
    // (the two-wavefront build of the Float64 kernels was reachable through WX_2D64_WPE=2 only: not built since round 6)
#define WX_GO2(NSS)                                                                                                                  \
    case NSS:                                                                                                                        \
        hipLaunchKernelGGL((k_lat2d64_f64<NSS, PAIR ? 2 : 3, IO, INV>), dim3(pp.nwave), dim3(64), 0, st, x, y, L, pp.tail_sig, (unsigned)in_img, cw); \
        break;
    switch (wx_lat_stages(filt.F)) {
        WX_GO2(1) WX_GO2(2) WX_GO2(4)
    default:
        if constexpr (NSMAX > 4) {
            switch (wx_lat_stages(filt.F)) {
                WX_GO2(6) WX_GO2(8)
            default: return 0;
            }
        } else
            return 0;
    }
#undef WX_GO2
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice 2-D launch (64 x 64 images)", __FILE__, __LINE__);
    return 1;
}
