#pragma once
// wx_lattice_rows.h -- the ROW pass of the 2-D full-tree packet transforms (images of 128 ... 1024 columns) on the interleaved lattice
// kernels of wx_lattice_dev.h.  Reference: the second half of the 2-D dwt_step! / idwt_step! of every node (dwt/dwt_one_level.jl:319-354,
// 401-436) as applied by the 2-D wpt / iwpt by level (DWT.jl:500-548, 662-710); a full tree is separable, so the L row
// levels of all nodes are one 1-D full-tree transform of every image row.
//
// A wavefront takes 2^SH ADJACENT rows of n = 4096 >> SH columns (Float32: two such sets, lat_f2v) -- in the registers exactly the
// 2^SH interleaved signals of k_lat_wpt_g_f64, in memory the signal number is the contiguous dimension: the routing class lat_isT of
// lat_emit / lat_absorb puts it in the LOW address bits and multiplies the rest by the column stride m.  A 16-element line of the
// exchange is then 2^SH rows x (16 >> SH) columns: 256- and 128-byte runs of Float64 for 128 and 256 columns; 512 and 1024 columns (64- and
// 32-byte runs) run in workgroups of two / four wavefronts on adjacent row groups and only pay for deep trees (policy: wx_lattice_rows.hip).  Until round 5 the row pass ran out of LDS strips only (k_rows_fused,
// wx_dwt2d.hip: LDS-issue bound).  Measured per GiB, column pass + row pass, db4 (profiles/r05_floor2d.txt): 128 x 128 Float64 full depth
// 1.13 -> 0.80 ms, 256 x 256 Float64 1.08 -> 0.85 ms, 128 x 128 Float32 1.25 -> 0.95 ms; Float32 at depth 3 no gain (not taken below 5).
#include "wx_lattice_dev.h"

// W wavefronts per workgroup (512 columns: 2, 1024 columns: 4): they take ADJACENT row groups, so that the 64- / 32-byte runs of a wavefront
// are halves / quarters of 128-byte lines the workgroup's other wavefronts ask for at about the same time on the same CU
template <int SH> struct WxRowsW { static constexpr int value = SH >= 4 ? 1 : (SH == 3 ? 2 : 4); };
template <int NS, int WPE, int SH, typename IO, bool INV>
__global__ __launch_bounds__(64 * WxRowsW<SH>::value) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_rows_g_f64(
    const IO *__restrict__ x, IO *__restrict__ y, int L, unsigned m, unsigned groups, int64_t in_img, int64_t out_img, WxLatW cw)
{
    constexpr int W = WxRowsW<SH>::value;
    __shared__ double lds[W][WX_LAT_LDS];
    const int wv = W > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds[wv];
    const int lane = threadIdx.x & 63;
    typedef typename std::conditional<std::is_same<IO, float>::value, lat_f2v, double>::type V;
    constexpr bool PAIR = lat_vtraits<V>::pair != 0;
    constexpr unsigned per = (PAIR ? 2u : 1u) << SH;                        // rows of a wavefront
    // consecutive workgroup ids go round-robin over the 8 XCDs: give every XCD a contiguous range of row groups, so that the wavefronts
    // whose runs are halves or quarters of the same 128-byte lines (512, 1024 columns) meet in one L2 at about the same time
    const unsigned nq = gridDim.x >> 3, b = blockIdx.x;
    const unsigned w = (b < 8 * nq ? (b & 7) * nq + (b >> 3) : b) * W + (unsigned)wv;
    const unsigned img = w / groups, g = w - img * groups;
    const IO *xs = x + (int64_t)img * in_img + g * per;
    IO *ys = y + (int64_t)img * out_img + g * per;
    const unsigned bofs = PAIR ? (1u << SH) : 0xffffffffu;                  // the second set: the next 2^SH rows
    const typename std::conditional<PAIR, WxLatF, const WxLat &>::type cf = lat_cfsel<NS, PAIR>(cw.c);
    if constexpr (INV) lat_g_inv<NS, SH, 512, IO, V>(xs, ys, L, lds0, lane, cw, cf, m, m, bofs, bofs);
    else lat_g_fwd<NS, SH, 512, IO, V>(xs, ys, L, lds0, lane, cw, cf, m, m, bofs, bofs);
}

// 0 = not applicable (the caller runs k_rows_fused), 1 = launched, < 0 = error
template <typename IO, int SH, int NSMAX, bool INV>
static int wx_lattice_rows_launch(const IO *x, IO *y, int64_t in_img, int64_t out_img, int64_t m, int L, int64_t batch, const WxFilt &filt,
                                  hipStream_t st)
{
    constexpr bool PAIR = std::is_same<IO, float>::value;
    constexpr int64_t per = (int64_t)(PAIR ? 2 : 1) << SH;
    if (L < 1 || L + SH < 6 || L + SH > 12 || filt.F < 2 || (filt.F & 1) || filt.F > 2 * NSMAX) return 0;
    if (PAIR && L < 5) return 0;                                           // shallow Float32 trees: the LDS strips are as fast
    if (m < per || m % per || (m & 3) || (in_img & 3) || (out_img & 3) || m > 0x3fffffff) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return 0;
    if (m * (int64_t)(4096 >> SH) > 0x3fffffff) return 0;                  // 32-bit element offsets inside an image
    constexpr int W = WxRowsW<SH>::value;
    const int64_t groups = m / per, nwave = groups * batch;
    if (batch < 1 || nwave > 0x7fffffff || nwave % W) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, INV, &cw.c)) return 0;
    for (int l = 0; l <= 12; ++l) cw.gl[l] = 0.0;
    cw.gl[L] = cw.c.g0;
    cw.gl[0] = 1.0;
    cw.tail_bsig = 0;
#define WX_GOR(NSS)                                                                                                                  \
    case NSS:                                                                                                                        \
        hipLaunchKernelGGL((k_lat_rows_g_f64<NSS, 2, SH, IO, INV>), dim3((unsigned)(nwave / W)), dim3(64 * W), 0, st, x, y, L, (unsigned)m, \
                           (unsigned)groups, in_img, out_img, cw);                                                                   \
        break;
    switch (wx_lat_stages(filt.F)) {
        WX_GOR(1) WX_GOR(2) WX_GOR(4)
    default:
        if constexpr (NSMAX > 4) {
            switch (wx_lat_stages(filt.F)) {
                WX_GOR(6) WX_GOR(8)
            default: return 0;
            }
        } else
            return 0;
    }
#undef WX_GOR
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice row pass launch", __FILE__, __LINE__);
    return 1;
}

#ifdef WX_ROWS_SH
// included by wx_lattice_rows_{2,3,4,5}{f,i}.hip with WX_ROWS_SH, WX_ROWS_INV and WX_ROWS_FN(type suffix)
int WX_ROWS_FN(f64)(const double *x, double *y, int64_t in_img, int64_t out_img, int64_t m, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    return wx_lattice_rows_launch<double, WX_ROWS_SH, 8, WX_ROWS_INV>(x, y, in_img, out_img, m, L, batch, filt, st);
}
// (the Float32 row kernels -- 128 columns: 0.22 -> 0.28 of the roofline, nothing at depth 3 -- are not built since round 6: they were half of
// this family's 34 CPU-minutes of compile time, profiles/r06_clean_build.txt; -DWX_ROWS_F32=1 brings them back with the dispatch below)
#if WX_ROWS_F32
int WX_ROWS_FN(f32)(const float *x, float *y, int64_t in_img, int64_t out_img, int64_t m, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    return wx_lattice_rows_launch<float, WX_ROWS_SH, 8, WX_ROWS_INV>(x, y, in_img, out_img, m, L, batch, filt, st);
}
#endif
#endif
