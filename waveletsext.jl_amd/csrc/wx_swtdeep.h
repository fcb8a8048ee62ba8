#pragma once
// wx_swtdeep.h -- the last levels of the stationary / autocorrelation packet transforms (swpt, acwpt) for any filter, in
// the registers of a lane.
//
// Reference (paths relative to /root/reference/src/mod): sdwt_step! swt/swt_one_level.jl:99-127 (a[i] = sum_j q[j] v[i + (j-1) s],
// d[i] = sum_j (-1)^j q[j] v[i - j s], s = 2^d), acdwt_step! acwt/acwt_one_level.jl (w1 = v/sqrt2 + S, w2 = v/sqrt2 - S, S over the
// odd lags), swpt! SWT.jl:439-472 / acwpt! ACWT.jl:427-460 (the (n, 2^L) table, children overwrite the parent column).
//
// From depth D0 = log2(n) - 4 on a level of dilation 2^d only moves inside a residue class mod 2^D0, and a class has
// n' = 16 samples: the whole subtree below (node q of depth D0, class r) is a lane-local problem.  A wavefront takes 64
// consecutive classes of one node -- every load and every store is one contiguous 512-byte run -- each lane reads its 16
// samples once and walks the packet tree depth first: a node's two children are computed together (16 x 2 F multiply-adds with
// compile-time register indices, the taps wrapping inside the 16 samples), the detail child waits in registers while the
// approximation child's subtree is finished, leaves are stored as soon as they exist.  No LDS, no barrier; the level
// kernels of wx_swt1d.hip, which read F taps per output from LDS and run one or two levels per pass over the table, stop
// at depth D0 (1/16 of the final volume for a full-depth transform).  Same tap order and arithmetic as k_swt_fwd_level.
#include "wx_common.h"
#include "wx_kernels.h"
#include "wx_host.h"
#include <cstdlib>

namespace {

#ifndef WX_SD_LOG
#define WX_SD_LOG 4
#endif
constexpr int SD_LOG = WX_SD_LOG, SD_NP = 1 << SD_LOG;      // samples per residue class

// children of one node at class-local dilation t = 2^J
template <int F, bool AC, int J>
__device__ __forceinline__ void sd_split(const double (&v)[SD_NP], double (&a)[SD_NP], double (&d)[SD_NP], const WxFilt &filt,
                                         const WxAcFilt &ac)
{
    constexpr int t = 1 << J;
#pragma unroll
    for (int m = 0; m < SD_NP; ++m) {
        if constexpr (!AC) {
            double sa = 0.0, sd = 0.0;
#pragma unroll
            for (int j = 0; j < F; ++j) {
                sa = fma(filt.q[j], v[(m + (j - 1) * t) & (SD_NP - 1)], sa);
                sd = fma((j & 1) ? -filt.q[j] : filt.q[j], v[(m - j * t) & (SD_NP - 1)], sd);
            }
            a[m] = sa;
            d[m] = sd;
        } else {
            double S = 0.0;
#pragma unroll
            for (int l = 1; l < F; l += 2)
                S = fma(ac.b[l - 1], v[(m - l * t) & (SD_NP - 1)] + v[(m + l * t) & (SD_NP - 1)], S);
            const double c = ac.c1 * v[m];
            a[m] = c + S;
            d[m] = c - S;
        }
    }
}

template <int F, bool AC, int J, int LP>
__device__ __forceinline__ void sd_node(const double (&v)[SD_NP], double *__restrict__ col, int64_t n, int64_t pstride, const WxFilt &filt,
                                        const WxAcFilt &ac)
{
    double a[SD_NP], d[SD_NP];
    sd_split<F, AC, J>(v, a, d, filt, ac);
    double *hi = col + ((int64_t)(1 << (LP - J - 1))) * n;             // the detail child sits half the node's width further
    if constexpr (J + 1 == LP) {
#pragma unroll
        for (int m = 0; m < SD_NP; ++m) col[m * pstride] = a[m];
#pragma unroll
        for (int m = 0; m < SD_NP; ++m) hi[m * pstride] = d[m];
    } else {
        sd_node<F, AC, J + 1, LP>(a, col, n, pstride, filt, ac);
        sd_node<F, AC, J + 1, LP>(d, hi, n, pstride, filt, ac);
    }
}

// swpd / acwpd (heap-ordered table, every node kept): tab = column 0 of the signal's table, node (depth, index p) lives in
// column 2^depth - 1 + p (SWT.jl:840-868); both children are stored when they exist, then the walk goes on
template <int F, bool AC, int J, int LP>
__device__ __forceinline__ void sd_node_wpd(const double (&v)[SD_NP], double *__restrict__ tab, int64_t n, int64_t pstride, int depth, int64_t p,
                                            const WxFilt &filt, const WxAcFilt &ac)
{
    double a[SD_NP], d[SD_NP];
    sd_split<F, AC, J>(v, a, d, filt, ac);
    double *lo = tab + ((((int64_t)1) << (depth + 1)) - 1 + 2 * p) * n;
    double *hi = lo + n;
#pragma unroll
    for (int m = 0; m < SD_NP; ++m) lo[m * pstride] = a[m];
#pragma unroll
    for (int m = 0; m < SD_NP; ++m) hi[m * pstride] = d[m];
    if constexpr (J + 1 < LP) {
        sd_node_wpd<F, AC, J + 1, LP>(a, tab, n, pstride, depth + 1, 2 * p, filt, ac);
        sd_node_wpd<F, AC, J + 1, LP>(d, tab, n, pstride, depth + 1, 2 * p + 1, filt, ac);
    }
}

template <int F, bool AC, int LP>
__global__ __launch_bounds__(64) void k_swpd_deep_fwd(double *__restrict__ xw, int log2n, int L, int64_t batch, WxFilt filt, WxAcFilt ac)
{
    const int D0 = log2n - SD_LOG;
    const int64_t n = (int64_t)1 << log2n, pstride = (int64_t)1 << D0;
    const int64_t ncols = ((int64_t)1 << (L + 1)) - 1;
    const int cblocks = 1 << (D0 - 6);
    const int q = blockIdx.x / cblocks, cb = blockIdx.x - q * cblocks;
    const int r = cb * 64 + threadIdx.x;
    for (int64_t sig = blockIdx.y; sig < batch; sig += gridDim.y) {
        double *tab = xw + sig * ncols * n + r;
        const double *col = tab + ((((int64_t)1) << D0) - 1 + q) * n;
        double v[SD_NP];
#pragma unroll
        for (int m = 0; m < SD_NP; ++m) v[m] = col[m * pstride];
        sd_node_wpd<F, AC, 0, LP>(v, tab, n, pstride, D0, q, filt, ac);
    }
}

// xw: (n, 2^L, batch) wpt layout; node q of depth D0 = L - LP lives in column q 2^LP
template <int F, bool AC, int LP>
__global__ __launch_bounds__(64) void k_swpt_deep_fwd(double *__restrict__ xw, int log2n, int L, int64_t batch, WxFilt filt, WxAcFilt ac)
{
    const int D0 = log2n - SD_LOG;
    const int64_t n = (int64_t)1 << log2n, pstride = (int64_t)1 << D0;
    const int cblocks = 1 << (D0 - 6);                                   // blocks of 64 classes per node
    const int q = blockIdx.x / cblocks, cb = blockIdx.x - q * cblocks;
    const int r = cb * 64 + threadIdx.x;
    for (int64_t sig = blockIdx.y; sig < batch; sig += gridDim.y) {
        double *col = xw + (sig << L) * n + ((int64_t)q << LP) * n + r;
        double v[SD_NP];
#pragma unroll
        for (int m = 0; m < SD_NP; ++m) v[m] = col[m * pstride];
        sd_node<F, AC, 0, LP>(v, col, n, pstride, filt, ac);
    }
}

// ---- inverse (average-based iswpt): the mirror walk.  The average of the two shift variants of a stationary synthesis step
// is the adjoint of the analysis step, parent[p] = 1/2 sum_j q[j] lo[p + (1 - j) s] + (-1)^j q[j] hi[p + j s]
// (swt/swt_one_level.jl:257-318; wx_swt1d.hip uses the same identity for its fused passes)
template <int F, int J>
__device__ __forceinline__ void sd_merge(double (&out)[SD_NP], const double (&lo)[SD_NP], const double (&hi)[SD_NP], const WxFilt &filt)
{
    constexpr int t = 1 << J;
#pragma unroll
    for (int m = 0; m < SD_NP; ++m) {
        double sa = 0.0, sd = 0.0;
#pragma unroll
        for (int j = 0; j < F; ++j) {
            sa = fma(filt.q[j], lo[(m + (1 - j) * t) & (SD_NP - 1)], sa);
            sd = fma((j & 1) ? -filt.q[j] : filt.q[j], hi[(m + j * t) & (SD_NP - 1)], sd);
        }
        out[m] = (sa + sd) * 0.5;
    }
}

template <int F, int J, int LP>
__device__ __forceinline__ void sd_inode(double (&out)[SD_NP], const double *__restrict__ col, int64_t n, int64_t pstride, const WxFilt &filt)
{
    double lo[SD_NP], hi[SD_NP];
    const double *hcol = col + ((int64_t)(1 << (LP - J - 1))) * n;
    if constexpr (J + 1 == LP) {
#pragma unroll
        for (int m = 0; m < SD_NP; ++m) lo[m] = col[m * pstride];
#pragma unroll
        for (int m = 0; m < SD_NP; ++m) hi[m] = hcol[m * pstride];
    } else {
        sd_inode<F, J + 1, LP>(lo, col, n, pstride, filt);
        sd_inode<F, J + 1, LP>(hi, hcol, n, pstride, filt);
    }
    sd_merge<F, J>(out, lo, hi, filt);
}

// src: (n, src_cols, batch) leaves in wpt order (leaf q 2^LP + j in that column); dst: (n, dst_cols, batch), node q -> column q
template <int F, int LP>
__global__ __launch_bounds__(64) void k_swpt_deep_inv(const double *__restrict__ src, int64_t src_cols, double *__restrict__ dst,
                                                      int64_t dst_cols, int log2n, int64_t batch, WxFilt filt)
{
    const int D0 = log2n - SD_LOG;
    const int64_t n = (int64_t)1 << log2n, pstride = (int64_t)1 << D0;
    const int cblocks = 1 << (D0 - 6);
    const int q = blockIdx.x / cblocks, cb = blockIdx.x - q * cblocks;
    const int r = cb * 64 + threadIdx.x;
    for (int64_t sig = blockIdx.y; sig < batch; sig += gridDim.y) {
        const double *col = src + sig * src_cols * n + ((int64_t)q << LP) * n + r;
        double out[SD_NP];
        sd_inode<F, 0, LP>(out, col, n, pstride, filt);
        double *o = dst + sig * dst_cols * n + (int64_t)q * n + r;
#pragma unroll
        for (int m = 0; m < SD_NP; ++m) o[m * pstride] = out[m];
    }
}

typedef void (*sd_ikern)(const double *, int64_t, double *, int64_t, int, int64_t, WxFilt);
template <int F> sd_ikern sd_ipick(int LP)
{
    switch (LP) {
    case 1: return k_swpt_deep_inv<F, 1>;
    case 2: return k_swpt_deep_inv<F, 2>;
    case 3: return k_swpt_deep_inv<F, 3>;
    default: return k_swpt_deep_inv<F, 4>;
    }
}

typedef void (*sd_kern)(double *, int, int, int64_t, WxFilt, WxAcFilt);
template <int F, bool AC, bool WPD> sd_kern sd_pick(int LP)
{
    if constexpr (WPD)
        switch (LP) {
        case 1: return k_swpd_deep_fwd<F, AC, 1>;
        case 2: return k_swpd_deep_fwd<F, AC, 2>;
        case 3: return k_swpd_deep_fwd<F, AC, 3>;
        default: return k_swpd_deep_fwd<F, AC, 4>;
        }
    else switch (LP) {
    case 1: return k_swpt_deep_fwd<F, AC, 1>;
    case 2: return k_swpt_deep_fwd<F, AC, 2>;
    case 3: return k_swpt_deep_fwd<F, AC, 3>;
    default: return k_swpt_deep_fwd<F, AC, 4>;
    }
}

}  // namespace
