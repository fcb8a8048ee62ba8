// wx_dwttail.hip -- the deep levels of the wavelet pyramid (dwt / dwtall: the tree whose only decomposed nodes are the
// approximations) in the registers of a lane.
//
// Reference: dwt! of Wavelets.jl as called by dwtall (dwt/dwt_all.jl:39-54) = wpt with maketree(n, L, :dwt); one level is
// dwt_step! (dwt/dwt_one_level.jl:79-107): a[t] = sum_k q[k] v[(2t + k) mod m], d[t] = sum_k (-1)^k q[k] v[(2t + 1 - k) mod m].
//
// In the fused LDS kernels (wx_dwt1d.hip) every level of a tree is a dependent chain with a wavefront-level fence, and below
// 512 samples one wavefront of the workgroup runs it while the others wait: 0.07 ms per level and launch for 65536 signals,
// although the levels from 64 samples down hold 1.6 % of the data.  Here the tree-driven kernel stops at the depth where the
// approximation has 64 samples, and one LANE per signal finishes the pyramid: 64 values in registers, every level unrolled
// with compile-time indices (the taps wrap inside the node by masking), results written back over the 64 values as
// [a_L | d_L | ... ] -- the layout of the pyramid.  33 MB of traffic for 65536 signals of 4096 Float64 samples.
#include "wx_common.h"
#include "wx_kernels.h"
#include "wx_host.h"
#include <cstdlib>

namespace {

template <int F, int M>
__device__ __forceinline__ void tail_level(double (&v)[64], const WxFilt &filt)
{
    double a[M / 2], d[M / 2];
#pragma unroll
    for (int t = 0; t < M / 2; ++t) {
        double sa = 0.0, sd = 0.0;
#pragma unroll
        for (int k = 0; k < F; ++k) {
            sa = fma(filt.q[k], v[(2 * t + k) & (M - 1)], sa);
            sd = fma((k & 1) ? -filt.q[k] : filt.q[k], v[(2 * t + 1 - k) & (M - 1)], sd);
        }
        a[t] = sa; d[t] = sd;
    }
#pragma unroll
    for (int t = 0; t < M / 2; ++t) { v[t] = a[t]; v[M / 2 + t] = d[t]; }
}

// y: (n, batch), the first 64 samples of every signal hold the approximation of depth log2(n) - 6; Lt = 1 .. 6 more levels
template <typename T, int F>
__global__ __launch_bounds__(64) void k_dwt_tail(T *__restrict__ y, int64_t n, int64_t batch, int Lt, WxFilt filt)
{
    const int64_t sig = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (sig >= batch) return;
    T *p = y + sig * n;
    double v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = (double)p[i];
    tail_level<F, 64>(v, filt);
    if (Lt >= 2) tail_level<F, 32>(v, filt);
    if (Lt >= 3) tail_level<F, 16>(v, filt);
    if (Lt >= 4) tail_level<F, 8>(v, filt);
    if (Lt >= 5) tail_level<F, 4>(v, filt);
    if (Lt >= 6) tail_level<F, 2>(v, filt);
#pragma unroll
    for (int i = 0; i < 64; ++i) p[i] = (T)v[i];
}

// idwt_step! (dwt/dwt_one_level.jl:192-223) on [a (M/2) | d (M/2)] -> M samples:
// x[2t] = sum_m q[2m] a[t - m] - q[2m+1] d[t + m], x[2t+1] = sum_m q[2m+1] a[t - m] + q[2m] d[t + m]  (indices mod M/2)
template <int F, int M>
__device__ __forceinline__ void tail_ilevel(double (&v)[64], const WxFilt &filt)
{
    constexpr int H = M / 2;
    double o[M];
#pragma unroll
    for (int t = 0; t < H; ++t) {
        double v0 = 0.0, v1 = 0.0;
#pragma unroll
        for (int m = 0; m < F / 2; ++m) {
            const double av = v[(t - m) & (H - 1)], dv = v[H + ((t + m) & (H - 1))];
            v0 = fma(filt.q[2 * m], av, v0);
            v0 = fma(-filt.q[2 * m + 1], dv, v0);
            v1 = fma(filt.q[2 * m + 1], av, v1);
            v1 = fma(filt.q[2 * m], dv, v1);
        }
        o[2 * t] = v0; o[2 * t + 1] = v1;
    }
#pragma unroll
    for (int i = 0; i < M; ++i) v[i] = o[i];
}

// xw: (n, batch) pyramids of depth log2(n) - 6 + Lt; head: (64, batch) <- the approximation of depth log2(n) - 6
template <typename T, int F>
__global__ __launch_bounds__(64) void k_idwt_tail(const T *__restrict__ xw, T *__restrict__ head, int64_t n, int64_t batch, int Lt,
                                                  WxFilt filt, WxThreshArg thr)
{
    const int64_t sig = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (sig >= batch) return;
    const T *p = xw + sig * n;
    double v[64];
    if (thr.t) {                                     // denoise: the threshold of rows >= thr.lo rides on the load
        const T tt = (T)((double)reinterpret_cast<const T *>(thr.t)[thr.per_signal ? sig : 0] * thr.scale);
#pragma unroll
        for (int i = 0; i < 64; ++i) { const T c = p[i]; v[i] = (double)(i >= thr.lo ? wx_thresh<T>(c, tt, thr.kind) : c); }
    } else {
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = (double)p[i];
    }
    if (Lt >= 6) tail_ilevel<F, 2>(v, filt);
    if (Lt >= 5) tail_ilevel<F, 4>(v, filt);
    if (Lt >= 4) tail_ilevel<F, 8>(v, filt);
    if (Lt >= 3) tail_ilevel<F, 16>(v, filt);
    if (Lt >= 2) tail_ilevel<F, 32>(v, filt);
    tail_ilevel<F, 64>(v, filt);
    T *o = head + sig * 64;
#pragma unroll
    for (int i = 0; i < 64; ++i) o[i] = (T)v[i];
}

// ---- 2-D pyramid: the levels from an 8 x 8 approximation down, one lane per image (dwt_step! 2-D, dwt/dwt_one_level.jl:319-354:
// the 1-D step along dimension 1 of every column, then along dimension 2 of every row; quadrants [aa ad; da dd] in place)
template <int F, int M>
__device__ __forceinline__ void tail2d_level(double (&v)[8][8], const WxFilt &filt)
{
    constexpr int H = M / 2;
    double t[M][M];
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int r = 0; r < H; ++r) {
            double sa = 0.0, sd = 0.0;
#pragma unroll
            for (int k = 0; k < F; ++k) {
                sa = fma(filt.q[k], v[(2 * r + k) & (M - 1)][j], sa);
                sd = fma((k & 1) ? -filt.q[k] : filt.q[k], v[(2 * r + 1 - k) & (M - 1)][j], sd);
            }
            t[r][j] = sa; t[H + r][j] = sd;
        }
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int r = 0; r < H; ++r) {
            double sa = 0.0, sd = 0.0;
#pragma unroll
            for (int k = 0; k < F; ++k) {
                sa = fma(filt.q[k], t[i][(2 * r + k) & (M - 1)], sa);
                sd = fma((k & 1) ? -filt.q[k] : filt.q[k], t[i][(2 * r + 1 - k) & (M - 1)], sd);
            }
            v[i][r] = sa; v[i][H + r] = sd;
        }
}

// y: (m, m, batch) column-major images whose top-left 8 x 8 block is the approximation of depth log2(m) - 3; Lt = 1 .. 3
template <typename T, int F>
__global__ __launch_bounds__(64) void k_dwt2d_tail(T *__restrict__ y, int64_t m, int64_t batch, int Lt, WxFilt filt)
{
    const int64_t sig = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (sig >= batch) return;
    T *p = y + sig * m * m;
    double v[8][8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i][j] = (double)p[i + j * m];
    tail2d_level<F, 8>(v, filt);
    if (Lt >= 2) tail2d_level<F, 4>(v, filt);
    if (Lt >= 3) tail2d_level<F, 2>(v, filt);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i + j * m] = (T)v[i][j];
}

}  // namespace

// 2-D: levels the tail takes off a pyramid of depth L of m x m images (0 = none)
int wx_dwt2d_tail_levels(int64_t m, int64_t n, int L, int F, size_t esz)
{
    static const bool off = wx_getenv("WX_DWT_TAIL") && atoi(wx_getenv("WX_DWT_TAIL")) == 0;
    if (off || (esz != 8 && esz != 4) || m != n || m < 16 || (m & (m - 1))) return 0;
    switch (F) { case 2: case 4: case 6: case 8: case 10: case 12: case 14: case 16: case 18: case 20: break; default: return 0; }
    int log2m = 0;
    while (((int64_t)1 << (log2m + 1)) <= m) ++log2m;
    const int Ls = log2m - 3;
    return L - Ls >= 1 ? L - Ls : 0;
}

template <typename T>
int wx_dwt2d_tail(T *y, int64_t m, int Lt, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    if (batch <= 0) return WX_OK;
    typedef void (*KT)(T *, int64_t, int64_t, int, WxFilt);
    KT k = nullptr;
    switch (filt.F) {
#define WX_TL(FF) case FF: k = k_dwt2d_tail<T, FF>; break;
        WX_TL(2) WX_TL(4) WX_TL(6) WX_TL(8) WX_TL(10) WX_TL(12) WX_TL(14) WX_TL(16) WX_TL(18) WX_TL(20)
#undef WX_TL
        default: return wx_set_error(WX_EHIP, "dwt 2-D tail: unsupported filter length");
    }
    hipLaunchKernelGGL(k, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st, y, m, batch, Lt, filt);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template int wx_dwt2d_tail<double>(double *, int64_t, int, int64_t, const WxFilt &, hipStream_t);
template int wx_dwt2d_tail<float>(float *, int64_t, int, int64_t, const WxFilt &, hipStream_t);

// number of levels the tail takes off the end of a pyramid of depth L (0 = none): the tree-driven kernel then runs L - that
int wx_dwt_tail_levels(int64_t n, int L, int F, size_t esz)
{
    static const bool off = wx_getenv("WX_DWT_TAIL") && atoi(wx_getenv("WX_DWT_TAIL")) == 0;
    if (off || (esz != 8 && esz != 4) || n < 128 || (n & (n - 1))) return 0;
    switch (F) { case 2: case 4: case 6: case 8: case 10: case 12: case 14: case 16: case 18: case 20: break; default: return 0; }
    int log2n = 0;
    while (((int64_t)1 << (log2n + 1)) <= n) ++log2n;
    const int Ls = log2n - 6;
    return L - Ls >= 1 ? L - Ls : 0;
}

template <typename T>
int wx_dwt_tail(T *y, int64_t n, int Lt, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    if (batch <= 0) return WX_OK;
    typedef void (*KT)(T *, int64_t, int64_t, int, WxFilt);
    KT k = nullptr;
    switch (filt.F) {
#define WX_TL(FF) case FF: k = k_dwt_tail<T, FF>; break;
        WX_TL(2) WX_TL(4) WX_TL(6) WX_TL(8) WX_TL(10) WX_TL(12) WX_TL(14) WX_TL(16) WX_TL(18) WX_TL(20)
#undef WX_TL
        default: return wx_set_error(WX_EHIP, "dwt tail: unsupported filter length");
    }
    hipLaunchKernelGGL(k, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st, y, n, batch, Lt, filt);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template int wx_dwt_tail<double>(double *, int64_t, int, int64_t, const WxFilt &, hipStream_t);
template int wx_dwt_tail<float>(float *, int64_t, int, int64_t, const WxFilt &, hipStream_t);

template <typename T>
int wx_idwt_tail(const T *xw, T *head, int64_t n, int Lt, int64_t batch, const WxFilt &filt, const WxThreshArg &thr, hipStream_t st)
{
    if (batch <= 0) return WX_OK;
    typedef void (*KT)(const T *, T *, int64_t, int64_t, int, WxFilt, WxThreshArg);
    KT k = nullptr;
    switch (filt.F) {
#define WX_TL(FF) case FF: k = k_idwt_tail<T, FF>; break;
        WX_TL(2) WX_TL(4) WX_TL(6) WX_TL(8) WX_TL(10) WX_TL(12) WX_TL(14) WX_TL(16) WX_TL(18) WX_TL(20)
#undef WX_TL
        default: return wx_set_error(WX_EHIP, "idwt tail: unsupported filter length");
    }
    hipLaunchKernelGGL(k, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st, xw, head, n, batch, Lt, filt, thr);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template int wx_idwt_tail<double>(const double *, double *, int64_t, int, int64_t, const WxFilt &, const WxThreshArg &, hipStream_t);
template int wx_idwt_tail<float>(const float *, float *, int64_t, int, int64_t, const WxFilt &, const WxThreshArg &, hipStream_t);
