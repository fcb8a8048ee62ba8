// wx_dwt1d.hip -- batched 1-D decimated wavelet-packet kernels for gfx950 (MI355X).
//
// Reference semantics (paths relative to /root/reference/src/mod):
//   dwt_step!   dwt/dwt_one_level.jl:79-107      a[i] = sum_k q[k] v[(2i+k) mod n]
//                                                d[i] = sum_k (-1)^k q[k] v[(2i+1-k) mod n]
//   idwt_step!  dwt/dwt_one_level.jl:192-223     v[2k]   = sum_m q[2m] a[k-m] - q[2m+1] d[k+m]
//                                                v[2k+1] = sum_m q[2m+1] a[k-m] + q[2m] d[k+m]
//   wpd!        DWT.jl:131-161 (all levels kept, (n, L+1) table per signal)
//   wpt!/iwpt!  Wavelets.jl 1-D (leaves of the tree), call sites dwt/dwt_all.jl:162,221
//   iwpd!       DWT.jl:340-351 (getbasiscoef gather, Utils.jl:101-134, then iwpt!)
//
// Two kernel families:
//   * fused: one workgroup per signal, the whole signal LDS-resident (ping-pong pair),
//     every tree level computed on chip; HBM sees the input once and each output once.
//     Forward levels keep the parent as an even/odd split (E[k]=v[2k], O[k]=v[2k+1]) so both
//     QMF branches are unit-stride convolutions and each lane reads its tap window with
//     conflict-free 16-byte LDS loads (two outputs per branch per lane).
//   * generic: one level per launch straight from/to HBM, true modulo wrap; used for
//     n = odd * 2^k, signals that do not fit LDS, and filters without a fused instantiation.
#include "wx_common.h"
#include "wx_kernels.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------
// generic (one level per launch)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_fwd1d_level(const T *__restrict__ src, T *__restrict__ dst,
                                                     int64_t src_stride, int64_t dst_stride, int n,
                                                     int np, int depth, int64_t batch, WxFilt filt,
                                                     const uint8_t *__restrict__ status, int64_t nstatus)
{
    const int h = np >> 1;
    const int64_t total = (int64_t)batch * (n >> 1);
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = g / (n >> 1);
        const int i = (int)(g - b * (n >> 1));
        const int j = i / h, t = i - j * h;
        const T *v = src + b * src_stride + (int64_t)j * np;
        T *o = dst + b * dst_stride + (int64_t)j * np;
        bool active = true;
        if (status) {
            const int64_t node = ((int64_t)1 << depth) + j;
            active = node <= nstatus && status[node - 1];
        }
        if (!active) {
            o[2 * t] = v[2 * t];
            o[2 * t + 1] = v[2 * t + 1];
            continue;
        }
        double a = 0.0, d = 0.0;
        int k1 = wx_modn(2 * t, np), k2 = wx_modn(2 * t + 1, np);
        for (int k = 0; k < filt.F; ++k) {
            a = fma(filt.q[k], (double)v[k1], a);
            d = fma((k & 1) ? -filt.q[k] : filt.q[k], (double)v[k2], d);
            k1 = k1 + 1 == np ? 0 : k1 + 1;
            k2 = k2 == 0 ? np - 1 : k2 - 1;
        }
        o[t] = (T)a;
        o[h + t] = (T)d;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_inv1d_level(const T *__restrict__ src, T *__restrict__ dst,
                                                     int64_t src_stride, int64_t dst_stride, int n,
                                                     int np, int depth, int64_t batch, WxFilt filt,
                                                     const uint8_t *__restrict__ status, int64_t nstatus)
{
    const int h = np >> 1;
    const int64_t total = (int64_t)batch * (n >> 1);
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = g / (n >> 1);
        const int i = (int)(g - b * (n >> 1));
        const int j = i / h, k = i - j * h;
        const T *a = src + b * src_stride + (int64_t)j * np;
        const T *dd = a + h;
        T *o = dst + b * dst_stride + (int64_t)j * np;
        bool active = true;
        if (status) {
            const int64_t node = ((int64_t)1 << depth) + j;
            active = node <= nstatus && status[node - 1];
        }
        if (!active) {
            o[2 * k] = a[2 * k];
            o[2 * k + 1] = a[2 * k + 1];
            continue;
        }
        double v0 = 0.0, v1 = 0.0;
        int k1 = k, k2 = k;
        for (int m = 0; m < filt.F / 2; ++m) {
            const double av = (double)a[k1], dv = (double)dd[k2];
            v0 = fma(filt.q[2 * m], av, v0);
            v0 = fma(-filt.q[2 * m + 1], dv, v0);
            v1 = fma(filt.q[2 * m + 1], av, v1);
            v1 = fma(filt.q[2 * m], dv, v1);
            k1 = k1 == 0 ? h - 1 : k1 - 1;
            k2 = k2 + 1 == h ? 0 : k2 + 1;
        }
        o[2 * k] = (T)v0;
        o[2 * k + 1] = (T)v1;
    }
}

// getbasiscoef gather (Utils.jl:117-131): out[p] = Xw[p, col(p)] for every signal
template <typename T>
__global__ __launch_bounds__(256) void k_gather_leaves1d(const T *__restrict__ Xw, T *__restrict__ out,
                                                         int n, int k, int64_t batch,
                                                         const int *__restrict__ colmap, int blk)
{
    const int64_t total = (int64_t)batch * n;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = g / n;
        const int p = (int)(g - b * n);
        const int c = colmap[p / blk];
        out[g] = Xw[(b * k + c) * (int64_t)n + p];
    }
}

// ------------------------------------------------------------------------------------------
// fused forward: wpd (WRITE_ALL) / wpt
// ------------------------------------------------------------------------------------------
template <typename T, int F, int NT, bool WRITE_ALL>
__global__ __launch_bounds__(NT) void k_fwd1d_fused(const T *__restrict__ x, T *__restrict__ y,
                                                    int log2n, int L, int64_t batch, int64_t x_stride,
                                                    int64_t y_stride, WxFilt filt,
                                                    const uint8_t *__restrict__ status, int64_t nstatus)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    typedef typename WxVec2<T>::type V2;
    constexpr int HF = F / 2;
    constexpr int BACK = (HF & 1) ? HF - 1 : HF;
    constexpr int NP = BACK + 1;                 // 16-byte (f64) / 8-byte (f32) pairs per window
    const int n = 1 << log2n;
    const int half = n >> 1;
    T *buf0 = reinterpret_cast<T *>(wx_smem);
    T *buf1 = buf0 + n;

    T q[F];
#pragma unroll
    for (int k = 0; k < F; ++k) q[k] = (T)filt.q[k];

    for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) {
        const T *xs = x + b * x_stride;
        T *ys = y + b * y_stride;
        T *cur = buf0, *nxt = buf1;
        for (int p = threadIdx.x; p < half; p += NT) {
            const V2 v = reinterpret_cast<const V2 *>(xs)[p];
            cur[p] = v.x;
            cur[half + p] = v.y;
            if (WRITE_ALL) reinterpret_cast<V2 *>(ys)[p] = v;
        }
        __syncthreads();
        for (int d = 0; d < L; ++d) {
            const int lh = log2n - d - 1;        // log2(child length)
            const T *E = cur, *O = cur + half;
            T *En = nxt, *On = nxt + half;
            V2 *yl = reinterpret_cast<V2 *>(WRITE_ALL ? ys + (int64_t)(d + 1) * n : ys);
            if (lh >= 1) {
                const int hq = 1 << (lh - 1);
                for (int w = threadIdx.x; w < (n >> 2); w += NT) {
                    const int j = w >> (lh - 1);
                    const int t = w & (hq - 1);
                    const int B = j << lh;
                    if (status) {
                        const int64_t node = ((int64_t)1 << d) + j;
                        if (!(node <= nstatus && status[node - 1])) {
                            reinterpret_cast<V2 *>(En + B)[t] = reinterpret_cast<const V2 *>(E + B)[t];
                            reinterpret_cast<V2 *>(On + B)[t] = reinterpret_cast<const V2 *>(O + B)[t];
                            continue;
                        }
                    }
                    T e[2 * NP], o[2 * NP];
#pragma unroll
                    for (int r = 0; r < NP; ++r) {
                        const int u = (t - BACK / 2 + r) & (hq - 1);
                        const V2 ve = reinterpret_cast<const V2 *>(E + B)[u];
                        const V2 vo = reinterpret_cast<const V2 *>(O + B)[u];
                        e[2 * r] = ve.x; e[2 * r + 1] = ve.y;
                        o[2 * r] = vo.x; o[2 * r + 1] = vo.y;
                    }
                    T a0 = 0, a1 = 0, d0 = 0, d1 = 0;
#pragma unroll
                    for (int m = 0; m < HF; ++m) {
                        a0 = fma(q[2 * m], e[BACK + m], a0);
                        a0 = fma(q[2 * m + 1], o[BACK + m], a0);
                        a1 = fma(q[2 * m], e[BACK + 1 + m], a1);
                        a1 = fma(q[2 * m + 1], o[BACK + 1 + m], a1);
                        d0 = fma(q[2 * m], o[BACK - m], d0);
                        d0 = fma(-q[2 * m + 1], e[BACK - m], d0);
                        d1 = fma(q[2 * m], o[BACK + 1 - m], d1);
                        d1 = fma(-q[2 * m + 1], e[BACK + 1 - m], d1);
                    }
                    En[B + t] = a0; On[B + t] = a1;
                    En[B + hq + t] = d0; On[B + hq + t] = d1;
                    if (WRITE_ALL) {
                        V2 va; va.x = a0; va.y = a1;
                        V2 vd; vd.x = d0; vd.y = d1;
                        yl[B + t] = va;
                        yl[B + hq + t] = vd;
                    }
                }
            } else {                             // parent nodes of length 2
                for (int w = threadIdx.x; w < half; w += NT) {
                    const T ev = E[w], ov = O[w];
                    if (status) {
                        const int64_t node = ((int64_t)1 << d) + w;
                        if (!(node <= nstatus && status[node - 1])) { En[w] = ev; On[w] = ov; continue; }
                    }
                    T a = 0, dd = 0;
#pragma unroll
                    for (int k = 0; k < F; ++k) {
                        a = fma(q[k], (k & 1) ? ov : ev, a);
                        dd = fma((k & 1) ? -q[k] : q[k], (k & 1) ? ev : ov, dd);
                    }
                    En[w] = a; On[w] = dd;
                    if (WRITE_ALL) { V2 v; v.x = a; v.y = dd; yl[w] = v; }
                }
            }
            __syncthreads();
            T *tmp = cur; cur = nxt; nxt = tmp;
        }
        if (!WRITE_ALL) {
            for (int p = threadIdx.x; p < half; p += NT) {
                V2 v; v.x = cur[p]; v.y = cur[half + p];
                reinterpret_cast<V2 *>(ys)[p] = v;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// fused inverse: iwpt / iwpd (leaf gather on load)
// ------------------------------------------------------------------------------------------
template <typename T, int F, int NT>
__global__ __launch_bounds__(NT) void k_inv1d_fused(const T *__restrict__ xw, T *__restrict__ xh,
                                                    int log2n, int L, int64_t batch, int64_t in_stride,
                                                    int64_t out_stride, WxFilt filt,
                                                    const uint8_t *__restrict__ status, int64_t nstatus,
                                                    const int *__restrict__ colmap, int log2blk)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    typedef typename WxVec2<T>::type V2;
    constexpr int HF = F / 2;
    constexpr int BACK = (HF & 1) ? HF - 1 : HF;
    constexpr int NPI = BACK / 2 + 1;            // pairs per child window
    const int n = 1 << log2n;
    const int half = n >> 1;
    T *buf0 = reinterpret_cast<T *>(wx_smem);
    T *buf1 = buf0 + n;

    T q[F];
#pragma unroll
    for (int k = 0; k < F; ++k) q[k] = (T)filt.q[k];

    for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) {
        const T *xs = xw + b * in_stride;
        T *os = xh + b * out_stride;
        T *cur = buf0, *nxt = buf1;
        if (colmap) {                            // packet table (n, k): leaf of depth c lives in column c
            for (int p = threadIdx.x; p < half; p += NT) {
                const int c = colmap[(2 * p) >> log2blk];
                reinterpret_cast<V2 *>(cur)[p] = reinterpret_cast<const V2 *>(xs + (int64_t)c * n)[p];
            }
        } else {
            for (int p = threadIdx.x; p < half; p += NT)
                reinterpret_cast<V2 *>(cur)[p] = reinterpret_cast<const V2 *>(xs)[p];
        }
        __syncthreads();
        bool direct = false;
        for (int d = L - 1; d >= 0; --d) {
            const int lh = log2n - d - 1;
            if (lh >= 1) {
                const int hq = 1 << (lh - 1);
                direct = (d == 0);
                for (int w = threadIdx.x; w < (n >> 2); w += NT) {
                    const int j = w >> (lh - 1);
                    const int t = w & (hq - 1);
                    const int base = j << (lh + 1);
                    V2 *dst = direct ? reinterpret_cast<V2 *>(os) : reinterpret_cast<V2 *>(nxt);
                    if (status) {
                        const int64_t node = ((int64_t)1 << d) + j;
                        if (!(node <= nstatus && status[node - 1])) {
                            const int i0 = (base >> 1) + 2 * t;
                            dst[i0] = reinterpret_cast<const V2 *>(cur)[i0];
                            dst[i0 + 1] = reinterpret_cast<const V2 *>(cur)[i0 + 1];
                            continue;
                        }
                    }
                    const V2 *A = reinterpret_cast<const V2 *>(cur + base);
                    const V2 *D = reinterpret_cast<const V2 *>(cur + base + (1 << lh));
                    T aw[2 * NPI], dw[2 * NPI];
#pragma unroll
                    for (int r = 0; r < NPI; ++r) {
                        const V2 va = A[(t - BACK / 2 + r) & (hq - 1)];
                        const V2 vd = D[(t + r) & (hq - 1)];
                        aw[2 * r] = va.x; aw[2 * r + 1] = va.y;
                        dw[2 * r] = vd.x; dw[2 * r + 1] = vd.y;
                    }
                    T v0 = 0, v1 = 0, v2 = 0, v3 = 0;   // v[4t..4t+3]
#pragma unroll
                    for (int m = 0; m < HF; ++m) {
                        v0 = fma(q[2 * m], aw[BACK - m], v0);
                        v0 = fma(-q[2 * m + 1], dw[m], v0);
                        v1 = fma(q[2 * m + 1], aw[BACK - m], v1);
                        v1 = fma(q[2 * m], dw[m], v1);
                        v2 = fma(q[2 * m], aw[BACK + 1 - m], v2);
                        v2 = fma(-q[2 * m + 1], dw[1 + m], v2);
                        v3 = fma(q[2 * m + 1], aw[BACK + 1 - m], v3);
                        v3 = fma(q[2 * m], dw[1 + m], v3);
                    }
                    V2 lo; lo.x = v0; lo.y = v1;
                    V2 hi; hi.x = v2; hi.y = v3;
                    const int i0 = (base >> 1) + 2 * t;
                    dst[i0] = lo;
                    dst[i0 + 1] = hi;
                }
            } else {                             // children of length 1
                direct = false;
                for (int w = threadIdx.x; w < half; w += NT) {
                    const V2 c = reinterpret_cast<const V2 *>(cur)[w];
                    if (status) {
                        const int64_t node = ((int64_t)1 << d) + w;
                        if (!(node <= nstatus && status[node - 1])) { reinterpret_cast<V2 *>(nxt)[w] = c; continue; }
                    }
                    T v0 = 0, v1 = 0;
#pragma unroll
                    for (int m = 0; m < HF; ++m) {
                        v0 = fma(q[2 * m], c.x, v0);
                        v0 = fma(-q[2 * m + 1], c.y, v0);
                        v1 = fma(q[2 * m + 1], c.x, v1);
                        v1 = fma(q[2 * m], c.y, v1);
                    }
                    V2 v; v.x = v0; v.y = v1;
                    reinterpret_cast<V2 *>(nxt)[w] = v;
                }
            }
            __syncthreads();
            T *tmp = cur; cur = nxt; nxt = tmp;
        }
        if (!direct) {
            for (int p = threadIdx.x; p < half; p += NT)
                reinterpret_cast<V2 *>(os)[p] = reinterpret_cast<const V2 *>(cur)[p];
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
static int wx_grid_for(int64_t total, int block)
{
    int64_t g = (total + block - 1) / block;
    const int64_t cap = 256 * 16;               // 256 CUs x 16 resident blocks of 256 threads
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

static bool wx_is_pow2(int64_t n) { return n >= 1 && (n & (n - 1)) == 0; }
static int wx_log2(int64_t n) { int l = 0; while (((int64_t)1 << (l + 1)) <= n) ++l; return l; }

template <typename T> static size_t wx_fused_lds_bytes(int64_t n) { return (size_t)2 * n * sizeof(T); }

template <typename T> bool wx_fused1d_ok(int64_t n, int F)
{
    if (!wx_is_pow2(n) || n < 2) return false;
    if (wx_fused_lds_bytes<T>(n) > 160 * 1024) return false;
    switch (F) { case 2: case 4: case 6: case 8: case 10: case 12: case 16: case 18: case 20: return true; }
    return false;
}
template bool wx_fused1d_ok<double>(int64_t, int);
template bool wx_fused1d_ok<float>(int64_t, int);

static int wx_fused_grid(size_t lds, int64_t batch, int nt)
{
    int per_cu = (int)((160 * 1024) / (lds ? lds : 1));
    const int by_waves = 2048 / nt;                 // 32 waves per CU
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 16) per_cu = 16;
    int64_t g = (int64_t)256 * per_cu;
    if (g > batch) g = batch;
    return (int)g;
}

// threads per workgroup of the fused kernels: one lane per 4-output item of a level, capped by
// WX_FUSED_NT (tuning knob, default 512: measured best on MI355X)
static int wx_fused_nt(int64_t n)
{
    static int cap = 0;
    if (!cap) {
        const char *e = getenv("WX_FUSED_NT");
        cap = e ? atoi(e) : 512;
        if (cap != 64 && cap != 128 && cap != 256 && cap != 512 && cap != 1024) cap = 512;
    }
    int64_t want = n / 4;
    int nt = 64;
    while (nt < cap && nt < want) nt <<= 1;
    return nt;
}

template <typename K> static hipError_t wx_allow_lds(K kernel, size_t lds)
{
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <typename T, int F, bool WRITE_ALL, int NT>
static int launch_fwd_fused_FN(const T *x, T *y, int64_t n, int L, int64_t batch, int64_t xs, int64_t ys,
                               const WxFilt &filt, const uint8_t *status, int64_t nstatus, hipStream_t st)
{
    const size_t lds = wx_fused_lds_bytes<T>(n);
    auto kern = k_fwd1d_fused<T, F, NT, WRITE_ALL>;
    WX_HIP_CHECK(wx_allow_lds(kern, lds));
    hipLaunchKernelGGL(kern, dim3(wx_fused_grid(lds, batch, NT)), dim3(NT), lds, st, x, y, wx_log2(n), L, batch,
                       xs, ys, filt, status, nstatus);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template <typename T, int F, bool WRITE_ALL>
static int launch_fwd_fused_F(const T *x, T *y, int64_t n, int L, int64_t batch, int64_t xs, int64_t ys,
                              const WxFilt &filt, const uint8_t *status, int64_t nstatus, hipStream_t st)
{
    switch (wx_fused_nt(n)) {
    case 64: return launch_fwd_fused_FN<T, F, WRITE_ALL, 64>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st);
    case 128: return launch_fwd_fused_FN<T, F, WRITE_ALL, 128>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st);
    case 256: return launch_fwd_fused_FN<T, F, WRITE_ALL, 256>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st);
    case 512: return launch_fwd_fused_FN<T, F, WRITE_ALL, 512>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st);
    default: return launch_fwd_fused_FN<T, F, WRITE_ALL, 1024>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st);
    }
}

template <typename T, bool WRITE_ALL>
static int launch_fwd_fused(const T *x, T *y, int64_t n, int L, int64_t batch, int64_t xs, int64_t ys,
                            const WxFilt &filt, const uint8_t *status, int64_t nstatus, hipStream_t st)
{
    switch (filt.F) {
#define WX_CASE(FF) case FF: return launch_fwd_fused_F<T, FF, WRITE_ALL>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st);
        WX_CASE(2) WX_CASE(4) WX_CASE(6) WX_CASE(8) WX_CASE(10) WX_CASE(12) WX_CASE(16) WX_CASE(18) WX_CASE(20)
#undef WX_CASE
    }
    return wx_set_error(WX_EUNSUPPORTED, "no fused instantiation for this filter length");
}

template <typename T, int F, int NT>
static int launch_inv_fused_FN(const T *xw, T *xh, int64_t n, int L, int64_t batch, int64_t is, int64_t os,
                               const WxFilt &filt, const uint8_t *status, int64_t nstatus, const int *colmap,
                               int log2blk, hipStream_t st)
{
    const size_t lds = wx_fused_lds_bytes<T>(n);
    auto kern = k_inv1d_fused<T, F, NT>;
    WX_HIP_CHECK(wx_allow_lds(kern, lds));
    hipLaunchKernelGGL(kern, dim3(wx_fused_grid(lds, batch, NT)), dim3(NT), lds, st, xw, xh, wx_log2(n), L, batch,
                       is, os, filt, status, nstatus, colmap, log2blk);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template <typename T, int F>
static int launch_inv_fused_F(const T *xw, T *xh, int64_t n, int L, int64_t batch, int64_t is, int64_t os,
                              const WxFilt &filt, const uint8_t *status, int64_t nstatus, const int *colmap,
                              int log2blk, hipStream_t st)
{
    switch (wx_fused_nt(n)) {
    case 64: return launch_inv_fused_FN<T, F, 64>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st);
    case 128: return launch_inv_fused_FN<T, F, 128>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st);
    case 256: return launch_inv_fused_FN<T, F, 256>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st);
    case 512: return launch_inv_fused_FN<T, F, 512>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st);
    default: return launch_inv_fused_FN<T, F, 1024>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st);
    }
}

template <typename T>
static int launch_inv_fused(const T *xw, T *xh, int64_t n, int L, int64_t batch, int64_t is, int64_t os,
                            const WxFilt &filt, const uint8_t *status, int64_t nstatus, const int *colmap,
                            int log2blk, hipStream_t st)
{
    switch (filt.F) {
#define WX_CASE(FF) case FF: return launch_inv_fused_F<T, FF>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st);
        WX_CASE(2) WX_CASE(4) WX_CASE(6) WX_CASE(8) WX_CASE(10) WX_CASE(12) WX_CASE(16) WX_CASE(18) WX_CASE(20)
#undef WX_CASE
    }
    return wx_set_error(WX_EUNSUPPORTED, "no fused instantiation for this filter length");
}

// wpd: y is (n, L+1, batch); all device pointers
template <typename T>
int wx_dev_wpd1d(const T *x, T *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st,
                 int force_generic)
{
    if (batch == 0 || n == 0) return WX_OK;
    const int64_t ys = n * (L + 1);
    if (!force_generic && wx_fused1d_ok<T>(n, filt.F))
        return launch_fwd_fused<T, true>(x, y, n, L, batch, n, ys, filt, nullptr, 0, st);
    // column 0 = x (strided 2-D copy), then level by level inside the table
    WX_HIP_CHECK(hipMemcpy2DAsync(y, ys * sizeof(T), x, n * sizeof(T), n * sizeof(T), batch,
                                  hipMemcpyDeviceToDevice, st));
    for (int d = 0; d < L; ++d) {
        const int64_t total = batch * (n / 2);
        hipLaunchKernelGGL(k_fwd1d_level<T>, dim3(wx_grid_for(total, 256)), dim3(256), 0, st, y + d * n,
                           y + (d + 1) * n, ys, ys, (int)n, (int)(n >> d), d, batch, filt,
                           (const uint8_t *)nullptr, (int64_t)0);
    }
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

// wpt: y is (n, batch).  status == nullptr -> full tree of depth L.  scratch: n*batch elements
// (only touched on the generic path when L > 1).
template <typename T>
int wx_dev_wpt1d(const T *x, T *y, int64_t n, int L, int64_t batch, const WxFilt &filt,
                 const uint8_t *status, int64_t nstatus, T *scratch, hipStream_t st, int force_generic)
{
    if (batch == 0 || n == 0) return WX_OK;
    if (L == 0) {
        WX_HIP_CHECK(hipMemcpyAsync(y, x, sizeof(T) * n * batch, hipMemcpyDeviceToDevice, st));
        return WX_OK;
    }
    if (!force_generic && wx_fused1d_ok<T>(n, filt.F))
        return launch_fwd_fused<T, false>(x, y, n, L, batch, n, n, filt, status, nstatus, st);
    // ping-pong so that the last level lands in y
    const T *src = x;
    for (int d = 0; d < L; ++d) {
        T *dst = ((L - 1 - d) & 1) ? scratch : y;
        const int64_t total = batch * (n / 2);
        hipLaunchKernelGGL(k_fwd1d_level<T>, dim3(wx_grid_for(total, 256)), dim3(256), 0, st, src, dst, n, n,
                           (int)n, (int)(n >> d), d, batch, filt, status, nstatus);
        src = dst;
    }
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

// iwpt / iwpd.  in_stride = elements between consecutive signals of xw.  colmap != nullptr: xw is
// the (n, k, batch) packet table (in_stride = n*k) and colmap[blk] names the column each block
// of (1 << log2blk) positions is read from.
template <typename T>
int wx_dev_iwpt1d(const T *xw, T *xh, int64_t n, int L, int64_t batch, const WxFilt &filt,
                  const uint8_t *status, int64_t nstatus, const int *colmap, int log2blk, int64_t in_stride,
                  T *scratch, T *scratch2, hipStream_t st, int force_generic)
{
    if (batch == 0 || n == 0) return WX_OK;
    const int64_t is = in_stride;
    if (L == 0) {
        WX_HIP_CHECK(hipMemcpy2DAsync(xh, n * sizeof(T), xw, is * sizeof(T), n * sizeof(T), batch,
                                      hipMemcpyDeviceToDevice, st));
        return WX_OK;
    }
    if (!force_generic && wx_fused1d_ok<T>(n, filt.F))
        return launch_inv_fused<T>(xw, xh, n, L, batch, is, n, filt, status, nstatus, colmap, log2blk, st);
    const T *src = xw;
    int64_t src_stride = is;
    if (colmap) {
        const int64_t total = batch * n;
        hipLaunchKernelGGL(k_gather_leaves1d<T>, dim3(wx_grid_for(total, 256)), dim3(256), 0, st, xw, scratch2,
                           (int)n, (int)(is / n), batch, colmap, 1 << log2blk);
        src = scratch2;
        src_stride = n;
    }
    for (int d = L - 1; d >= 0; --d) {
        T *dst = (d & 1) ? scratch : xh;
        const int64_t total = batch * (n / 2);
        hipLaunchKernelGGL(k_inv1d_level<T>, dim3(wx_grid_for(total, 256)), dim3(256), 0, st, src, dst, src_stride,
                           n, (int)n, (int)(n >> d), d, batch, filt, status, nstatus);
        src = dst;
        src_stride = n;
    }
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

template <typename T>
int wx_dev_getbasiscoef1d(const T *Xw, T *out, int64_t n, int k, int64_t batch, const int *colmap, int blk,
                          hipStream_t st)
{
    if (batch == 0 || n == 0) return WX_OK;
    const int64_t total = batch * n;
    hipLaunchKernelGGL(k_gather_leaves1d<T>, dim3(wx_grid_for(total, 256)), dim3(256), 0, st, Xw, out, (int)n, k,
                       batch, colmap, blk);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

#define WX_INST(T)                                                                                          \
    template int wx_dev_wpd1d<T>(const T *, T *, int64_t, int, int64_t, const WxFilt &, hipStream_t, int);  \
    template int wx_dev_wpt1d<T>(const T *, T *, int64_t, int, int64_t, const WxFilt &, const uint8_t *,    \
                                 int64_t, T *, hipStream_t, int);                                           \
    template int wx_dev_iwpt1d<T>(const T *, T *, int64_t, int, int64_t, const WxFilt &, const uint8_t *,   \
                                  int64_t, const int *, int, int64_t, T *, T *, hipStream_t, int);              \
    template int wx_dev_getbasiscoef1d<T>(const T *, T *, int64_t, int, int64_t, const int *, int, hipStream_t);
WX_INST(double)
WX_INST(float)
