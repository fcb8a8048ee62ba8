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
#include "wx_host.h"
#include "wx_toptile.h"
#include "wx_lanetree.h"

// the tree-driven lattice kernels for either signal type (Float32: dense leaves only, no threshold riding on the loads)
template <typename T>
static int wx_lattice_tree_T(bool inverse, const T *x, T *y, int64_t n, int L, int64_t batch, int64_t in_stride, int64_t col_stride,
                             const WxFilt &filt, const uint8_t *dstatus, int64_t nstatus, hipStream_t st, const WxThreshArg *thr = nullptr,
                             int64_t out_stride = 0)
{
    if constexpr (sizeof(T) == 8)
        return wx_lattice_tree_f64(inverse, (const double *)x, (double *)y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, st, thr,
                                   out_stride);
    else {
        if (col_stride) return 0;
        return wx_lattice_tree_f32(inverse, (const float *)x, (float *)y, n, L, batch, in_stride, filt, dstatus, nstatus, st, thr, out_stride);
    }
}
template <typename T> static bool wx_lattice_tree_applicable_T(int64_t n, const WxFilt &filt)
{
    if constexpr (sizeof(T) == 8) return wx_lattice_tree_applicable_f64(n, filt);
    else return wx_lattice_tree_applicable_f32(n, filt);
}
#include <stdlib.h>
#include <string.h>
#include <mutex>

// A full tree of depth L <= 12 in the form the tree-driven kernels take: 2^L - 1 status bytes of one.  One constant buffer
// per device, made at the first use.  (Full trees that no dedicated kernel takes -- Float32 signals of 1024 and 2048 samples,
// shallow trees on 2048 -- run 1.5-2 x faster on the tree-driven lattice than on the fused LDS kernels.)
static const uint8_t *wx_full_tree_ones(hipStream_t st)
{
    // through the per-device cache of small constant tables (wx_host.hip): uploaded once per device, ordered on the caller's stream by
    // an event (no synchronisation of that stream under a lock: graph capture stays legal), released by wx_shutdown
    static const struct Ones { uint8_t b[4096]; Ones() { memset(b, 1, sizeof b); } } ones;
    return (const uint8_t *)wx_const_upload(ones.b, sizeof ones.b, st, true);
}
static bool wx_full_as_tree()
{
    static const bool off = wx_getenv("WX_FULL_AS_TREE") && atoi(wx_getenv("WX_FULL_AS_TREE")) == 0;
    return !off;
}

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

// One level of long nodes (np >= 2 WX_LT samples, every node decomposed) through LDS tiles: a workgroup stages WX_LT
// outputs pairs' worth of input (2 WX_LT samples plus the F - 1 wrap-around halo on both sides, coalesced 8-byte runs),
// and every thread computes its pairs out of LDS -- the per-level kernels above read their 2 F taps per pair through
// L1 (8 x the bytes of the output for F = 8) and stop at 3.5 TB/s.  Same arithmetic and tap order as k_fwd1d_level /
// k_inv1d_level.  Used for the top levels of signals of 16384 samples and more (wx_dev_wpt1d / wx_dev_iwpt1d).
constexpr int WX_LT = 2048;                                  // output pairs per tile

// Split form (the pyramid of a long signal, wx_dev_dwt_long / wx_dev_idwt_long): the detail half lives elsewhere -- forward:
// approximations to dst, details to dst2; inverse: approximations from src, details from src2 (each with its own stride
// between signals); src2 / dst2 = nullptr: the two halves are one array as above.
template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void k_level1_tile(const T *__restrict__ src, T *__restrict__ dst, int np, int nper,
                                                     int64_t src_stride, int64_t dst_stride, WxFilt filt,
                                                     const T *__restrict__ src2 = nullptr, int64_t src2_stride = 0,
                                                     T *__restrict__ dst2 = nullptr, int64_t dst2_stride = 0)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    T *buf = reinterpret_cast<T *>(wx_smem);
    const int F = filt.F, h = np >> 1;
    const int tiles = h / WX_LT;
    const int64_t node = blockIdx.x / tiles;
    const int t0 = (int)(blockIdx.x - node * tiles) * WX_LT;            // first output pair of the tile
    const int64_t sig = node / nper;                                    // nper nodes per signal, signals src_stride / dst_stride apart
    const int64_t jn = node - sig * nper;
    const T *v = src + sig * src_stride + jn * (int64_t)np;
    T *o = dst + sig * dst_stride + jn * (int64_t)np;
    const T *vd = src2 ? src2 + sig * src2_stride + jn * (int64_t)np : v + (np >> 1);      // details in (inverse)
    T *od = dst2 ? dst2 + sig * dst2_stride + jn * (int64_t)np : o + (np >> 1);           // details out (forward)
    if (!INVERSE) {
        // inputs 2 t0 - (F - 2) .. 2 t0 + 2 WX_LT + F - 2 (mod np): buf[e] = v[(2 t0 - (F - 2) + e) mod np]
        const int cnt = 2 * WX_LT + 2 * F - 3;
        const int lo = 2 * t0 - (F - 2);
        for (int e = threadIdx.x; e < cnt; e += 256) buf[e] = v[wx_modn(lo + e, np)];
        __syncthreads();
        for (int t = threadIdx.x; t < WX_LT; t += 256) {
            double a = 0.0, d = 0.0;
            int k1 = 2 * t + (F - 2), k2 = 2 * t + 1 + (F - 2);        // positions of v[2 (t0 + t)], v[2 (t0 + t) + 1] in buf
            for (int k = 0; k < F; ++k) {
                a = fma(filt.q[k], (double)buf[k1], a);
                d = fma((k & 1) ? -filt.q[k] : filt.q[k], (double)buf[k2], d);
                ++k1; --k2;
            }
            o[t0 + t] = (T)a;
            od[t0 + t] = (T)d;
        }
    } else {
        // a[k - m], m = 0 .. F/2 - 1 and d[k + m]: bufA[e] = a[(t0 - (F/2 - 1) + e) mod h], bufD[e] = d[(t0 + e) mod h]
        const int HF = F >> 1, cnt = WX_LT + HF - 1;
        T *bufA = buf, *bufD = buf + cnt + 1;
        for (int e = threadIdx.x; e < cnt; e += 256) {
            bufA[e] = v[wx_modn(t0 - (HF - 1) + e, h)];
            bufD[e] = vd[wx_modn(t0 + e, h)];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < WX_LT; t += 256) {
            double v0 = 0.0, v1 = 0.0;
            int k1 = t + HF - 1, k2 = t;
            for (int m = 0; m < HF; ++m) {
                const double av = (double)bufA[k1], dv = (double)bufD[k2];
                v0 = fma(filt.q[2 * m], av, v0);
                v0 = fma(-filt.q[2 * m + 1], dv, v0);
                v1 = fma(filt.q[2 * m + 1], av, v1);
                v1 = fma(filt.q[2 * m], dv, v1);
                --k1; ++k2;
            }
            reinterpret_cast<typename WxVec2<T>::type *>(o)[t0 + t] = typename WxVec2<T>::type{(T)v0, (T)v1};
        }
    }
}

// one full level of every node of np samples (np a multiple of 2 WX_LT): nper nodes per signal, `nsig` signals
template <typename T, bool INVERSE>
static int launch_level1_tile(const T *src, T *dst, int64_t np, int64_t nper, int64_t nsig, int64_t src_stride, int64_t dst_stride,
                              const WxFilt &filt, hipStream_t st, const T *src2 = nullptr, int64_t src2_stride = 0, T *dst2 = nullptr,
                              int64_t dst2_stride = 0)
{
    const int64_t grid = nper * nsig * ((np >> 1) / WX_LT);
    if (grid <= 0 || grid > 0x7fffffff) return wx_set_error(WX_EUNSUPPORTED, "tiled level: grid too large");
    const size_t lds = sizeof(T) * (size_t)(2 * WX_LT + 2 * WX_MAXF + 4);
    hipLaunchKernelGGL((k_level1_tile<T, INVERSE>), dim3((unsigned)grid), dim3(256), lds, st, src, dst, (int)np, (int)nper, src_stride,
                       dst_stride, filt, src2, src2_stride, dst2, dst2_stride);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
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
//
// LDS layout of one level ("4 planes"): natural position x of the level maps to
//   eo = x & 1 (even/odd phase), pair P = x >> 2, w = (x >> 1) & 1, plane parity pp = P & 1,
//   idx = P >> 1;  element = plane(eo, pp)[2*idx + w].
// i.e. plane E<pp> holds the pairs (v[4P], v[4P+2]) and O<pp> the pairs (v[4P+1], v[4P+3]) of the
// pairs with parity pp.  With the even/odd phases separated both QMF branches are unit-stride
// convolutions; with the pair-parity split a lane that owns FOUR adjacent outputs of each branch
// reads its tap window as 16-byte pairs at a lane stride of exactly 16 bytes in every plane
// (conflict-free ds_read_b128), and consecutive planes are offset by 128 B so accesses that
// alternate planes between neighbouring lanes stay conflict-free too.
// Levels whose nodes are shorter than 16 samples are done one node per lane with the filter
// periodised to the node length (WxFold), which also removes the redundant wrapped taps.
// ------------------------------------------------------------------------------------------
#ifdef WX_STAMPS
// diagnostic build only (never shipped): per-phase shader-clock shares of the fused forward kernel
__device__ unsigned long long wx_stamp_buf[8];
#define WX_T(var) const unsigned long long var = clock64()
#else
#define WX_T(var)
#endif

struct WxFold {                 // periodised analysis filters: a[i] = sum_u qaN[u] v[(2i+u) mod N]
    double qa8[8], qd8[8], qa4[4], qd4[4], qa2[2], qd2[2];
};

template <typename T> struct WxVec4;
template <> struct WxVec4<double> { typedef double4 type; };
template <> struct WxVec4<float> { typedef float4 type; };
// store of a lane's 4 consecutive elements (two 16-byte instructions at a lane stride of 32 bytes) to the packet table /
// the output.  Plain stores: each instruction covers half of every 32 bytes and the L2 merges the halves into full
// lines; with the non-temporal hint (WX_WPD_NT=1) the halves reach HBM separately -- measured on config 2:
// 6.4 ms -> 19.6 ms.  (The lattice kernels write whole 128-byte lines per instruction and gain from the hint.)
#ifndef WX_WPD_NT
#define WX_WPD_NT 0
#endif
template <typename V4> __device__ __forceinline__ void wx_gst(V4 *p, const V4 &v)
{
#if WX_WPD_NT
    typedef decltype(v.x) T;
    typedef T EV __attribute__((ext_vector_type(4)));
    EV e = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(e, reinterpret_cast<EV *>(p));
#else
    *p = v;
#endif
}

// Element offset of work item q when every signal is split into 2^sub contiguous nodes of n samples: signal
// q >> sub (stride apart), node q & (2^sub - 1).  sub == 0: plain q * stride.
__device__ __forceinline__ int64_t wx_sub_base(int64_t q, int64_t stride, int sub, int n)
{
    return (q >> sub) * stride + (q & (((int64_t)1 << sub) - 1)) * n;
}

static constexpr int wx_floor_half(int s) { return s >= 0 ? s / 2 : -((-s + 1) / 2); }

// WX_PF = 4-element groups a lane stages per signal (n/4 <= WX_PF * NT): 2 up to n = 8*NT, else 4
template <typename T, int F, int NT, bool WRITE_ALL, int WX_PF>
__global__ __launch_bounds__(NT, 4) void k_fwd1d_fused(const T *__restrict__ x, T *__restrict__ y,
                                                    int log2n, int L, int64_t batch, int64_t x_stride,
                                                    int64_t y_stride, WxFilt filt, WxFold fold,
                                                    const uint8_t *__restrict__ status_g, int64_t nstatus,
                                                    int sub_log2, int64_t lvl_stride)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    typedef typename WxVec2<T>::type V2;
    typedef typename WxVec4<T>::type V4;
    constexpr int HF = F / 2;
    constexpr int BACK = (HF & 1) ? HF - 1 : HF;
    constexpr int NP4 = BACK + 2;                // 16-byte pairs per plane-pair window (4 outputs)
    constexpr int PAD = 64 / (int)sizeof(T);
    const int n = 1 << log2n;
    const int Q = n >> 2;                        // elements per plane
    const int PS = Q + PAD;                      // plane stride (elements)
    T *buf0 = reinterpret_cast<T *>(wx_smem);
    T *buf1 = buf0 + 4 * PS;
    T *fl = buf1 + 4 * PS;                       // periodised filters (28 values), read as LDS broadcasts
    const int tid = threadIdx.x;
    // the tree (one byte per node, the same for every signal) is kept in LDS: the per-item node tests of every
    // level would otherwise each wait for a global load
    uint8_t *sst = reinterpret_cast<uint8_t *>(fl + 32);
    const uint8_t *status = status_g ? sst : nullptr;
    if (status_g) for (int i = tid; i < n && i < nstatus; i += NT) sst[i] = status_g[i];
    if (tid < 28) fl[tid] = (T)reinterpret_cast<const double *>(&fold)[tid];
    const T *qa8 = fl, *qd8 = fl + 8, *qa4 = fl + 16, *qd4 = fl + 20, *qa2 = fl + 24, *qd2 = fl + 26;

    T q[F];
#pragma unroll
    for (int k = 0; k < F; ++k) q[k] = (T)filt.q[k];

    // prefetch the first signal
    V4 pre[WX_PF];
    int64_t b = blockIdx.x;
    if (b < batch) {
#pragma unroll
        for (int k = 0; k < WX_PF; ++k) {
            const int u = tid + k * NT;
            if (u < Q) pre[k] = reinterpret_cast<const V4 *>(x + wx_sub_base(b, x_stride, sub_log2, n))[u];
        }
    }
#ifdef WX_STAMPS
    unsigned long long st_stage = 0, st_comp = 0, st_bar = 0, st_final = 0, st_total = 0;
#endif
    for (; b < batch; b += gridDim.x) {
        WX_T(t_begin);
        T *ys = y + wx_sub_base(b, y_stride, sub_log2, n);
        T *cur = buf0, *nxt = buf1;
        // registers -> planes (and column 0 of the packet table)
#pragma unroll
        for (int k = 0; k < WX_PF; ++k) {
            const int u = tid + k * NT;
            if (u < Q) {
                const V4 v = pre[k];
                V2 ev; ev.x = v.x; ev.y = v.z;
                V2 ov; ov.x = v.y; ov.y = v.w;
                const int pp = u & 1, idx = u >> 1;
                reinterpret_cast<V2 *>(cur + (0 + pp) * PS)[idx] = ev;
                reinterpret_cast<V2 *>(cur + (2 + pp) * PS)[idx] = ov;
                if (WRITE_ALL && sub_log2 == 0) wx_gst(reinterpret_cast<V4 *>(ys) + u, v);
            }
        }
        __syncthreads();
        // next signal's loads fly while this one is transformed
        if (b + gridDim.x < batch) {
#pragma unroll
            for (int k = 0; k < WX_PF; ++k) {
                const int u = tid + k * NT;
                if (u < Q) pre[k] = reinterpret_cast<const V4 *>(x + wx_sub_base(b + gridDim.x, x_stride, sub_log2, n))[u];
            }
        }
        bool direct = false;
        WX_T(t_staged);
#ifdef WX_STAMPS
        st_stage += t_staged - t_begin;
#endif
        for (int d = 0; d < L; ++d) {
            WX_T(t_l0);
            const int lh = log2n - d - 1;        // log2(child length h); node length np = 2h
            const V2 *E0 = reinterpret_cast<const V2 *>(cur), *E1 = reinterpret_cast<const V2 *>(cur + PS);
            const V2 *O0 = reinterpret_cast<const V2 *>(cur + 2 * PS), *O1 = reinterpret_cast<const V2 *>(cur + 3 * PS);
            V2 *En0 = reinterpret_cast<V2 *>(nxt), *En1 = reinterpret_cast<V2 *>(nxt + PS);
            V2 *On0 = reinterpret_cast<V2 *>(nxt + 2 * PS), *On1 = reinterpret_cast<V2 *>(nxt + 3 * PS);
            const bool last = (d == L - 1);
            direct = !WRITE_ALL && last && status == nullptr;     // leaves go straight to HBM
            const bool to_global = WRITE_ALL || direct;
            const bool to_lds = !direct && !(WRITE_ALL && last);
            V4 *yl = reinterpret_cast<V4 *>(WRITE_ALL ? ys + (int64_t)(d + 1) * lvl_stride : ys);
            if (lh >= 3) {
                // ---- four outputs per branch per lane ----
                const int hq4 = 1 << (lh - 2);
                for (int w = tid; w < (n >> 3); w += NT) {
                    const int j = w >> (lh - 2);
                    const int t = w & (hq4 - 1);
                    const int IB = j << (lh - 2);
                    if (status) {
                        const int64_t node = ((int64_t)1 << d) + j;
                        if (!(node <= nstatus && status[node - 1])) {
                            En0[IB + t] = E0[IB + t]; En1[IB + t] = E1[IB + t];
                            On0[IB + t] = O0[IB + t]; On1[IB + t] = O1[IB + t];
                            continue;
                        }
                    }
                    T e[2 * NP4], o[2 * NP4];
#pragma unroll
                    for (int r = 0; r < NP4; ++r) {
                        const int pr = (BACK / 2 + r) & 1;
                        const int kr = wx_floor_half(r - BACK / 2);
                        const int idx = IB + ((t + kr) & (hq4 - 1));
                        const V2 ve = pr ? E1[idx] : E0[idx];
                        const V2 vo = pr ? O1[idx] : O0[idx];
                        e[2 * r] = ve.x; e[2 * r + 1] = ve.y;
                        o[2 * r] = vo.x; o[2 * r + 1] = vo.y;
                    }
                    T a[4], dd[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        T as = 0, ds = 0;
#pragma unroll
                        for (int m = 0; m < HF; ++m) {
                            as = fma(q[2 * m], e[s + m + BACK], as);
                            as = fma(q[2 * m + 1], o[s + m + BACK], as);
                            ds = fma(q[2 * m], o[s - m + BACK], ds);
                            ds = fma(-q[2 * m + 1], e[s - m + BACK], ds);
                        }
                        a[s] = as; dd[s] = ds;
                    }
                    if (to_lds) {
                        const int io = IB + (t >> 1);
                        V2 v;
                        if (t & 1) {
                            v.x = a[0]; v.y = a[2]; En1[io] = v;
                            v.x = a[1]; v.y = a[3]; On1[io] = v;
                            v.x = dd[0]; v.y = dd[2]; En1[io + (hq4 >> 1)] = v;
                            v.x = dd[1]; v.y = dd[3]; On1[io + (hq4 >> 1)] = v;
                        } else {
                            v.x = a[0]; v.y = a[2]; En0[io] = v;
                            v.x = a[1]; v.y = a[3]; On0[io] = v;
                            v.x = dd[0]; v.y = dd[2]; En0[io + (hq4 >> 1)] = v;
                            v.x = dd[1]; v.y = dd[3]; On0[io + (hq4 >> 1)] = v;
                        }
                    }
                    if (to_global) {
                        V4 va; va.x = a[0]; va.y = a[1]; va.z = a[2]; va.w = a[3];
                        V4 vd; vd.x = dd[0]; vd.y = dd[1]; vd.z = dd[2]; vd.w = dd[3];
                        const int g0 = (j << (lh - 1)) + t;           // (j*np + 4t) / 4
                        wx_gst(yl + (g0), va);
                        wx_gst(yl + (g0 + hq4), vd);
                    }
                }
            } else if (lh == 2) {
                // ---- nodes of 8 samples: one node per lane, filter periodised to length 8 ----
                for (int j = tid; j < (n >> 3); j += NT) {
                    const V2 e0 = E0[j], e1 = E1[j], o0 = O0[j], o1 = O1[j];
                    if (status) {
                        const int64_t node = ((int64_t)1 << d) + j;
                        if (!(node <= nstatus && status[node - 1])) { En0[j] = e0; En1[j] = e1; On0[j] = o0; On1[j] = o1; continue; }
                    }
                    const T v[8] = {e0.x, o0.x, e0.y, o0.y, e1.x, o1.x, e1.y, o1.y};
                    T a[4], dd[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        T as = 0, ds = 0;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            as = fma(qa8[u], v[(2 * i + u) & 7], as);
                            ds = fma(qd8[u], v[(2 * i + u) & 7], ds);
                        }
                        a[i] = as; dd[i] = ds;
                    }
                    if (to_lds) {
                        V2 t0; t0.x = a[0]; t0.y = a[2]; En0[j] = t0;
                        t0.x = a[1]; t0.y = a[3]; On0[j] = t0;
                        t0.x = dd[0]; t0.y = dd[2]; En1[j] = t0;
                        t0.x = dd[1]; t0.y = dd[3]; On1[j] = t0;
                    }
                    if (to_global) {
                        V4 va; va.x = a[0]; va.y = a[1]; va.z = a[2]; va.w = a[3];
                        V4 vd; vd.x = dd[0]; vd.y = dd[1]; vd.z = dd[2]; vd.w = dd[3];
                        wx_gst(yl + (2 * j), va); wx_gst(yl + (2 * j + 1), vd);
                    }
                }
            } else if (lh == 1) {
                // ---- nodes of 4 samples ----
                for (int jj = tid; jj < (n >> 3); jj += NT)
                for (int j = 2 * jj; j < 2 * jj + 2; ++j) {
                    const int pp = j & 1, idx = j >> 1;
                    const V2 ev = pp ? E1[idx] : E0[idx];
                    const V2 ov = pp ? O1[idx] : O0[idx];
                    V2 eo_ = ev, oo_ = ov;
                    bool act = true;
                    if (status) {
                        const int64_t node = ((int64_t)1 << d) + j;
                        act = node <= nstatus && status[node - 1];
                    }
                    T a0 = ev.x, a1 = ov.x, d0 = ev.y, d1 = ov.y;    // pass-through values (natural order kept)
                    if (act) {
                        const T v[4] = {ev.x, ov.x, ev.y, ov.y};
                        a0 = a1 = d0 = d1 = 0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            a0 = fma(qa4[u], v[u & 3], a0);
                            a1 = fma(qa4[u], v[(2 + u) & 3], a1);
                            d0 = fma(qd4[u], v[u & 3], d0);
                            d1 = fma(qd4[u], v[(2 + u) & 3], d1);
                        }
                        eo_.x = a0; eo_.y = d0; oo_.x = a1; oo_.y = d1;
                    }
                    if (to_lds) {
                        if (pp) { En1[idx] = eo_; On1[idx] = oo_; } else { En0[idx] = eo_; On0[idx] = oo_; }
                    }
                    if (to_global) {
                        V4 vv;
                        if (act) { vv.x = a0; vv.y = a1; vv.z = d0; vv.w = d1; }
                        else { vv.x = ev.x; vv.y = ov.x; vv.z = ev.y; vv.w = ov.y; }
                        wx_gst(yl + (j), vv);
                    }
                }
            } else {
                // ---- nodes of 2 samples: two nodes per lane ----
                for (int uu = tid; uu < (n >> 3); uu += NT)
                for (int u = 2 * uu; u < 2 * uu + 2; ++u) {
                    const int pp = u & 1, idx = u >> 1;
                    const V2 ev = pp ? E1[idx] : E0[idx];
                    const V2 ov = pp ? O1[idx] : O0[idx];
                    V2 en = ev, on = ov;
                    bool act0 = true, act1 = true;
                    if (status) {
                        const int64_t node = ((int64_t)1 << d) + 2 * u;
                        act0 = node <= nstatus && status[node - 1];
                        act1 = node + 1 <= nstatus && status[node];
                    }
                    if (act0) {
                        en.x = fma(qa2[1], ov.x, qa2[0] * ev.x);
                        on.x = fma(qd2[1], ov.x, qd2[0] * ev.x);
                    }
                    if (act1) {
                        en.y = fma(qa2[1], ov.y, qa2[0] * ev.y);
                        on.y = fma(qd2[1], ov.y, qd2[0] * ev.y);
                    }
                    if (to_lds) {
                        if (pp) { En1[idx] = en; On1[idx] = on; } else { En0[idx] = en; On0[idx] = on; }
                    }
                    if (to_global) { V4 vv; vv.x = en.x; vv.y = on.x; vv.z = en.y; vv.w = on.y; wx_gst(yl + (u), vv); }
                }
            }
            WX_T(t_l1);
            // a level whose nodes fit one wave's 512-sample region (64 items x 8 samples) only
            // exchanges data inside that wave: no workgroup barrier, the wave runs on down the tree
            if (lh <= 8) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else {
                __syncthreads();
            }
            WX_T(t_l2);
#ifdef WX_STAMPS
            st_comp += t_l1 - t_l0; st_bar += t_l2 - t_l1;
#endif
            T *tmp = cur; cur = nxt; nxt = tmp;
        }
        WX_T(t_lev);
        if (!WRITE_ALL && !direct) {
            __syncthreads();                       // the copy-out crosses the waves' regions
            for (int u = tid; u < Q; u += NT) {
                const int pp = u & 1, idx = u >> 1;
                const V2 ev = reinterpret_cast<const V2 *>(cur + (0 + pp) * PS)[idx];
                const V2 ov = reinterpret_cast<const V2 *>(cur + (2 + pp) * PS)[idx];
                V4 v; v.x = ev.x; v.y = ov.x; v.z = ev.y; v.w = ov.y;
                wx_gst(reinterpret_cast<V4 *>(ys) + u, v);
            }
        }
        __syncthreads();                           // all waves are done with this signal's LDS image
#ifdef WX_STAMPS
        { const unsigned long long t_end = clock64(); st_final += t_end - t_lev; st_total += t_end - t_begin; }
#endif
    }
#ifdef WX_STAMPS
    if ((tid & 63) == 0) {
        atomicAdd(&wx_stamp_buf[0], st_stage); atomicAdd(&wx_stamp_buf[1], st_comp); atomicAdd(&wx_stamp_buf[2], st_bar);
        atomicAdd(&wx_stamp_buf[3], st_final); atomicAdd(&wx_stamp_buf[4], st_total); atomicAdd(&wx_stamp_buf[5], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------
// fused forward, in-place variant: ONE level buffer per signal (the 4-plane layout is a function of
// the natural position only, so children overwrite their parent).  Half the LDS of the ping-pong
// kernel -> twice as many signals resident per CU (4 at n = 4096 Float64), i.e. more independent
// load / compute / store phases in flight.  Levels whose nodes fit a wave's 512-sample region run
// with wave-level ordering only; the wider top levels use a read | barrier | write | barrier
// split with the (at most two) items of a lane held in registers.
// ------------------------------------------------------------------------------------------
template <typename T, int F, int NT, bool WRITE_ALL>
__global__ __launch_bounds__(NT, 4) void k_fwd1d_inplace(const T *__restrict__ x, T *__restrict__ y,
                                                      int log2n, int L, int64_t batch, int64_t x_stride,
                                                      int64_t y_stride, WxFilt filt, WxFold fold,
                                                      const uint8_t *__restrict__ status_g, int64_t nstatus)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    typedef typename WxVec2<T>::type V2;
    typedef typename WxVec4<T>::type V4;
    constexpr int HF = F / 2;
    constexpr int BACK = (HF & 1) ? HF - 1 : HF;
    constexpr int NP4 = BACK + 2;
    constexpr int PAD = 64 / (int)sizeof(T);
    constexpr int KI = 2;                        // items per lane and level (n/8 <= KI*NT)
    const int n = 1 << log2n;
    const int Q = n >> 2;
    const int PS = Q + PAD;
    T *buf = reinterpret_cast<T *>(wx_smem);
    T *fl = buf + 4 * PS;
    const int tid = threadIdx.x;
    uint8_t *sst = reinterpret_cast<uint8_t *>(fl + 32);            // the tree in LDS (see k_fwd1d_fused)
    const uint8_t *status = status_g ? sst : nullptr;
    if (status_g) for (int i = tid; i < n && i < nstatus; i += NT) sst[i] = status_g[i];
    if (tid < 28) fl[tid] = (T)reinterpret_cast<const double *>(&fold)[tid];
    const T *qa8 = fl, *qd8 = fl + 8, *qa4 = fl + 16, *qd4 = fl + 20, *qa2 = fl + 24, *qd2 = fl + 26;
    V2 *E0 = reinterpret_cast<V2 *>(buf), *E1 = reinterpret_cast<V2 *>(buf + PS);
    V2 *O0 = reinterpret_cast<V2 *>(buf + 2 * PS), *O1 = reinterpret_cast<V2 *>(buf + 3 * PS);

    T q[F];
#pragma unroll
    for (int k = 0; k < F; ++k) q[k] = (T)filt.q[k];

    // one 8-output item: window reads + FMAs
    auto compute_item = [&](int lh, int w, T (&a)[4], T (&dd)[4]) {
        const int hq4 = 1 << (lh - 2);
        const int j = w >> (lh - 2);
        const int t = w & (hq4 - 1);
        const int IB = j << (lh - 2);
        T e[2 * NP4], o[2 * NP4];
#pragma unroll
        for (int r = 0; r < NP4; ++r) {
            const int pr = (BACK / 2 + r) & 1;
            const int kr = wx_floor_half(r - BACK / 2);
            const int idx = IB + ((t + kr) & (hq4 - 1));
            const V2 ve = pr ? E1[idx] : E0[idx];
            const V2 vo = pr ? O1[idx] : O0[idx];
            e[2 * r] = ve.x; e[2 * r + 1] = ve.y;
            o[2 * r] = vo.x; o[2 * r + 1] = vo.y;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            T as = 0, ds = 0;
#pragma unroll
            for (int m = 0; m < HF; ++m) {
                as = fma(q[2 * m], e[s + m + BACK], as);
                as = fma(q[2 * m + 1], o[s + m + BACK], as);
                ds = fma(q[2 * m], o[s - m + BACK], ds);
                ds = fma(-q[2 * m + 1], e[s - m + BACK], ds);
            }
            a[s] = as; dd[s] = ds;
        }
    };
    auto write_item = [&](int lh, int w, const T (&a)[4], const T (&dd)[4], bool to_lds, bool to_global, V4 *yl) {
        const int hq4 = 1 << (lh - 2);
        const int j = w >> (lh - 2);
        const int t = w & (hq4 - 1);
        const int IB = j << (lh - 2);
        if (to_lds) {
            const int io = IB + (t >> 1);
            V2 *Ea = (t & 1) ? E1 : E0, *Oa = (t & 1) ? O1 : O0;
            V2 v;
            v.x = a[0]; v.y = a[2]; Ea[io] = v;
            v.x = a[1]; v.y = a[3]; Oa[io] = v;
            v.x = dd[0]; v.y = dd[2]; Ea[io + (hq4 >> 1)] = v;
            v.x = dd[1]; v.y = dd[3]; Oa[io + (hq4 >> 1)] = v;
        }
        if (to_global) {
            V4 va; va.x = a[0]; va.y = a[1]; va.z = a[2]; va.w = a[3];
            V4 vd; vd.x = dd[0]; vd.y = dd[1]; vd.z = dd[2]; vd.w = dd[3];
            const int g0 = (j << (lh - 1)) + t;
            wx_gst(yl + (g0), va);
            wx_gst(yl + (g0 + hq4), vd);
        }
    };
    auto node_active = [&](int d, int j) -> bool {
        if (!status) return true;
        const int64_t node = ((int64_t)1 << d) + j;
        return node <= nstatus && status[node - 1];
    };
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) {
        const T *xs = x + b * x_stride;
        T *ys = y + b * y_stride;
        auto stage = [&](int u, const V4 &v) {
            V2 ev; ev.x = v.x; ev.y = v.z;
            V2 ov; ov.x = v.y; ov.y = v.w;
            if (u & 1) { E1[u >> 1] = ev; O1[u >> 1] = ov; } else { E0[u >> 1] = ev; O0[u >> 1] = ov; }
            if (WRITE_ALL) wx_gst(reinterpret_cast<V4 *>(ys) + u, v);
        };
        if (Q == 2 * NT) {
            // both 16-byte groups of a lane in flight together (a loop would wait for the first before the second)
            const V4 v0 = reinterpret_cast<const V4 *>(xs)[tid], v1 = reinterpret_cast<const V4 *>(xs)[tid + NT];
            stage(tid, v0); stage(tid + NT, v1);
        } else if (Q == 4 * NT) {
            const V4 v0 = reinterpret_cast<const V4 *>(xs)[tid], v1 = reinterpret_cast<const V4 *>(xs)[tid + NT];
            const V4 v2 = reinterpret_cast<const V4 *>(xs)[tid + 2 * NT], v3 = reinterpret_cast<const V4 *>(xs)[tid + 3 * NT];
            stage(tid, v0); stage(tid + NT, v1); stage(tid + 2 * NT, v2); stage(tid + 3 * NT, v3);
        } else {
            for (int u = tid; u < Q; u += NT) stage(u, reinterpret_cast<const V4 *>(xs)[u]);
        }
        __syncthreads();
        bool direct = false;
        for (int d = 0; d < L; ++d) {
            const int lh = log2n - d - 1;
            const bool last = (d == L - 1);
            direct = !WRITE_ALL && last && status == nullptr;
            const bool to_global = WRITE_ALL || direct;
            const bool to_lds = !direct && !(WRITE_ALL && last);
            V4 *yl = reinterpret_cast<V4 *>(WRITE_ALL ? ys + (int64_t)(d + 1) * n : ys);
            if (lh >= 3) {
                if (lh <= 8) {
                    // wave-local: read | wave order | write, item by item
                    for (int w = tid; w < (n >> 3); w += NT) {
                        if (!node_active(d, w >> (lh - 2))) continue;         // in place: nothing to copy
                        T a[4], dd[4];
                        compute_item(lh, w, a, dd);
                        wave_sync();
                        write_item(lh, w, a, dd, to_lds, to_global, yl);
                    }
                } else {
                    T a[KI][4], dd[KI][4];
#pragma unroll
                    for (int k = 0; k < KI; ++k) {
                        const int w = tid + k * NT;
                        if (w < (n >> 3) && node_active(d, w >> (lh - 2))) compute_item(lh, w, a[k], dd[k]);
                    }
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < KI; ++k) {
                        const int w = tid + k * NT;
                        if (w < (n >> 3) && node_active(d, w >> (lh - 2))) write_item(lh, w, a[k], dd[k], to_lds, to_global, yl);
                    }
                }
            } else if (lh == 2) {
                for (int j = tid; j < (n >> 3); j += NT) {
                    if (!node_active(d, j)) continue;
                    const V2 e0 = E0[j], e1 = E1[j], o0 = O0[j], o1 = O1[j];
                    const T v[8] = {e0.x, o0.x, e0.y, o0.y, e1.x, o1.x, e1.y, o1.y};
                    T a[4], dd[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        T as = 0, ds = 0;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            as = fma(qa8[u], v[(2 * i + u) & 7], as);
                            ds = fma(qd8[u], v[(2 * i + u) & 7], ds);
                        }
                        a[i] = as; dd[i] = ds;
                    }
                    if (to_lds) {
                        V2 t0; t0.x = a[0]; t0.y = a[2]; E0[j] = t0;
                        t0.x = a[1]; t0.y = a[3]; O0[j] = t0;
                        t0.x = dd[0]; t0.y = dd[2]; E1[j] = t0;
                        t0.x = dd[1]; t0.y = dd[3]; O1[j] = t0;
                    }
                    if (to_global) {
                        V4 va; va.x = a[0]; va.y = a[1]; va.z = a[2]; va.w = a[3];
                        V4 vd; vd.x = dd[0]; vd.y = dd[1]; vd.z = dd[2]; vd.w = dd[3];
                        wx_gst(yl + (2 * j), va); wx_gst(yl + (2 * j + 1), vd);
                    }
                }
            } else if (lh == 1) {
                for (int jj = tid; jj < (n >> 3); jj += NT)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int j = 2 * jj + c;
                    if (!to_global && !node_active(d, j)) continue;           // sparse trees (the dwt pyramid): nothing to read
                    const int pp = j & 1, idx = j >> 1;
                    const V2 ev = pp ? E1[idx] : E0[idx];
                    const V2 ov = pp ? O1[idx] : O0[idx];
                    V4 vv; vv.x = ev.x; vv.y = ov.x; vv.z = ev.y; vv.w = ov.y;
                    if (node_active(d, j)) {
                        const T v[4] = {ev.x, ov.x, ev.y, ov.y};
                        T a0 = 0, a1 = 0, d0 = 0, d1 = 0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            a0 = fma(qa4[u], v[u & 3], a0);
                            a1 = fma(qa4[u], v[(2 + u) & 3], a1);
                            d0 = fma(qd4[u], v[u & 3], d0);
                            d1 = fma(qd4[u], v[(2 + u) & 3], d1);
                        }
                        vv.x = a0; vv.y = a1; vv.z = d0; vv.w = d1;
                        if (to_lds) {
                            V2 en; en.x = a0; en.y = d0;
                            V2 on; on.x = a1; on.y = d1;
                            if (pp) { E1[idx] = en; O1[idx] = on; } else { E0[idx] = en; O0[idx] = on; }
                        }
                    }
                    if (to_global) wx_gst(yl + (j), vv);
                }
            } else {
                for (int uu = tid; uu < (n >> 3); uu += NT)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int u = 2 * uu + c;
                    if (!to_global && !node_active(d, 2 * u) && !node_active(d, 2 * u + 1)) continue;
                    const int pp = u & 1, idx = u >> 1;
                    const V2 ev = pp ? E1[idx] : E0[idx];
                    const V2 ov = pp ? O1[idx] : O0[idx];
                    V2 en = ev, on = ov;
                    if (node_active(d, 2 * u)) {
                        en.x = fma(qa2[1], ov.x, qa2[0] * ev.x);
                        on.x = fma(qd2[1], ov.x, qd2[0] * ev.x);
                    }
                    if (node_active(d, 2 * u + 1)) {
                        en.y = fma(qa2[1], ov.y, qa2[0] * ev.y);
                        on.y = fma(qd2[1], ov.y, qd2[0] * ev.y);
                    }
                    if (to_lds) { if (pp) { E1[idx] = en; O1[idx] = on; } else { E0[idx] = en; O0[idx] = on; } }
                    if (to_global) { V4 vv; vv.x = en.x; vv.y = on.x; vv.z = en.y; vv.w = on.y; wx_gst(yl + (u), vv); }
                }
            }
            if (lh <= 8) wave_sync(); else __syncthreads();
        }
        if (!WRITE_ALL && !direct) {
            __syncthreads();
            for (int u = tid; u < Q; u += NT) {
                const V2 ev = (u & 1) ? E1[u >> 1] : E0[u >> 1];
                const V2 ov = (u & 1) ? O1[u >> 1] : O0[u >> 1];
                V4 v; v.x = ev.x; v.y = ov.x; v.z = ev.y; v.w = ov.y;
                wx_gst(reinterpret_cast<V4 *>(ys) + u, v);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// fused inverse: iwpt / iwpd (leaf gather on load)
//
// LDS layout of one level ("2 planes"): natural position x -> pair P = x >> 1, w = x & 1,
// plane pp = P & 1, idx = P >> 1; element = plane(pp)[2*idx + w]; plane 1 is offset by 128 B.
// A lane owns EIGHT adjacent parent samples (k = 4t..4t+3): both child windows are runs of
// 16-byte pairs read at a lane stride of 16 B per plane (conflict-free).  Nodes of 4 and 2
// samples use the synthesis step folded into a small matrix (WxFoldInv).  Levels whose nodes fit
// a wave's 512-sample region run without workgroup barriers.
// ------------------------------------------------------------------------------------------
struct WxFoldInv {               // v = M x over one node in natural order [a | d]
    double m4[16], m2[4];
};

template <typename T, int F, int NT, int WX_PF>
__global__ __launch_bounds__(NT, 4) void k_inv1d_fused(const T *__restrict__ xw, T *__restrict__ xh,
                                                       int log2n, int L, int64_t batch, int64_t in_stride,
                                                       int64_t out_stride, WxFilt filt, WxFoldInv fold,
                                                       const uint8_t *__restrict__ status_g, int64_t nstatus,
                                                       const int *__restrict__ colmap, int log2blk, WxThreshArg thr)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    typedef typename WxVec2<T>::type V2;
    typedef typename WxVec4<T>::type V4;
    constexpr int HF = F / 2;
    constexpr int BACK = (HF & 1) ? HF - 1 : HF;
    constexpr int NPA = BACK / 2 + 2;            // 16-byte pairs per child window (8 parent samples)
    constexpr int PAD = 128 / (int)sizeof(T);
    const int n = 1 << log2n;
    const int Q = n >> 2;
    const int PS = (n >> 1) + PAD;               // plane stride (elements)
    T *buf0 = reinterpret_cast<T *>(wx_smem);
    T *buf1 = buf0 + 2 * PS;
    T *fl = buf1 + 2 * PS;
    const int tid = threadIdx.x;
    uint8_t *sst = reinterpret_cast<uint8_t *>(fl + 32);            // the tree in LDS (see k_fwd1d_fused)
    const uint8_t *status = status_g ? sst : nullptr;
    if (status_g) for (int i = tid; i < n && i < nstatus; i += NT) sst[i] = status_g[i];
    if (tid < 20) fl[tid] = (T)reinterpret_cast<const double *>(&fold)[tid];
    const T *m4 = fl, *m2 = fl + 16;

    T q[F];
#pragma unroll
    for (int k = 0; k < F; ++k) q[k] = (T)filt.q[k];

    V2 pre[2 * WX_PF];
    auto prefetch = [&](int64_t sig) {
        const T *xs = xw + sig * in_stride;
#pragma unroll
        for (int k = 0; k < WX_PF; ++k) {
            const int u = tid + k * NT;
            if (u < Q) {
                if (colmap) {                    // packet table: the leaf of depth c lives in column c
                    const int c0 = colmap[(4 * u) >> log2blk], c1 = colmap[(4 * u + 2) >> log2blk];
                    pre[2 * k] = reinterpret_cast<const V2 *>(xs + (int64_t)c0 * n)[2 * u];
                    pre[2 * k + 1] = reinterpret_cast<const V2 *>(xs + (int64_t)c1 * n)[2 * u + 1];
                } else {
                    V4 v = (thr.head && u < 16) ? reinterpret_cast<const V4 *>(reinterpret_cast<const T *>(thr.head) + sig * 64)[u]
                                                : reinterpret_cast<const V4 *>(xs)[u];
                    if (thr.t && !(thr.head && u < 16)) {   // denoise: the threshold rides on the load (rows >= thr.lo)
                        const T tt = (T)((double)reinterpret_cast<const T *>(thr.t)[thr.per_signal ? sig : 0] * thr.scale);
                        if (4 * u >= thr.lo) v.x = wx_thresh<T>(v.x, tt, thr.kind);
                        if (4 * u + 1 >= thr.lo) v.y = wx_thresh<T>(v.y, tt, thr.kind);
                        if (4 * u + 2 >= thr.lo) v.z = wx_thresh<T>(v.z, tt, thr.kind);
                        if (4 * u + 3 >= thr.lo) v.w = wx_thresh<T>(v.w, tt, thr.kind);
                    }
                    pre[2 * k].x = v.x; pre[2 * k].y = v.y; pre[2 * k + 1].x = v.z; pre[2 * k + 1].y = v.w;
                }
            }
        }
    };
    int64_t b = blockIdx.x;
    if (b < batch) prefetch(b);
    for (; b < batch; b += gridDim.x) {
        T *os = xh + b * out_stride;
        T *cur = buf0, *nxt = buf1;
#pragma unroll
        for (int k = 0; k < WX_PF; ++k) {
            const int u = tid + k * NT;
            if (u < Q) {
                reinterpret_cast<V2 *>(cur)[u] = pre[2 * k];
                reinterpret_cast<V2 *>(cur + PS)[u] = pre[2 * k + 1];
            }
        }
        __syncthreads();
        if (b + gridDim.x < batch) prefetch(b + gridDim.x);
        for (int d = L - 1; d >= 0; --d) {
            const int lh = log2n - d - 1;        // log2(child length h); parent node length np = 2h
            const V2 *C0 = reinterpret_cast<const V2 *>(cur), *C1 = reinterpret_cast<const V2 *>(cur + PS);
            V2 *N0 = reinterpret_cast<V2 *>(nxt), *N1 = reinterpret_cast<V2 *>(nxt + PS);
            const bool root = (d == 0);
            if (lh >= 2) {
                const int hq4 = 1 << (lh - 2);                       // items per node = h/4
                for (int w = tid; w < (n >> 3); w += NT) {
                    const int j = w >> (lh - 2);
                    const int t = w & (hq4 - 1);
                    const int IA = j << (lh - 1);                    // idx base of child a: j*h/2
                    const int ID = IA + hq4;                         // idx base of child d
                    const int io = IA + 2 * t;                       // idx of the parent's first output pair
                    if (status) {
                        const int64_t node = ((int64_t)1 << d) + j;
                        if (!(node <= nstatus && status[node - 1])) {
                            N0[io] = C0[io]; N1[io] = C1[io]; N0[io + 1] = C0[io + 1]; N1[io + 1] = C1[io + 1];
                            continue;
                        }
                    }
                    T aw[2 * NPA], dw[2 * NPA];
#pragma unroll
                    for (int r = 0; r < NPA; ++r) {
                        const int pa = (BACK / 2 + r) & 1;
                        const int ka = wx_floor_half(r - BACK / 2);
                        const int ia = IA + ((t + ka) & (hq4 - 1));
                        const V2 va = pa ? C1[ia] : C0[ia];
                        const int pd = r & 1;
                        const int id = ID + ((t + (r >> 1)) & (hq4 - 1));
                        const V2 vd = pd ? C1[id] : C0[id];
                        aw[2 * r] = va.x; aw[2 * r + 1] = va.y;
                        dw[2 * r] = vd.x; dw[2 * r + 1] = vd.y;
                    }
                    T v[8];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        T ve = 0, vo = 0;
#pragma unroll
                        for (int m = 0; m < HF; ++m) {
                            ve = fma(q[2 * m], aw[BACK + s - m], ve);
                            ve = fma(-q[2 * m + 1], dw[s + m], ve);
                            vo = fma(q[2 * m + 1], aw[BACK + s - m], vo);
                            vo = fma(q[2 * m], dw[s + m], vo);
                        }
                        v[2 * s] = ve; v[2 * s + 1] = vo;
                    }
                    if (root) {
                        V4 lo; lo.x = v[0]; lo.y = v[1]; lo.z = v[2]; lo.w = v[3];
                        V4 hi; hi.x = v[4]; hi.y = v[5]; hi.z = v[6]; hi.w = v[7];
                        reinterpret_cast<V4 *>(os)[2 * w] = lo;
                        reinterpret_cast<V4 *>(os)[2 * w + 1] = hi;
                    } else {
                        V2 p;
                        p.x = v[0]; p.y = v[1]; N0[io] = p;
                        p.x = v[2]; p.y = v[3]; N1[io] = p;
                        p.x = v[4]; p.y = v[5]; N0[io + 1] = p;
                        p.x = v[6]; p.y = v[7]; N1[io + 1] = p;
                    }
                }
            } else if (lh == 1) {
                // parent nodes of 4 samples [a0 a1 d0 d1]: two nodes (8 positions) per item
                for (int w = tid; w < (n >> 3); w += NT) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int j = 2 * w + c;
                        const int io = 2 * w + c;                    // pairs 2j (plane 0) and 2j+1 (plane 1) share idx j
                        const V2 ap = C0[io], dp = C1[io];
                        bool act = true;
                        if (status) {
                            const int64_t node = ((int64_t)1 << d) + j;
                            act = node <= nstatus && status[node - 1];
                        }
                        V2 lo = ap, hi = dp;
                        if (act) {
                            const T xx[4] = {ap.x, ap.y, dp.x, dp.y};
                            T o[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                T acc = 0;
#pragma unroll
                                for (int u = 0; u < 4; ++u) acc = fma(m4[4 * i + u], xx[u], acc);
                                o[i] = acc;
                            }
                            lo.x = o[0]; lo.y = o[1]; hi.x = o[2]; hi.y = o[3];
                        }
                        N0[io] = lo; N1[io] = hi;
                    }
                }
            } else {
                // parent nodes of 2 samples [a d]: four nodes (8 positions) per item
                for (int w = tid; w < (n >> 3); w += NT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int j = 4 * w + c;                     // node j = pair j
                        const int io = j >> 1;
                        const V2 p = (j & 1) ? C1[io] : C0[io];
                        bool act = true;
                        if (status) {
                            const int64_t node = ((int64_t)1 << d) + j;
                            act = node <= nstatus && status[node - 1];
                        }
                        V2 o = p;
                        if (act) {
                            o.x = fma(m2[1], p.y, m2[0] * p.x);
                            o.y = fma(m2[3], p.y, m2[2] * p.x);
                        }
                        if (j & 1) N1[io] = o; else N0[io] = o;
                    }
                }
            }
            // the next level to run is d-1 (node length 4h): wave-local once 4h <= 512
            if (d > 0) {
                if (lh + 2 <= 9) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                } else {
                    __syncthreads();
                }
            }
            T *tmp = cur; cur = nxt; nxt = tmp;
        }
        __syncthreads();                           // all waves are done with this signal's LDS image
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

template <typename T> static size_t wx_fused_lds_bytes(int64_t n) { return (size_t)2 * n * sizeof(T) + 1024 + 256; }

template <typename T> bool wx_fused1d_ok(int64_t n, int F)
{
    if (!wx_is_pow2(n) || n < 8) return false;
    if (wx_fused_lds_bytes<T>(n) > 160 * 1024) return false;
    switch (F) { case 2: case 4: case 6: case 8: case 10: case 12: case 14: case 16: case 18: case 20: return true; }
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
        const char *e = wx_getenv("WX_FUSED_NT");
        cap = e ? atoi(e) : 512;
        if (cap != 64 && cap != 128 && cap != 256 && cap != 512 && cap != 1024) cap = 512;
    }
    int64_t want = n / 8;
    int nt = 64;
    while (nt < cap && nt < want) nt <<= 1;
    // staging registers: n/4 <= 2*NT below 512 threads (a knob value that is too small is raised), n/4 <= 4*NT above
    while (nt < 512 && n / 4 > 2 * (int64_t)nt) nt <<= 1;
    while (nt < 1024 && n / 4 > 4 * (int64_t)nt) nt <<= 1;
    return nt;
}

// periodise the analysis pair to node lengths 8, 4, 2:  a[i] = sum_k q[k] v[(2i+k) mod N],
// d[i] = sum_k (-1)^k q[k] v[(2i+1-k) mod N]  ->  both as sum_u c[u] v[(2i+u) mod N]
static WxFold wx_make_fold(const WxFilt &f)
{
    WxFold o;
    memset(&o, 0, sizeof o);
    for (int k = 0; k < f.F; ++k) {
        const double qa = f.q[k], qd = (k & 1) ? -f.q[k] : f.q[k];
        o.qa8[k & 7] += qa; o.qa4[k & 3] += qa; o.qa2[k & 1] += qa;
        o.qd8[((1 - k) % 8 + 8) & 7] += qd; o.qd4[((1 - k) % 4 + 4) & 3] += qd; o.qd2[((1 - k) % 2 + 2) & 1] += qd;
    }
    return o;
}

// synthesis step (idwt_step!, dwt_one_level.jl:192-223) of a whole node of length N = 4 or 2 as a
// matrix over the node in natural order x = [a | d]:  v[2k] = sum_m q[2m] a[k-m] - q[2m+1] d[k+m],
// v[2k+1] = sum_m q[2m+1] a[k-m] + q[2m] d[k+m]  (indices mod N/2)
static WxFoldInv wx_make_fold_inv(const WxFilt &f)
{
    WxFoldInv o;
    memset(&o, 0, sizeof o);
    for (int N = 2; N <= 4; N += 2) {
        const int h = N / 2;
        double *M = N == 4 ? o.m4 : o.m2;
        for (int k = 0; k < h; ++k)
            for (int m = 0; m < f.F / 2; ++m) {
                const int ia = ((k - m) % h + h) % h, id = h + (k + m) % h;
                M[(2 * k) * N + ia] += f.q[2 * m];
                M[(2 * k) * N + id] -= f.q[2 * m + 1];
                M[(2 * k + 1) * N + ia] += f.q[2 * m + 1];
                M[(2 * k + 1) * N + id] += f.q[2 * m];
            }
    }
    return o;
}

template <typename K> static hipError_t wx_allow_lds(K kernel, size_t lds)
{
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

// wpd of long signals: the fused kernel starts at depth sub.log2 of the packet table (2^log2 nodes per signal,
// levels sub.lvl elements apart)
struct WxSub { int log2 = 0; int64_t lvl = 0; };

template <typename T, int F, bool WRITE_ALL, int NT, int PF>
static int launch_fwd_fused_FNP(const T *x, T *y, int64_t n, int L, int64_t batch, int64_t xs, int64_t ys,
                                const WxFilt &filt, const uint8_t *status, int64_t nstatus, hipStream_t st, WxSub sub = WxSub())
{
    const size_t lds = wx_fused_lds_bytes<T>(n) + (status ? (size_t)n : 0);     // + the tree bytes
    auto kern = k_fwd1d_fused<T, F, NT, WRITE_ALL, PF>;
    WX_HIP_CHECK(wx_allow_lds(kern, lds));
    const WxFold fold = wx_make_fold(filt);
    hipLaunchKernelGGL(kern, dim3(wx_fused_grid(lds, batch, NT)), dim3(NT), lds, st, x, y, wx_log2(n), L, batch,
                       xs, ys, filt, fold, status, nstatus, sub.log2, sub.lvl ? sub.lvl : n);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template <typename T, int F, bool WRITE_ALL, int NT>
static int launch_fwd_fused_FN(const T *x, T *y, int64_t n, int L, int64_t batch, int64_t xs, int64_t ys,
                               const WxFilt &filt, const uint8_t *status, int64_t nstatus, hipStream_t st, WxSub sub = WxSub())
{
    if (n / 4 <= 2 * NT) return launch_fwd_fused_FNP<T, F, WRITE_ALL, NT, 2>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st, sub);
    if (NT >= 512 && n / 4 <= 4 * NT)
        return launch_fwd_fused_FNP<T, F, WRITE_ALL, (NT >= 512 ? NT : 512), 4>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st, sub);
    return wx_set_error(WX_EUNSUPPORTED, "fused forward: signal too long for the staging registers");
}
static int wx_fwd_mode()
{
    static int mode = -1;
    if (mode < 0) { const char *e = wx_getenv("WX_FWD_MODE"); mode = e ? atoi(e) : 1; }
    return mode;                                   // 0 = ping-pong + prefetch, 1 = in-place
}
template <typename T> static size_t wx_inplace_lds_bytes(int64_t n) { return (size_t)n * sizeof(T) + 512 + 256; }

template <typename T, int F, bool WRITE_ALL, int NT>
static int launch_fwd_inplace_FN(const T *x, T *y, int64_t n, int L, int64_t batch, int64_t xs, int64_t ys,
                                 const WxFilt &filt, const uint8_t *status, int64_t nstatus, hipStream_t st)
{
    const size_t lds = wx_inplace_lds_bytes<T>(n) + (status ? (size_t)n : 0);   // + the tree bytes
    auto kern = k_fwd1d_inplace<T, F, NT, WRITE_ALL>;
    WX_HIP_CHECK(wx_allow_lds(kern, lds));
    const WxFold fold = wx_make_fold(filt);
    hipLaunchKernelGGL(kern, dim3(wx_fused_grid(lds, batch, NT)), dim3(NT), lds, st, x, y, wx_log2(n), L, batch,
                       xs, ys, filt, fold, status, nstatus);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template <typename T, int F, bool WRITE_ALL>
static int launch_fwd_inplace_F(const T *x, T *y, int64_t n, int L, int64_t batch, int64_t xs, int64_t ys,
                                const WxFilt &filt, const uint8_t *status, int64_t nstatus, hipStream_t st)
{
    // two items per lane: NT = n/16 (64..1024)
    int nt = 64;
    while (nt < 1024 && nt * 16 < n) nt <<= 1;
    switch (nt) {
    case 64: return launch_fwd_inplace_FN<T, F, WRITE_ALL, 64>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st);
    case 128: return launch_fwd_inplace_FN<T, F, WRITE_ALL, 128>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st);
    case 256: return launch_fwd_inplace_FN<T, F, WRITE_ALL, 256>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st);
    case 512: return launch_fwd_inplace_FN<T, F, WRITE_ALL, 512>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st);
    default: return launch_fwd_inplace_FN<T, F, WRITE_ALL, 1024>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st);
    }
}

template <typename T, int F, bool WRITE_ALL>
static int launch_fwd_fused_F(const T *x, T *y, int64_t n, int L, int64_t batch, int64_t xs, int64_t ys,
                              const WxFilt &filt, const uint8_t *status, int64_t nstatus, hipStream_t st, WxSub sub = WxSub())
{
    // measured on MI355X (tools/ktime.py): the in-place kernel wins for short filters when only the
    // leaves are written (wpt); long filters run out of registers with two items per lane, and wpd is
    // bound by its store stream either way
    if constexpr (F <= 8 && !WRITE_ALL) {
        if (wx_fwd_mode() == 1 && n <= 16384)
            return launch_fwd_inplace_F<T, F, WRITE_ALL>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st);
    }
    switch (wx_fused_nt(n)) {
    case 64: return launch_fwd_fused_FN<T, F, WRITE_ALL, 64>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st, sub);
    case 128: return launch_fwd_fused_FN<T, F, WRITE_ALL, 128>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st, sub);
    case 256: return launch_fwd_fused_FN<T, F, WRITE_ALL, 256>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st, sub);
    case 512: return launch_fwd_fused_FN<T, F, WRITE_ALL, 512>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st, sub);
    default: return launch_fwd_fused_FN<T, F, WRITE_ALL, 1024>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st, sub);
    }
}

template <typename T, bool WRITE_ALL>
static int launch_fwd_fused(const T *x, T *y, int64_t n, int L, int64_t batch, int64_t xs, int64_t ys,
                            const WxFilt &filt, const uint8_t *status, int64_t nstatus, hipStream_t st, WxSub sub = WxSub())
{
    switch (filt.F) {
#define WX_CASE(FF) case FF: return launch_fwd_fused_F<T, FF, WRITE_ALL>(x, y, n, L, batch, xs, ys, filt, status, nstatus, st, sub);
        WX_CASE(2) WX_CASE(4) WX_CASE(6) WX_CASE(8) WX_CASE(10) WX_CASE(12) WX_CASE(14) WX_CASE(16) WX_CASE(18) WX_CASE(20)
#undef WX_CASE
    }
    return wx_set_error(WX_EUNSUPPORTED, "no fused instantiation for this filter length");
}

template <typename T, int F, int NT, int PF>
static int launch_inv_fused_FNP(const T *xw, T *xh, int64_t n, int L, int64_t batch, int64_t is, int64_t os,
                                const WxFilt &filt, const uint8_t *status, int64_t nstatus, const int *colmap,
                                int log2blk, hipStream_t st, const WxThreshArg &thr)
{
    const size_t lds = wx_fused_lds_bytes<T>(n) + (status ? (size_t)n : 0);     // + the tree bytes
    auto kern = k_inv1d_fused<T, F, NT, PF>;
    WX_HIP_CHECK(wx_allow_lds(kern, lds));
    const WxFoldInv fold = wx_make_fold_inv(filt);
    hipLaunchKernelGGL(kern, dim3(wx_fused_grid(lds, batch, NT)), dim3(NT), lds, st, xw, xh, wx_log2(n), L, batch,
                       is, os, filt, fold, status, nstatus, colmap, log2blk, thr);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template <typename T, int F, int NT>
static int launch_inv_fused_FN(const T *xw, T *xh, int64_t n, int L, int64_t batch, int64_t is, int64_t os,
                               const WxFilt &filt, const uint8_t *status, int64_t nstatus, const int *colmap,
                               int log2blk, hipStream_t st, const WxThreshArg &thr)
{
    if (n / 4 <= 2 * NT) return launch_inv_fused_FNP<T, F, NT, 2>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st, thr);
    if (NT >= 512 && n / 4 <= 4 * NT)
        return launch_inv_fused_FNP<T, F, (NT >= 512 ? NT : 512), 4>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st, thr);
    return wx_set_error(WX_EUNSUPPORTED, "fused inverse: signal too long for the staging registers");
}
template <typename T, int F>
static int launch_inv_fused_F(const T *xw, T *xh, int64_t n, int L, int64_t batch, int64_t is, int64_t os,
                              const WxFilt &filt, const uint8_t *status, int64_t nstatus, const int *colmap,
                              int log2blk, hipStream_t st, const WxThreshArg &thr)
{
    switch (wx_fused_nt(n)) {
    case 64: return launch_inv_fused_FN<T, F, 64>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st, thr);
    case 128: return launch_inv_fused_FN<T, F, 128>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st, thr);
    case 256: return launch_inv_fused_FN<T, F, 256>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st, thr);
    case 512: return launch_inv_fused_FN<T, F, 512>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st, thr);
    default: return launch_inv_fused_FN<T, F, 1024>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st, thr);
    }
}

template <typename T>
static int launch_inv_fused(const T *xw, T *xh, int64_t n, int L, int64_t batch, int64_t is, int64_t os,
                            const WxFilt &filt, const uint8_t *status, int64_t nstatus, const int *colmap,
                            int log2blk, hipStream_t st, const WxThreshArg &thr = WxThreshArg{nullptr, 0, 0, 0, 1.0})
{
    switch (filt.F) {
#define WX_CASE(FF) case FF: return launch_inv_fused_F<T, FF>(xw, xh, n, L, batch, is, os, filt, status, nstatus, colmap, log2blk, st, thr);
        WX_CASE(2) WX_CASE(4) WX_CASE(6) WX_CASE(8) WX_CASE(10) WX_CASE(12) WX_CASE(14) WX_CASE(16) WX_CASE(18) WX_CASE(20)
#undef WX_CASE
    }
    return wx_set_error(WX_EUNSUPPORTED, "no fused instantiation for this filter length");
}

// iwpt along a tree with the threshold of denoise() applied while the coefficients are loaded; the caller has checked
// wx_iwpt1d_thresh_fusable
template <typename T> bool wx_iwpt1d_thresh_fusable(int64_t n, int F, const uint8_t *status)
{
    return status != nullptr && wx_fused1d_ok<T>(n, F);
}
template <typename T>
int wx_dev_iwpt1d_thresh(const T *xw, T *xh, int64_t n, int L, int64_t batch, const WxFilt &filt, const uint8_t *status,
                         int64_t nstatus, const WxThreshArg &thr, hipStream_t st)
{
    if (batch == 0 || n == 0) return WX_OK;
    if constexpr (sizeof(T) == 8) {
        // leaves of a tree (denoiseall(:wpt), Denoising.jl:527): the threshold rides on the loads of the tree-driven lattice inverse
        if (status && !wx_skip_register_kernels()) {
            const int r = wx_lattice_tree_f64(true, (const double *)xw, (double *)xh, n, L, batch, n, 0, filt, status, nstatus, st, &thr);
            if (r) return r < 0 ? r : WX_OK;
        }
    }
    return launch_inv_fused<T>(xw, xh, n, L, batch, n, n, filt, status, nstatus, nullptr, 0, st, thr);
}
template bool wx_iwpt1d_thresh_fusable<double>(int64_t, int, const uint8_t *);
template bool wx_iwpt1d_thresh_fusable<float>(int64_t, int, const uint8_t *);
template int wx_dev_iwpt1d_thresh<double>(const double *, double *, int64_t, int, int64_t, const WxFilt &, const uint8_t *, int64_t,
                                          const WxThreshArg &, hipStream_t);
template int wx_dev_iwpt1d_thresh<float>(const float *, float *, int64_t, int, int64_t, const WxFilt &, const uint8_t *, int64_t,
                                         const WxThreshArg &, hipStream_t);

int wx_lattice_wpd_g_f32(const float *x, float *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st);   // wx_lattice_sgw.hip
int wx_lattice_tree8k_fwd_f64(const double *x, double *y, int64_t batch, const WxFilt &filt, const uint8_t *dstatus0, int depth0,
                              const uint8_t *dstatus1, int depth1, hipStream_t st);                                          // wx_lattice_8ktf.hip
// wpd: y is (n, L+1, batch); all device pointers
template <typename T>
int wx_dev_wpd1d(const T *x, T *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st,
                 int force_generic)
{
    if (batch == 0 || n == 0) return WX_OK;
    const int64_t ys = n * (L + 1);
    if constexpr (sizeof(T) == 8) {
        // 4096-sample signals: rotations in registers, every level leaves through an LDS transposition (wx_lattice.hip)
        if (!force_generic && !wx_skip_register_kernels()) {
            const int r = wx_lattice_wpd_f64((const double *)x, (double *)y, n, L, batch, filt, st);
            if (r) return r < 0 ? r : WX_OK;
        }
    }
    if constexpr (sizeof(T) == 4) {
        // short Float32 signals: the interleaved lattice wpd kernel with Float32 at the two ends (wx_lattice_sgw.hip)
        static const bool g32w_off = wx_getenv("WX_LATTICE_WPD_G32") && atoi(wx_getenv("WX_LATTICE_WPD_G32")) == 0;
        if (!force_generic && !wx_skip_register_kernels() && !g32w_off && n <= 128) {        // 256 samples: the fused LDS kernel is as fast (0.47-0.58 against 0.44-0.50)
            const int r = wx_lattice_wpd_g_f32((const float *)x, (float *)y, n, L, batch, filt, st);
            if (r) return r < 0 ? r : WX_OK;
        }
    }
    if (!force_generic && wx_fused1d_ok<T>(n, filt.F))
        return launch_fwd_fused<T, true>(x, y, n, L, batch, n, ys, filt, nullptr, 0, st);
    // long signals: from the first depth d0 whose nodes fit the LDS of a CU on, the fused kernel finishes each
    // node's subtree on chip, reading column d0 and writing columns d0+1..L of the same table
    int d0 = L;
    if (!force_generic && wx_is_pow2(n)) {
        d0 = 0;
        while (d0 < L && !wx_fused1d_ok<T>(n >> d0, filt.F)) ++d0;
        if (d0 < L && (!wx_fused1d_ok<T>(n >> d0, filt.F) || (batch << d0) > ((int64_t)1 << 40))) d0 = L;
    }
    // the top levels (at most four: signals up to 16 x the longest fused node): ONE tiled pass writes slices 0 .. dtop -- the signal is
    // read once (round 5; until then slice 0 was a copy and every top level read the slice above it: per level a read and a write)
    const int dtop = d0 < L ? d0 : L;
    static const bool topwpd_off = wx_getenv("WX_TOPTILE_WPD") && atoi(wx_getenv("WX_TOPTILE_WPD")) == 0;
    bool top_done = false;
    if (!force_generic && !topwpd_off && dtop >= 1 && dtop <= 4 && wx_is_pow2(n) && n >= 8192 && wx_top_levels_ok(filt.F) && x != y) {
        const int rc = wx_dev_top_levels_wpd<T>(x, y, n, dtop, batch, n, ys, n, filt, st);
        if (rc) return rc;
        top_done = true;
    }
    // column 0 = x (strided 2-D copy), then level by level inside the table
    if (!top_done)
        WX_HIP_CHECK(hipMemcpy2DAsync(y, ys * sizeof(T), x, n * sizeof(T), n * sizeof(T), batch,
                                      hipMemcpyDeviceToDevice, st));
    for (int d = 0; !top_done && d < dtop; ++d) {
        if (!force_generic && (n >> d) >= 4 * WX_LT && wx_is_pow2(n)) {
            const int rc = launch_level1_tile<T, false>(y + d * n, y + (d + 1) * n, n >> d, (int64_t)1 << d, batch, ys, ys, filt, st);
            if (rc) return rc;
            continue;
        }
        const int64_t total = batch * (n / 2);
        hipLaunchKernelGGL(k_fwd1d_level<T>, dim3(wx_grid_for(total, 256)), dim3(256), 0, st, y + d * n,
                           y + (d + 1) * n, ys, ys, (int)n, (int)(n >> d), d, batch, filt,
                           (const uint8_t *)nullptr, (int64_t)0);
    }
    WX_HIP_CHECK(hipGetLastError());
    if (d0 < L) {
        WxSub sub; sub.log2 = d0; sub.lvl = n;
        return launch_fwd_fused<T, true>(y + d0 * n, y + d0 * n, n >> d0, L - d0, batch << d0, ys, ys, filt, nullptr,
                                         0, st, sub);
    }
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
    const bool noreg = wx_skip_register_kernels();      // test hook: fused LDS kernels only
    if constexpr (sizeof(T) == 4) {
        // Float32 full trees of 128 / 256 samples (the columns of 128- / 256-row images arrive here too): the masked tree kernels in Float32
        // arithmetic on pairs of signals, as a tree of ones (wx_lattice_tree_s.h; policy and numbers: api_wpt1d)
        static const bool f32tree_off = wx_getenv("WX_TREES32_FULL") && atoi(wx_getenv("WX_TREES32_FULL")) == 0;
        static const bool f32tree_big = wx_getenv("WX_TREES32_FULL") && atoi(wx_getenv("WX_TREES32_FULL")) == 2;
        if (!force_generic && !noreg && !status && !f32tree_off && (n == 256 || (n == 128 && L >= 2) || (f32tree_big && n >= 1024 && n <= 4096)) && filt.F <= 8 && x != y && batch >= 2 * 4096 / n) {
            const uint8_t *ones = wx_full_tree_ones(st);
            if (ones) {
                const int r = wx_lattice_tree_T<T>(false, x, y, n, L, batch, n, 0, filt, ones, ((int64_t)1 << L) - 1, st);
                if (r) return r < 0 ? r : WX_OK;
            }
        }
        // very short Float32 signals, full tree (the columns of small images arrive here too): one lane per signal (wx_lanetree.h)
        if (!force_generic && !noreg && !status && n <= 128) {
            const int r = wx_lattice_f32(false, (const float *)x, (float *)y, n, L, batch, n, filt, st);
            if (r) return r < 0 ? r : WX_OK;
        }
        if (!force_generic && !noreg && !status && wx_small_tree_wanted<T>(n, filt.F, false, false) && n <= 128 && x != y) {
            WxLaneTree lt;
            for (int i = 0; i < 8; ++i) lt.bits[i] = 0xffffffffu;
            return wx_lane_tree_f32(false, (const float *)x, (float *)y, n, L, batch, lt, filt, st);
        }
    }
    if constexpr (sizeof(T) == 8) {
        // full tree: rotations in registers (wx_lattice.hip; Haar is the lattice with one rotation: 0.78 ms against the
        // 0.94 ms of the Walsh-Hadamard kernel at the target's size)
        if (!force_generic && !noreg && !status) {
            const int r = wx_lattice_wpt_f64((const double *)x, (double *)y, n, L, batch, filt, st);
            if (r) return r < 0 ? r : WX_OK;
        }
        // Haar where the lattice does not apply (depths below 6, ...): Walsh-Hadamard formulation (wx_haar.hip)
        if (!force_generic && !noreg && !status) {
            const int r = wx_haar_wpt_f64((const double *)x, (double *)y, n, L, batch, filt, st);
            if (r) return r < 0 ? r : WX_OK;
        }
    }
    if constexpr (sizeof(T) == 4) {
        // Float32 signals of 4096 samples, full tree: the lattice kernels with Float32 at the two ends (wx_lattice_f32.hip)
        if (!force_generic && !noreg && !status) {
            const int r = wx_lattice_f32(false, (const float *)x, (float *)y, n, L, batch, n, filt, st);
            if (r) return r < 0 ? r : WX_OK;
        }
    }
    // Long signals, full tree: the top levels up to four per pass through LDS (wx_toptile.h; one pass per level until round 4),
    // after which every node is an independent signal -- contiguous, (n2, batch << d0) in Julia layout -- that the lattice
    // kernels (n2 = 4096) or the fused LDS kernel (the longest signal a CU's LDS holds) finish at their own rate.
    if (!force_generic && !status && x != y && wx_is_pow2(n) && n > 4096 && wx_top_levels_ok(filt.F)) {
        int dl = 0;
        while (((int64_t)4096 << dl) < n) ++dl;
        bool lat = !noreg && L - dl >= 6 && n < ((int64_t)1 << 30);
        if constexpr (sizeof(T) == 8) lat = lat && wx_lattice_applicable_f64(filt);
        int64_t n2 = 4096;
        if (!lat) { n2 = n; while (n2 > 2 && !wx_fused1d_ok<T>(n2, filt.F)) n2 >>= 1; }
        int d0 = 0;
        while ((n2 << d0) < n) ++d0;
        // at most four levels: ONE tiled pass whatever the length of the nodes it leaves (until round 5 the levels below the longest node a
        // CU's LDS holds went to the fused kernel: two passes, 0.87-0.93 ms per GiB for a full tree of depth 4 on 8192 ... 65536 samples)
        if (L <= 4 && d0 < L) d0 = L;
        if (d0 >= 1 && wx_fused1d_ok<T>(lat ? 4096 : n2, filt.F) && (scratch || (L <= d0 && L <= 4))) {
            const int Ltop = L < d0 ? L : d0;
            const int npass = (Ltop + 3) / 4;
            // the nodes of depth Ltop land in scratch when a finishing kernel follows (it writes y), in y otherwise
            T *target = L > Ltop ? scratch : y;
            const T *src = x;
            for (int p = 0; p < npass; ++p) {
                T *dst = ((npass - 1 - p) & 1) ? (target == y ? scratch : y) : target;
                const int NLp = Ltop - 4 * p < 4 ? Ltop - 4 * p : 4;
                const int64_t np = n >> (4 * p);
                const int rc = wx_dev_top_levels<T>(false, src, dst, (T *)nullptr, np, NLp, batch << (4 * p), np, np, np, 0xffffffffu, 0u, filt, st);
                if (rc) return rc;
                src = dst;
            }
            if (L == Ltop) return WX_OK;
            if (lat) {
                int r;
                if constexpr (sizeof(T) == 8)
                    r = wx_lattice_wpt_f64((const double *)scratch, (double *)y, 4096, L - d0, batch << d0, filt, st);
                else
                    r = wx_lattice_f32(false, (const float *)scratch, (float *)y, 4096, L - d0, batch << d0, 4096, filt, st);
                if (r < 0) return r;
                if (r == 1) return WX_OK;
            }
            return launch_fwd_fused<T, false>(scratch, y, n2, L - d0, batch << d0, n2, n2, filt, nullptr, 0, st);
        }
    }
    {
        // Longer signals, full tree: after d0 = log2(n / 4096) levels every node is an independent 4096-sample signal --
        // contiguous, (4096, batch << d0) in Julia layout -- which the lattice kernels finish at their own rate.  The top
        // levels are one pass each (the fused LDS kernel with L = 1 while the node fits a CU, the per-level kernel above
        // that); n = 8192, L = 11: 2.32 -> 1.6 ms for 32768 signals.
        int dl = 0;
        while (((int64_t)4096 << dl) < n) ++dl;
        if (!force_generic && !noreg && !status && scratch && dl >= 1 && dl <= 4 && n == ((int64_t)4096 << dl) && L - dl >= 6 &&
            x != y && wx_lattice_applicable_f64(filt)) {
            const T *src = x;
            for (int d = 0; d < dl; ++d) {
                T *dst = ((dl - 1 - d) & 1) ? y : scratch;             // depth dl lands in scratch
                const int64_t nd = n >> d;
                int rc = WX_OK;
                if (wx_fused1d_ok<T>(nd, filt.F))
                    rc = launch_fwd_fused<T, false>(src, dst, nd, 1, batch << d, nd, nd, filt, nullptr, 0, st);
                else
                    rc = launch_level1_tile<T, false>(src, dst, nd, (int64_t)1 << d, batch, n, n, filt, st);
                if (rc) return rc;
                src = dst;
            }
            int r;
            if constexpr (sizeof(T) == 8)
                r = wx_lattice_wpt_f64((const double *)scratch, (double *)y, 4096, L - dl, batch << dl, filt, st);
            else
                r = wx_lattice_f32(false, (const float *)scratch, (float *)y, 4096, L - dl, batch << dl, 4096, filt, st);
            if (r < 0) return r;
            if (r == 1) return WX_OK;
            // not taken after all (alignment): the top levels are in scratch, finish with the fused kernel below
            return launch_fwd_fused<T, false>(scratch, y, 4096, L - dl, batch << dl, 4096, 4096, filt, nullptr, 0, st);
        }
    }
    {
        // a tree (bestbasistree, maketree(:dwt), ...): the lattice computes every node in registers, the tree decides which
        // lines leave after which level (wx_lattice_tree.h; Float32 signals since round 4: wx_lattice_tree32.h)
        if (!force_generic && !noreg && status) {
            const int r = wx_lattice_tree_T<T>(false, x, y, n, L, batch, n, 0, filt, status, nstatus, st);
            if (r) return r < 0 ? r : WX_OK;
        }
        // a full tree nothing above took, as a tree
        if (!force_generic && !noreg && !status && L <= 12 && (n >= 1024 || (n >= 64 && filt.F > 8)) && n <= 4096 && wx_full_as_tree() &&
            wx_lattice_tree_applicable_T<T>(n, filt)) {
            const uint8_t *ones = wx_full_tree_ones(st);
            if (ones) {
                const int r = wx_lattice_tree_T<T>(false, x, y, n, L, batch, n, 0, filt, ones, ((int64_t)1 << L) - 1, st);
                if (r) return r < 0 ? r : WX_OK;
            }
        }
    }
    if (!force_generic && wx_fused1d_ok<T>(n, filt.F))
        return launch_fwd_fused<T, false>(x, y, n, L, batch, n, n, filt, status, nstatus, st);
    // Signals too long for the LDS of one CU (full tree): the first d0 levels run one level per launch; from
    // depth d0 on every node is an independent signal of n >> d0 samples -- contiguous, (n >> d0, batch << d0)
    // in Julia layout -- and the fused kernel finishes them on chip.
    int d0 = 0;
    if (!force_generic && !status && wx_is_pow2(n)) {
        while (d0 < L && !wx_fused1d_ok<T>(n >> d0, filt.F)) ++d0;
        if (d0 >= L || !wx_fused1d_ok<T>(n >> d0, filt.F) || (batch << d0) > ((int64_t)1 << 40)) d0 = 0;
    }
    if (d0 > 0) {
        const T *src = x;
        for (int d = 0; d < d0; ++d) {
            T *dst = ((d0 - 1 - d) & 1) ? y : scratch;                 // depth d0 lands in scratch
            if ((n >> d) >= 4 * WX_LT) {
                const int rc = launch_level1_tile<T, false>(src, dst, n >> d, (int64_t)1 << d, batch, n, n, filt, st);
                if (rc) return rc;
            } else {
                const int64_t total = batch * (n / 2);
                hipLaunchKernelGGL(k_fwd1d_level<T>, dim3(wx_grid_for(total, 256)), dim3(256), 0, st, src, dst, n, n,
                                   (int)n, (int)(n >> d), d, batch, filt, (const uint8_t *)nullptr, (int64_t)0);
            }
            src = dst;
        }
        WX_HIP_CHECK(hipGetLastError());
        const int64_t n2 = n >> d0;
        return launch_fwd_fused<T, false>(scratch, y, n2, L - d0, batch << d0, n2, n2, filt, nullptr, 0, st);
    }
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

// ---- dwt / idwt of long signals (the pyramid; Wavelets.jl dwt / idwt as called by dwtall / idwtall, dwt/dwt_all.jl:33-121) ----
// A pyramid only ever splits its approximation: level d reads the n >> d approximation samples and writes n >> (d + 1)
// approximations and as many details, so the whole transform moves 2 n (1 + 1/2 + 1/4 + ...) < 4 n samples.  The per-level
// kernels for trees (k_fwd1d_level, one launch per level over the WHOLE signal) moved 2 n L: 13 ms for 65536 x 4096 samples'
// worth of 16384-sample signals (4 % of the HBM peak).  Here: the top dl levels are one tiled pass each on the approximation
// (k_level1_tile, details straight to their final place, approximations ping-pong in scratch), and from 4096 samples on the
// tree-driven lattice kernel finishes the pyramid reading / writing with the long signal's stride.
// the length the tiled top levels stop at (0: the long-signal path does not apply) and who finishes: the lattice kernels at 4096
// samples (Float64, filters the lattice factors), otherwise the fused LDS kernel at the longest signal that fits a CU's LDS
// (8192 Float64 / 16384 Float32 samples)
template <typename T> static int64_t wx_dwt_long_plan(int64_t n, const WxFilt &filt, bool *lattice)
{
    static const bool off = wx_getenv("WX_DWT_LONG") && atoi(wx_getenv("WX_DWT_LONG")) == 0;
    *lattice = false;
    if (off || !wx_is_pow2(n) || n > ((int64_t)1 << 24)) return 0;
    // 8192 samples (and, Float32, 16384) fit the fused LDS kernel, but one top pass + the lattice pyramid on the approximation
    // moves 3 n samples at memory speed (Float64 8192 samples: 1.38 -> 1.2 ms per 2 GiB)
    if (n >= 8192 && wx_top_levels_ok(filt.F) && !wx_skip_register_kernels() && wx_lattice_tree_applicable_T<T>(4096, filt)) {
        *lattice = true;
        return 4096;
    }
    if (wx_fused1d_ok<T>(n, filt.F)) return 0;
    if (n >= 16384 && !wx_skip_register_kernels() && wx_lattice_tree_applicable_T<T>(4096, filt)) { *lattice = true; return 4096; }
    int64_t n2 = n;
    while (n2 > 4096 && !wx_fused1d_ok<T>(n2, filt.F)) n2 >>= 1;
    return (n2 >= 4096 && n2 < n && wx_fused1d_ok<T>(n2, filt.F)) ? n2 : 0;
}
template <typename T> bool wx_dwt_long_ok(int64_t n, const WxFilt &filt)
{
    bool lat;
    return wx_dwt_long_plan<T>(n, filt, &lat) != 0;
}
template bool wx_dwt_long_ok<double>(int64_t, const WxFilt &);
template bool wx_dwt_long_ok<float>(int64_t, const WxFilt &);

template <typename T>
int wx_dev_dwt_long(const T *x, T *y, int64_t n, int Lp, int64_t batch, const WxFilt &filt, const uint8_t *status, int64_t nstatus,
                    T *scratch, hipStream_t st)
{
    if (batch == 0 || n == 0) return WX_OK;
    bool lattice;
    const int64_t n2 = wx_dwt_long_plan<T>(n, filt, &lattice);
    if (!n2) return wx_set_error(WX_EUNSUPPORTED, "dwt of a long signal: no plan for this length / filter");
    int dl = 0;
    while ((n2 << dl) < n) ++dl;
    int top = Lp < dl ? Lp : dl;
    if (wx_top_levels_ok(filt.F)) {
        if (Lp <= 4 || (Lp > dl && Lp - dl <= 2 && Lp <= 8 && (n >> 4) >= 4096)) top = Lp;       // see wx_dev_idwt_long
        // up to four levels per pass (wx_toptile.h): the details leave at their final places in y, the approximation that is
        // split further goes to scratch (signal stride n; passes alternate between its two halves)
        const T *src = x;
        int64_t cur_n = n;
        int done = 0, pidx = 0;
        while (done < top) {
            const int NLp = top - done < 4 ? top - done : 4;
            const bool more = done + NLp < top || Lp > top;
            T *deep = scratch + ((pidx & 1) ? n / 2 : 0);
            unsigned split = 0;
            for (int l = 0; l < NLp; ++l) split |= 1u << ((1 << l) - 1);           // heap nodes 1, 2, 4, 8: the approximation branch
            const int rc = wx_dev_top_levels<T>(false, src, y, deep, cur_n, NLp, batch, n, n, n, split, more ? 1u : 0u, filt, st);
            if (rc) return rc;
            src = deep;
            cur_n >>= NLp;
            done += NLp;
            ++pidx;
        }
        if (Lp > top) {
            {
                if (lattice) {
                    const int r = wx_lattice_tree_T<T>(false, src, y, 4096, Lp - dl, batch, n, 0, filt, status, nstatus, st,
                                                      nullptr, n);
                    if (r < 0) return r;
                    if (r == 1) return WX_OK;
                    if (!wx_fused1d_ok<T>(n2, filt.F)) return wx_set_error(WX_EUNSUPPORTED, "dwt of a long signal: more than 2^31 - 1 signals in one call");
                }
            }
            return launch_fwd_fused<T, false>(src, y, n2, Lp - dl, batch, n, n, filt, status, n2 - 1 < nstatus ? n2 - 1 : nstatus, st);
        }
        return WX_OK;
    }
    const int64_t S = n / 2 + n / 4;                           // scratch per signal: approximations of odd / even depth
    T *bufs[2] = {scratch, scratch + n / 2};
    const T *src = x;
    int64_t src_stride = n;
    for (int d = 0; d < top; ++d) {
        const int64_t np = n >> d;
        const int rc = launch_level1_tile<T, false>(src, bufs[d & 1], np, 1, batch, src_stride, S, filt, st, nullptr, 0, y + (np >> 1), n);
        if (rc) return rc;
        src = bufs[d & 1];
        src_stride = S;
    }
    if (Lp > top) {
        {
            if (lattice) {
                const int r = wx_lattice_tree_T<T>(false, src, y, 4096, Lp - dl, batch, S, 0, filt, status, nstatus, st,
                                                  nullptr, n);
                if (r < 0) return r;
                if (r == 1) return WX_OK;
                if (!wx_fused1d_ok<T>(n2, filt.F)) return wx_set_error(WX_EUNSUPPORTED, "dwt of a long signal: more than 2^31 - 1 signals in one call");
            }
        }
        return launch_fwd_fused<T, false>(src, y, n2, Lp - dl, batch, S, n, filt, status, n2 - 1 < nstatus ? n2 - 1 : nstatus, st);
    }
    WX_HIP_CHECK(hipMemcpy2DAsync(y, n * sizeof(T), src, S * sizeof(T), (n >> top) * sizeof(T), batch, hipMemcpyDeviceToDevice, st));
    return WX_OK;
}

template <typename T>
int wx_dev_idwt_long(const T *xw, T *y, int64_t n, int Lp, int64_t batch, const WxFilt &filt, const uint8_t *status, int64_t nstatus,
                     const WxThreshArg &thr, T *scratch, hipStream_t st)
{
    if (batch == 0 || n == 0) return WX_OK;
    bool lattice;
    const int64_t n2 = wx_dwt_long_plan<T>(n, filt, &lattice);
    if (!n2) return wx_set_error(WX_EUNSUPPORTED, "idwt of a long signal: no plan for this length / filter");
    int dl = 0;
    while ((n2 << dl) < n) ++dl;
    int top = Lp < dl ? Lp : dl;
    if (wx_top_levels_ok(filt.F)) {
        // a shallow pyramid, or one that leaves only a level or two for the finishing kernel (which is slow or not applicable
        // there: idwtall of 32768-sample signals with L = 4 took 9.8 ms in round 3), runs entirely in the top passes
        // (the top passes apply neither denoise()'s threshold nor the rebuilt head of the lane-local tail: only without them)
        if (!thr.t && !thr.head && (Lp <= 4 || (Lp > dl && Lp - dl <= 2 && Lp <= 8 && (n >> 4) >= 4096))) top = Lp;   // every pass's input is at least a tile long
        // mirror of wx_dev_dwt_long: the lattice (or fused) inverse rebuilds the approximation of depth `top` in scratch, then up to
        // four synthesis levels per pass, the details coming straight from xw
        const int npass = (top + 3) / 4;
        auto buf = [&](int p) -> T * { return scratch + ((p & 1) ? n / 2 : 0); };
        if (Lp > top) {
            T *cur = buf(npass - 1);
            bool done = false;
            {
                if (lattice) {
                    const int r = wx_lattice_tree_T<T>(true, xw, cur, 4096, Lp - dl, batch, n, 0, filt, status, nstatus, st,
                                                      &thr, n);
                    if (r < 0) return r;
                    done = r == 1;
                    if (!done && !wx_fused1d_ok<T>(n2, filt.F))
                        return wx_set_error(WX_EUNSUPPORTED, "idwt of a long signal: more than 2^31 - 1 signals in one call");
                }
            }
            if (!done) {
                const int rc = launch_inv_fused<T>(xw, cur, n2, Lp - dl, batch, n, n, filt, status, n2 - 1 < nstatus ? n2 - 1 : nstatus, nullptr, 0,
                                                   st, thr);
                if (rc) return rc;
            }
        }
        for (int p = npass - 1; p >= 0; --p) {
            const int NLp = top - 4 * p < 4 ? top - 4 * p : 4;
            const bool more = p < npass - 1 || Lp > top;
            unsigned split = 0;
            for (int l = 0; l < NLp; ++l) split |= 1u << ((1 << l) - 1);
            T *dst = p == 0 ? y : buf(p - 1);
            const int rc = wx_dev_top_levels<T>(true, xw, dst, buf(p), n >> (4 * p), NLp, batch, n, n, n, split, more ? 1u : 0u, filt, st);
            if (rc) return rc;
        }
        return WX_OK;
    }
    const int64_t S = n / 2 + n / 4;
    T *bufs[2] = {scratch, scratch + n / 2};
    // the approximation of depth `top` goes where the forward left it: bufs[(top - 1) & 1]
    T *cur = bufs[(top - 1) & 1];
    if (Lp > top) {
        bool done = false;
        {
            if (lattice) {
                const int r = wx_lattice_tree_T<T>(true, xw, cur, 4096, Lp - dl, batch, n, 0, filt, status, nstatus, st,
                                                  &thr, S);
                if (r < 0) return r;
                done = r == 1;
                if (!done && !wx_fused1d_ok<T>(n2, filt.F))
                    return wx_set_error(WX_EUNSUPPORTED, "idwt of a long signal: more than 2^31 - 1 signals in one call");
            }
        }
        if (!done) {
            const int rc = launch_inv_fused<T>(xw, cur, n2, Lp - dl, batch, n, S, filt, status, n2 - 1 < nstatus ? n2 - 1 : nstatus, nullptr, 0,
                                               st, thr);
            if (rc) return rc;
        }
    } else {
        WX_HIP_CHECK(hipMemcpy2DAsync(cur, S * sizeof(T), xw, n * sizeof(T), (n >> top) * sizeof(T), batch, hipMemcpyDeviceToDevice, st));
    }
    for (int d = top - 1; d >= 0; --d) {
        const int64_t np = n >> d;
        T *dst = d == 0 ? y : bufs[(d - 1) & 1];
        const int rc = launch_level1_tile<T, true>(bufs[d & 1], dst, np, 1, batch, S, d == 0 ? n : S, filt, st, xw + (np >> 1), n);
        if (rc) return rc;
    }
    return WX_OK;
}
template int wx_dev_dwt_long<double>(const double *, double *, int64_t, int, int64_t, const WxFilt &, const uint8_t *, int64_t, double *, hipStream_t);
template int wx_dev_dwt_long<float>(const float *, float *, int64_t, int, int64_t, const WxFilt &, const uint8_t *, int64_t, float *, hipStream_t);
template int wx_dev_idwt_long<double>(const double *, double *, int64_t, int, int64_t, const WxFilt &, const uint8_t *, int64_t,
                                      const WxThreshArg &, double *, hipStream_t);
template int wx_dev_idwt_long<float>(const float *, float *, int64_t, int, int64_t, const WxFilt &, const uint8_t *, int64_t,
                                     const WxThreshArg &, float *, hipStream_t);

// (The launchers of the tree-driven lattice kernels decline pointers that do not start on a 32-byte boundary and batches beyond 2^31 - 1
// signals.  The first cannot reach this file any more -- WxIO hands every entry point aligned device arrays (wx_host.hip, round 6) --
// and the second needs more memory than the device has for signals of this length: what is left below reports exactly that.)
// ---- wpt / iwpt along any tree of long signals ------------------------------------------------------------------------------
// (what bestbasistree returns for signals of 16384 samples and more; until round 3: one launch per level over the whole signal,
// 2-3 % of the HBM peak).  Nodes of depth d < dl = log2(n / 4096) that are split take one tiled pass (runs of consecutive split
// nodes share a launch), children ping-pong between scratch and y at their own positions; a leaf is copied to / from its place; every
// node of 4096 samples that is split further is one launch of the tree-driven lattice kernels with its own subtree.
template <typename T> bool wx_wpt_long_tree_ok(int64_t n, const WxFilt &filt)
{
    bool lattice;
    static const int64_t minn = wx_getenv("WX_LONG_TREE_MINN") ? atoll(wx_getenv("WX_LONG_TREE_MINN")) : 8192;     // 8192: random trees 0.30 / 0.22 -> 0.40 / 0.39 of the HBM peak against the fused LDS kernel
    return n >= minn && wx_dwt_long_plan<T>(n, filt, &lattice) == 4096 && lattice && n <= 65536;
}
template bool wx_wpt_long_tree_ok<double>(int64_t, const WxFilt &);
template bool wx_wpt_long_tree_ok<float>(int64_t, const WxFilt &);

template <typename T>
int wx_dev_wpt_long_tree(const T *x, T *y, int64_t n, int Lp, int64_t batch, const WxFilt &filt, const uint8_t *htree, int64_t ntree,
                         T *scratch, bool inverse, hipStream_t st)
{
    if (batch == 0 || n == 0) return WX_OK;
    {
        int dl = 0;
        while (((int64_t)4096 << dl) < n) ++dl;
        auto split = [&](int d, int64_t j) {                       // node (d, j) exists and is decomposed
            const int64_t idx = ((int64_t)1 << d) + j;
            for (int t = 1; t <= d; ++t) { const int64_t a = idx >> t; if (a - 1 >= ntree || !htree[a - 1]) return false; }
            return d < Lp && idx - 1 < ntree && htree[idx - 1] != 0;
        };
        auto exists = [&](int d, int64_t j) { return d == 0 || split(d - 1, j >> 1); };
        // subtrees of the 4096-sample nodes that are split: one status array each (heap order from the node), uploaded together
        const int64_t NS = 4095;
        std::vector<int64_t> sub_nodes;
        std::vector<int> sub_depth;
        std::vector<uint8_t> hsub;
        if (Lp > dl) {
            for (int64_t j = 0; j < ((int64_t)1 << dl); ++j) {
                if (!split(dl, j)) continue;
                const int64_t h = ((int64_t)1 << dl) + j;               // heap index of the node
                const size_t base = hsub.size();
                hsub.resize(base + NS, 0);
                int depth = 0;
                for (int d2 = 0; d2 < 12 && dl + d2 < Lp; ++d2)
                    for (int64_t j2 = 0; j2 < ((int64_t)1 << d2); ++j2) {
                        const int64_t g = (h << d2) + j2;
                        if (g - 1 < ntree && htree[g - 1] && split(dl + d2, (j << d2) + j2)) { hsub[base + ((int64_t)1 << d2) + j2 - 1] = 1; depth = d2 + 1; }
                    }
                sub_nodes.push_back(j);
                sub_depth.push_back(depth);
            }
        }
        const uint8_t *dsub = nullptr;
        if (!hsub.empty()) {
            dsub = (const uint8_t *)wx_const_upload(hsub.data(), hsub.size(), st, true);
            if (!dsub) return WX_EHIP;
        }
        // 8192-sample Float64 signals, inverse: ONE pass (wx_lattice_8kt.h) -- two wavefronts per signal, each rebuilds its child of the root
        // (the masked tree kernel of its subtree, or a plain load of a leaf), the last synthesis level in direct form
        if constexpr (sizeof(T) == 8) {
            if (inverse && n == 8192 && Lp >= 1 && split(0, 0) && (const void *)x != (const void *)y) {
                const uint8_t *ds[2] = {nullptr, nullptr};
                int dp[2] = {0, 0};
                for (size_t k = 0; k < sub_nodes.size(); ++k) { ds[sub_nodes[k]] = dsub + k * NS; dp[sub_nodes[k]] = sub_depth[k]; }
                const int r = wx_lattice_tree8k_inv_f64((const double *)x, (double *)y, batch, filt, ds[0], dp[0], ds[1], dp[1], st);
                if (r < 0) return r;
                if (r == 1) return WX_OK;
            }
            // forward: the same in one pass (wx_lattice_8ktf.hip), one wavefront per SIMD
            if (!inverse && n == 8192 && Lp >= 1 && split(0, 0) && (const void *)x != (const void *)y) {
                const uint8_t *ds[2] = {nullptr, nullptr};
                int dp[2] = {0, 0};
                for (size_t k = 0; k < sub_nodes.size(); ++k) { ds[sub_nodes[k]] = dsub + k * NS; dp[sub_nodes[k]] = sub_depth[k]; }
                const int r = wx_lattice_tree8k_fwd_f64((const double *)x, (double *)y, batch, filt, ds[0], dp[0], ds[1], dp[1], st);
                if (r < 0) return r;
                if (r == 1) return WX_OK;
            }
        }
        const int top = Lp < dl ? Lp : dl;
        if (wx_top_levels_ok(filt.F) && top >= 1 && top <= 4) {
            // all top levels in one pass through LDS (wx_toptile.h): leaves leave / enter at their own places, the 4096-sample nodes
            // that are split further pass through scratch at theirs
            unsigned smask = 0, deepmask = 0;
            for (int d = 0; d < top; ++d)
                for (int64_t j = 0; j < ((int64_t)1 << d); ++j)
                    if (split(d, j)) smask |= 1u << (((int64_t)1 << d) - 1 + j);
            for (int64_t j : sub_nodes) deepmask |= 1u << j;
            if (!inverse) {
                const int rc = wx_dev_top_levels<T>(false, x, y, scratch, n, top, batch, n, n, n, smask, deepmask, filt, st);
                if (rc) return rc;
            }
            for (size_t k = 0; k < sub_nodes.size(); ++k) {
                const int64_t off = sub_nodes[k] * 4096;
                const int r = inverse ? wx_lattice_tree_T<T>(true, x + off, scratch + off, 4096, sub_depth[k], batch, n, 0,
                                                            filt, dsub + k * NS, NS, st, nullptr, n)
                                      : wx_lattice_tree_T<T>(false, scratch + off, y + off, 4096, sub_depth[k], batch, n, 0,
                                                            filt, dsub + k * NS, NS, st, nullptr, n);
                if (r < 0) return r;
                if (r != 1) return wx_set_error(WX_EUNSUPPORTED, "wpt of a long signal: more than 2^31 - 1 signals in one call");
            }
            if (inverse) return wx_dev_top_levels<T>(true, x, y, scratch, n, top, batch, n, n, n, smask, deepmask, filt, st);
            return WX_OK;
        }
        // depth-d nodes live in P(d): the input / output for d = 0, then scratch, y, scratch, ... (forward) -- every node at its own
        // positions, so a leaf's range is never touched by deeper nodes
        auto P = [&](int d) -> T * { return (d & 1) ? scratch : y; };
        auto copy = [&](const T *src, T *dst, int64_t off, int64_t len) {
            return hipMemcpy2DAsync(dst + off, n * sizeof(T), src + off, n * sizeof(T), len * sizeof(T), batch, hipMemcpyDeviceToDevice, st);
        };
        if (!inverse) {
            for (int d = 0; d <= top; ++d) {
                const int64_t np = n >> d, cnt = (int64_t)1 << d;
                const T *src = d == 0 ? x : P(d);
                for (int64_t j = 0; j < cnt;) {
                    if (!exists(d, j)) { ++j; continue; }
                    if (d < top && split(d, j)) {
                        int64_t j1 = j;
                        while (j1 < cnt && exists(d, j1) && split(d, j1)) ++j1;
                        const int rc = launch_level1_tile<T, false>(src + j * np, P(d + 1) + j * np, np, j1 - j, batch, n, n, filt, st);
                        if (rc) return rc;
                        j = j1;
                        continue;
                    }
                    if (d == dl && Lp > dl && split(d, j)) { ++j; continue; }       // finished by the lattice below
                    if (src != y) WX_HIP_CHECK(copy(src, y, j * np, np));            // a leaf of depth d
                    ++j;
                }
            }
            for (size_t k = 0; k < sub_nodes.size(); ++k) {
                const int64_t off = sub_nodes[k] * 4096;
                const int r = wx_lattice_tree_T<T>(false, (dl == 0 ? x : P(dl)) + off, y + off, 4096, sub_depth[k], batch, n,
                                                  0, filt, dsub + k * NS, NS, st, nullptr, n);
                if (r < 0) return r;
                if (r != 1) return wx_set_error(WX_EUNSUPPORTED, "wpt of a long signal: more than 2^31 - 1 signals in one call");
            }
            return WX_OK;
        }
        // inverse: depth-d nodes are rebuilt into Q(d) = y for d = 0, then scratch, y, ...; leaves come from xw
        auto Q = [&](int d) -> T * { return (d & 1) ? scratch : y; };
        for (size_t k = 0; k < sub_nodes.size(); ++k) {
            const int64_t off = sub_nodes[k] * 4096;
            const int r = wx_lattice_tree_T<T>(true, x + off, Q(dl) + off, 4096, sub_depth[k], batch, n, 0, filt,
                                              dsub + k * NS, NS, st, nullptr, n);
            if (r < 0) return r;
            if (r != 1) return wx_set_error(WX_EUNSUPPORTED, "iwpt of a long signal: more than 2^31 - 1 signals in one call");
        }
        for (int d = top; d >= 0; --d) {
            const int64_t np = n >> d, cnt = (int64_t)1 << d;
            for (int64_t j = 0; j < cnt;) {
                if (!exists(d, j)) { ++j; continue; }
                if (d < top && split(d, j)) {
                    int64_t j1 = j;
                    while (j1 < cnt && exists(d, j1) && split(d, j1)) ++j1;
                    const int rc = launch_level1_tile<T, true>(Q(d + 1) + j * np, Q(d) + j * np, np, j1 - j, batch, n, n, filt, st);
                    if (rc) return rc;
                    j = j1;
                    continue;
                }
                if (d == dl && Lp > dl && split(d, j)) { ++j; continue; }           // rebuilt by the lattice above
                WX_HIP_CHECK(copy(x, Q(d), j * np, np));                             // a leaf of depth d: from the input
                ++j;
            }
        }
        return WX_OK;
    }
}
template int wx_dev_wpt_long_tree<double>(const double *, double *, int64_t, int, int64_t, const WxFilt &, const uint8_t *, int64_t, double *, bool,
                                          hipStream_t);
template int wx_dev_wpt_long_tree<float>(const float *, float *, int64_t, int, int64_t, const WxFilt &, const uint8_t *, int64_t, float *, bool,
                                         hipStream_t);

// iwpt / iwpd.  in_stride = elements between consecutive signals of xw.  colmap != nullptr: xw is
// the (n, k, batch) packet table (in_stride = n*k) and colmap[blk] names the column each block
// of (1 << log2blk) positions is read from.
template <typename T>
int wx_dev_iwpt1d(const T *xw, T *xh, int64_t n, int L, int64_t batch, const WxFilt &filt,
                  const uint8_t *status, int64_t nstatus, const int *colmap, int log2blk, int64_t in_stride,
                  T *scratch, T *scratch2, hipStream_t st, int force_generic)
{
    if (batch == 0 || n == 0) return WX_OK;
    int64_t is = in_stride;
    if (L == 0) {
        WX_HIP_CHECK(hipMemcpy2DAsync(xh, n * sizeof(T), xw, is * sizeof(T), n * sizeof(T), batch,
                                      hipMemcpyDeviceToDevice, st));
        return WX_OK;
    }
    const bool noreg = wx_skip_register_kernels();
    // iwpd of a deep full tree on a long signal: the leaves are one slice of the packet table (signals n * k apart), and the kernels of the
    // long-signal path below want them dense -- one strided copy, then the tiled top levels + the lattice (three passes) instead of the fused
    // kernel + one launch per top level.  (At most four levels are one tiled pass that takes the stride itself.)
    if (!force_generic && !status && !colmap && is != n && scratch2 && L > 4 && xw != xh && wx_is_pow2(n) && n > 4096 && n < ((int64_t)1 << 30) &&
        wx_top_levels_ok(filt.F) && !wx_fused1d_ok<T>(n, filt.F)) {
        WX_HIP_CHECK(hipMemcpy2DAsync(scratch2, n * sizeof(T), xw, is * sizeof(T), n * sizeof(T), batch, hipMemcpyDeviceToDevice, st));
        xw = scratch2;
        is = n;
    }
    if constexpr (sizeof(T) == 4) {
        if (!force_generic && !noreg && !status && !colmap && n <= 128) {
            const int r = wx_lattice_f32(true, (const float *)xw, (float *)xh, n, L, batch, is, filt, st);
            if (r) return r < 0 ? r : WX_OK;
        }
        if (!force_generic && !noreg && !status && !colmap && is == n && wx_small_tree_wanted<T>(n, filt.F, false, false) && n <= 128 && xw != xh) {
            WxLaneTree lt;
            for (int i = 0; i < 8; ++i) lt.bits[i] = 0xffffffffu;
            return wx_lane_tree_f32(true, (const float *)xw, (float *)xh, n, L, batch, lt, filt, st);
        }
    }
    if constexpr (sizeof(T) == 8) {
        if (!force_generic && !noreg && !status && !colmap) {
            const int r = wx_lattice_iwpt_f64((const double *)xw, (double *)xh, n, L, batch, is, filt, st);
            if (r) return r < 0 ? r : WX_OK;
        }
        // Haar where the lattice does not apply, dense leaves: inverse Walsh-Hadamard formulation (wx_haar.hip)
        if (!force_generic && !noreg && !status && !colmap && is == n) {
            const int r = wx_haar_iwpt_f64((const double *)xw, (double *)xh, n, L, batch, filt, st);
            if (r) return r < 0 ? r : WX_OK;
        }
    }
    if constexpr (sizeof(T) == 4) {
        static const bool f32tree_off = wx_getenv("WX_TREES32_FULL") && atoi(wx_getenv("WX_TREES32_FULL")) == 0;
        static const bool f32tree_big = wx_getenv("WX_TREES32_FULL") && atoi(wx_getenv("WX_TREES32_FULL")) == 2;
        if (!force_generic && !noreg && !status && !colmap && !f32tree_off && (n == 256 || (n == 128 && L >= 2) || (f32tree_big && n >= 1024 && n <= 4096)) && filt.F <= 8 && xw != xh &&
            batch >= 2 * 4096 / n) {
            const uint8_t *ones = wx_full_tree_ones(st);
            if (ones) {
                const int r = wx_lattice_tree_T<T>(true, xw, xh, n, L, batch, is, 0, filt, ones, ((int64_t)1 << L) - 1, st);
                if (r) return r < 0 ? r : WX_OK;
            }
        }
        if (!force_generic && !noreg && !status && !colmap) {
            const int r = wx_lattice_f32(true, (const float *)xw, (float *)xh, n, L, batch, is, filt, st);
            if (r) return r < 0 ? r : WX_OK;
        }
    }
    // mirror of the long-signal path of wx_dev_wpt1d: the finishing kernels rebuild the nodes of depth d0 (the lattice inverse
    // on 4096-sample nodes, or the fused LDS kernel), then the top levels up to four per pass (wx_toptile.h)
    // (leaves that are not dense -- the deepest slice of a packet table, iwpd of a full tree -- only when the whole transform is one tiled
    // pass: the pass takes the stride of its input, the finishing kernels want (node, signal) contiguous)
    if (!force_generic && !status && !colmap && (is == n || (L <= 4 && is > n && !(is & 1))) && xw != xh && wx_is_pow2(n) && n > 4096 &&
        wx_top_levels_ok(filt.F)) {
        int dl = 0;
        while (((int64_t)4096 << dl) < n) ++dl;
        bool lat = !noreg && L - dl >= 6 && n < ((int64_t)1 << 30);
        if constexpr (sizeof(T) == 8) lat = lat && wx_lattice_applicable_f64(filt);
        int64_t n2 = 4096;
        if (!lat) { n2 = n; while (n2 > 2 && !wx_fused1d_ok<T>(n2, filt.F)) n2 >>= 1; }
        int d0 = 0;
        while ((n2 << d0) < n) ++d0;
        // at most four levels: ONE tiled pass whatever the length of the nodes it leaves (until round 5 the levels below the longest node a
        // CU's LDS holds went to the fused kernel: two passes, 0.87-0.93 ms per GiB for a full tree of depth 4 on 8192 ... 65536 samples)
        // (the inverse only where the fused kernel does not take the whole signal in one pass: its single pass is faster -- 8192 Float64
        // samples 0.52 against 0.61 ms per GiB, 16384 Float32 samples 0.55 against 0.75)
        if (L <= 4 && d0 >= 1 && d0 < L) d0 = L;
        if (is != n && d0 < L) d0 = -1;                      // strided leaves and more than one kernel: not here
        if (d0 >= 1 && wx_fused1d_ok<T>(lat ? 4096 : n2, filt.F) && (scratch || (L <= d0 && L <= 4))) {
            const int Ltop = L < d0 ? L : d0;
            const int npass = (Ltop + 3) / 4;
            // pass p (npass-1 .. 0) writes buffer B(p): xh for p = 0, then scratch, xh, ...; the finishing kernels write B(npass)
            auto B = [&](int p) -> T * { return (p & 1) ? scratch : xh; };
            const T *src = xw;
            if (L > Ltop) {
                T *first = B(npass);
                bool done = false;
                if (lat) {
                    int r;
                    if constexpr (sizeof(T) == 8)
                        r = wx_lattice_iwpt_f64((const double *)xw, (double *)first, 4096, L - d0, batch << d0, 4096, filt, st);
                    else
                        r = wx_lattice_f32(true, (const float *)xw, (float *)first, 4096, L - d0, batch << d0, 4096, filt, st);
                    if (r < 0) return r;
                    done = r == 1;
                }
                if (!done) {
                    const int rc = launch_inv_fused<T>(xw, first, n2, L - d0, batch << d0, n2, n2, filt, nullptr, 0, nullptr, 0, st);
                    if (rc) return rc;
                }
                src = first;
            }
            for (int p = npass - 1; p >= 0; --p) {
                const int NLp = Ltop - 4 * p < 4 ? Ltop - 4 * p : 4;
                const int64_t np = n >> (4 * p);
                const int rc = wx_dev_top_levels<T>(true, src, B(p), (T *)nullptr, np, NLp, batch << (4 * p), (src == xw && is != n) ? is : np, np, np,
                                                    0xffffffffu, 0u, filt, st);
                if (rc) return rc;
                src = B(p);
            }
            return WX_OK;
        }
    }
    {
        // mirror of the long-signal path of wx_dev_wpt1d: the lattice inverse on the 4096-sample nodes of depth dl, then dl
        // synthesis levels of one pass each
        int dl = 0;
        while (((int64_t)4096 << dl) < n) ++dl;
        if (!force_generic && !noreg && !status && !colmap && is == n && scratch && dl >= 1 && dl <= 4 &&
            n == ((int64_t)4096 << dl) && L - dl >= 6 && xw != xh && wx_lattice_applicable_f64(filt)) {
            T *first = (dl & 1) ? scratch : xh;                        // as if depth dl were one more level of the ping-pong
            int r;
            if constexpr (sizeof(T) == 8)
                r = wx_lattice_iwpt_f64((const double *)xw, (double *)first, 4096, L - dl, batch << dl, 4096, filt, st);
            else
                r = wx_lattice_f32(true, (const float *)xw, (float *)first, 4096, L - dl, batch << dl, 4096, filt, st);
            if (r < 0) return r;
            if (r == 1) {
                const T *src2 = first;
                for (int d = dl - 1; d >= 0; --d) {
                    T *dst = (d & 1) ? scratch : xh;
                    const int64_t nd = n >> d;
                    int rc = WX_OK;
                    if (wx_fused1d_ok<T>(nd, filt.F))
                        rc = launch_inv_fused<T>(src2, dst, nd, 1, batch << d, nd, nd, filt, nullptr, 0, nullptr, 0, st);
                    else
                        rc = launch_level1_tile<T, true>(src2, dst, nd, (int64_t)1 << d, batch, n, n, filt, st);
                    if (rc) return rc;
                    src2 = dst;
                }
                return WX_OK;
            }
        }
    }
    {
        // leaves of a tree, dense (iwpt) or in the columns of a packet table (iwpd by tree: colmap is set, the leaves of depth
        // l sit in column l): the lattice inverse takes them in level by level (wx_lattice_tree.h)
        if (!force_generic && !noreg && status) {
            const int r = wx_lattice_tree_T<T>(true, xw, xh, n, L, batch, is, colmap ? n : 0, filt, status,
                                              nstatus, st);
            if (r) return r < 0 ? r : WX_OK;
        }
        if (!force_generic && !noreg && !status && !colmap && L <= 12 && (n >= 1024 || (n >= 64 && filt.F > 8)) && n <= 4096 && wx_full_as_tree() &&
            wx_lattice_tree_applicable_T<T>(n, filt)) {
            const uint8_t *ones = wx_full_tree_ones(st);
            if (ones) {
                const int r = wx_lattice_tree_T<T>(true, xw, xh, n, L, batch, is, 0, filt, ones, ((int64_t)1 << L) - 1, st);
                if (r) return r < 0 ? r : WX_OK;
            }
        }
    }
    if (!force_generic && wx_fused1d_ok<T>(n, filt.F))
        return launch_inv_fused<T>(xw, xh, n, L, batch, is, n, filt, status, nstatus, colmap, log2blk, st);
    // long signals, full tree, dense leaves: depths L-1 .. d0 on chip as (n >> d0, batch << d0) sub-signals,
    // the remaining d0 levels one per launch (mirror of wx_dev_wpt1d)
    int d0 = 0;
    if (!force_generic && !status && !colmap && (is == n || scratch2) && wx_is_pow2(n)) {
        while (d0 < L && !wx_fused1d_ok<T>(n >> d0, filt.F)) ++d0;
        if (d0 >= L || !wx_fused1d_ok<T>(n >> d0, filt.F) || (batch << d0) > ((int64_t)1 << 40)) d0 = 0;
    }
    if (d0 > 0 && is != n) {
        // leaves of a packet table (one column per signal, signals n*k apart): densify once
        WX_HIP_CHECK(hipMemcpy2DAsync(scratch2, n * sizeof(T), xw, is * sizeof(T), n * sizeof(T), batch,
                                      hipMemcpyDeviceToDevice, st));
        xw = scratch2;
    }
    if (d0 > 0) {
        T *fbuf = (d0 & 1) ? scratch : xh;                              // as if depth d0 were one more generic level
        const int64_t n2 = n >> d0;
        int rc = launch_inv_fused<T>(xw, fbuf, n2, L - d0, batch << d0, n2, n2, filt, nullptr, 0, nullptr, 0, st);
        if (rc) return rc;
        const T *src2 = fbuf;
        for (int d = d0 - 1; d >= 0; --d) {
            T *dst = (d & 1) ? scratch : xh;
            if ((n >> d) >= 4 * WX_LT) {
                const int rc = launch_level1_tile<T, true>(src2, dst, n >> d, (int64_t)1 << d, batch, n, n, filt, st);
                if (rc) return rc;
            } else {
                const int64_t total = batch * (n / 2);
                hipLaunchKernelGGL(k_inv1d_level<T>, dim3(wx_grid_for(total, 256)), dim3(256), 0, st, src2, dst, n, n, (int)n,
                                   (int)(n >> d), d, batch, filt, (const uint8_t *)nullptr, (int64_t)0);
            }
            src2 = dst;
        }
        WX_HIP_CHECK(hipGetLastError());
        return WX_OK;
    }
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

#ifdef WX_STAMPS
extern "C" int wx_debug_read_stamps(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(wx_stamp_buf), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(wx_stamp_buf), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
