// wx_shrink.hip -- threshold selection of the reference's SureShrink and RelErrorShrink on the device, one workgroup per
// signal (the part round 1 left out).
//
// Reference (paths relative to /root/reference/src/mod):
//   surethreshold(coef, redundant, tree)              Denoising.jl:146-166
//       a = sort(abs.(y)).^2; b = cumsum(a); risk_i = (n - 2 i + b_i + (n - i) a_i) / n; t = sqrt(a[argmin(risk)])
//   relerrorthreshold(coef, redundant, tree, elbows)  Denoising.jl:285-327
//       x = sort(abs.(c), rev = true); r = orth2relerror(c) (Denoising.jl:344-349: sqrt(|sum - cumsum|) / sqrt(sum) of the
//       squares sorted downwards); the curve (x reversed / xmax, r reversed / ymax) with the point (0, r_n) in front and
//       r_1 repeated at the end; `elbows` nested applications of findelbow (Denoising.jl:367-381: the point farthest
//       from the chord between the first and the last point of the current prefix); t = x[elbow] * xmax
// Both are what `denoiseall(...; estnoise = relerrorthreshold)` evaluates for every signal (test/denoising.jl:59-83), so
// they are batched here exactly like the MAD noise estimate of wx_denoise.hip: the selected coefficients of one signal are
// sorted by a bitonic network in LDS (up to 8192 Float64 / 16384 Float32 coefficients; above that in a global scratch window that stays in L2;
// above the LDS window with launches over the whole chip),
// the cumulative sums are a workgroup scan, and argmin / argmax keep the first index on ties like Julia's findmin /
// findmax.  The results are picks from the sorted magnitudes, so they equal the reference's unless two candidates tie to
// within the rounding of the sums (the reference adds sequentially / pairwise, the scan adds by chunks).
#include "../../include/waveletsext_hip.h"     // the definitions below must match the public prototypes
#include "wx_common.h"
#include "wx_kernels.h"
#include "wx_host.h"
#include <vector>

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

extern "C" int wx_device_count(void);

namespace {

constexpr int SH_NT = 256;

template <typename T> __device__ __forceinline__ T sh_inf();
template <> __device__ __forceinline__ double sh_inf<double>() { return __longlong_as_double(0x7ff0000000000000LL); }
template <> __device__ __forceinline__ float sh_inf<float>() { return __int_as_float(0x7f800000); }

// |X[row, cols[c], signal]| of the selected columns into v[0, cnt), +Inf padding up to npad
template <typename T>
__device__ void sh_stage(T *v, const T *x, int n, const int *cols, int ncols, int cnt, int npad)
{
    for (int e = threadIdx.x; e < npad; e += SH_NT) {
        T val = sh_inf<T>();
        if (e < cnt) {
            const int c = e / n, row = e - c * n;
            val = (T)fabs((double)x[(int64_t)(cols ? cols[c] : c) * n + row]);
        }
        v[e] = val;
    }
    __syncthreads();
}

template <typename T> __device__ void sh_bitonic(T *v, int npad)       // ascending
{
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < (npad >> 1); i += SH_NT) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
                const bool up = (lo & k) == 0;
                const T a = v[lo], b = v[hi];
                if ((a > b) == up) { v[lo] = b; v[hi] = a; }
            }
            __syncthreads();
        }
}

// w[pos(e)] = f(inclusive running sum of g(e)), e = 0 .. cnt-1: every thread owns a contiguous chunk, the chunk sums are
// scanned through LDS; returns the total to every thread
template <typename T, typename G, typename W>
__device__ T sh_scan(int cnt, T *part, G g, W put)
{
    const int C = (cnt + SH_NT - 1) / SH_NT, e0 = threadIdx.x * C, e1 = min(cnt, e0 + C);
    T s = 0;
    for (int e = e0; e < e1; ++e) s += g(e);
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
        for (int t = 0; t < SH_NT; ++t) { const T p = part[t]; part[t] = run; run += p; }
        part[SH_NT] = run;
    }
    __syncthreads();
    T run = part[threadIdx.x];
    for (int e = e0; e < e1; ++e) { run += g(e); put(e, run); }
    const T total = part[SH_NT];
    __syncthreads();
    return total;
}

// first index of the extreme value of f(j), j = 0 .. m-1 (MAXIMUM: largest, else smallest), as findmax / argmin
template <typename T, bool MAXIMUM, typename F>
__device__ int sh_argext(int m, T *rv, int *ri, F f)
{
    T best = 0;
    int bi = -1;
    for (int j = threadIdx.x; j < m; j += SH_NT) {
        const T val = f(j);
        if (bi < 0 || (MAXIMUM ? val > best : val < best)) { best = val; bi = j; }
    }
    rv[threadIdx.x] = best; ri[threadIdx.x] = bi;
    __syncthreads();
    for (int s = SH_NT >> 1; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const int oi = ri[threadIdx.x + s];
            const T ov = rv[threadIdx.x + s];
            const int mi = ri[threadIdx.x];
            const T mv = rv[threadIdx.x];
            if (oi >= 0 && (mi < 0 || (MAXIMUM ? ov > mv : ov < mv) || (ov == mv && oi < mi))) { rv[threadIdx.x] = ov; ri[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    const int res = ri[0];
    __syncthreads();
    return res;
}

template <typename T> struct ShShared { T part[SH_NT + 1]; T rv[SH_NT]; int ri[SH_NT]; };

template <typename T>
__device__ __forceinline__ T *sh_window(T *gscratch, int npad)
{
    extern __shared__ __attribute__((aligned(16))) char sh_smem[];
    return gscratch ? gscratch + (int64_t)blockIdx.x * (2 * (int64_t)npad + 8) : reinterpret_cast<T *>(sh_smem);
}

// ---- large selections: the sort as launches over the whole chip ------------------------------------------------------------
// Above SH_WG_MAX values one workgroup per signal would spend log2(npad)^2 / 2 barrier-bound passes over a window in global memory
// (minutes per signal at 2^24).  Instead: stage, then a bitonic network whose passes are launches over (pairs, signals) -- strides
// of at least SH_CH in global memory (k_sh_bitonic_g), all smaller strides of a merge step inside a chunk of SH_CH values in LDS
// (k_sh_bitonic_lds): 44 launches for 2^20 values, 90 for 2^24.  The selection kernels below then run on the sorted window (SORTED).
constexpr int SH_CH = 4096;
template <typename T>
__global__ __launch_bounds__(SH_NT) void k_sh_stage_g(const T *__restrict__ X, int64_t sig_stride, int n, const int *__restrict__ cols, int cnt,
                                                      int npad, T *__restrict__ gs)
{
    const T *x = X + (int64_t)blockIdx.y * sig_stride;
    T *v = gs + (int64_t)blockIdx.y * (2 * (int64_t)npad + 8);
    for (int64_t e = (int64_t)blockIdx.x * SH_NT + threadIdx.x; e < npad; e += (int64_t)gridDim.x * SH_NT) {
        T val = sh_inf<T>();
        if (e < cnt) {
            const int c = (int)(e / n), row = (int)(e - (int64_t)c * n);
            val = (T)fabs((double)x[(int64_t)(cols ? cols[c] : c) * n + row]);
        }
        v[e] = val;
    }
}
// one compare-exchange pass of stride j >= SH_CH of the merge step k
template <typename T>
__global__ __launch_bounds__(SH_NT) void k_sh_bitonic_g(T *__restrict__ gs, int npad, int k, int j)
{
    T *v = gs + (int64_t)blockIdx.y * (2 * (int64_t)npad + 8);
    for (int64_t i = (int64_t)blockIdx.x * SH_NT + threadIdx.x; i < (npad >> 1); i += (int64_t)gridDim.x * SH_NT) {
        const int64_t lo = ((i & ~(int64_t)(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
        const bool up = (lo & k) == 0;
        const T a = v[lo], b = v[hi];
        if ((a > b) == up) { v[lo] = b; v[hi] = a; }
    }
}
// chunk of SH_CH values in LDS: every merge step up to SH_CH (FIRST), or the strides below SH_CH of the merge step k
template <typename T, bool FIRST>
__global__ __launch_bounds__(SH_NT) void k_sh_bitonic_lds(T *__restrict__ gs, int npad, int k)
{
    __shared__ T w[SH_CH];
    T *v = gs + (int64_t)blockIdx.y * (2 * (int64_t)npad + 8) + (int64_t)blockIdx.x * SH_CH;
    const int64_t g0 = (int64_t)blockIdx.x * SH_CH;                        // global index of w[0]: the direction of a pair depends on it
    for (int e = threadIdx.x; e < SH_CH; e += SH_NT) w[e] = v[e];
    __syncthreads();
    for (int kk = FIRST ? 2 : k; kk <= (FIRST ? SH_CH : k); kk <<= 1)
        for (int j = (FIRST ? kk : SH_CH) >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < (SH_CH >> 1); i += SH_NT) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
                const bool up = ((g0 + lo) & kk) == 0;
                const T a = w[lo], b = w[hi];
                if ((a > b) == up) { w[lo] = b; w[hi] = a; }
            }
            __syncthreads();
        }
    for (int e = threadIdx.x; e < SH_CH; e += SH_NT) v[e] = w[e];
}

// ---- SureShrink: t = sqrt(a[argmin risk]) ---------------------------------------------------------------
template <typename T, bool SORTED = false>
__global__ __launch_bounds__(SH_NT) void k_surethreshold(const T *__restrict__ X, int64_t sig_stride, int n, const int *__restrict__ cols,
                                                         int ncols, int cnt, int npad, T *gscratch, T *__restrict__ tout)
{
    __shared__ ShShared<T> S;
    T *v = sh_window<T>(gscratch, npad), *b = v + npad;
    if (!SORTED) {
        sh_stage<T>(v, X + (int64_t)blockIdx.x * sig_stride, n, cols, ncols, cnt, npad);
        sh_bitonic<T>(v, npad);
    }
    sh_scan<T>(cnt, S.part, [&](int e) { const T a = v[e] * v[e]; return a; }, [&](int e, T run) { b[e] = run; });
    const int im = sh_argext<T, false>(cnt, S.rv, S.ri, [&](int i) {
        const T a = v[i] * v[i];
        const T ca = (T)(cnt - 1 - i) * a;                               // c .* a, rounded on its own
        const T s = b[i] + ca;
        return (T)((T)(cnt - 2 * (i + 1)) + s) / (T)cnt;
    });
    if (threadIdx.x == 0) { const T a = v[im] * v[im]; tout[blockIdx.x] = (T)sqrt((double)a); }
}

// ---- RelErrorShrink ---------------------------------------------------------------------------------------
template <typename T, bool SORTED = false>
__global__ __launch_bounds__(SH_NT) void k_relerrorthreshold(const T *__restrict__ X, int64_t sig_stride, int n,
                                                             const int *__restrict__ cols, int ncols, int cnt, int npad, int elbows,
                                                             T *gscratch, T *__restrict__ tout)
{
    __shared__ ShShared<T> S;
    T *v = sh_window<T>(gscratch, npad), *w = v + npad;                   // cnt + 1 <= npad + 8 entries
    if (!SORTED) {
        sh_stage<T>(v, X + (int64_t)blockIdx.x * sig_stride, n, cols, ncols, cnt, npad);
        sh_bitonic<T>(v, npad);
    }
    // orth2relerror: squares sorted downwards, sum, |sum - cumsum|^0.5 / sum^0.5; r_k lands at curve position cnt - k
    auto sq_desc = [&](int e) { const T a = v[cnt - 1 - e] * v[cnt - 1 - e]; return a; };
    const T total = sh_scan<T>(cnt, S.part, sq_desc, [&](int e, T run) { w[cnt - 1 - e] = run; });
    const T rt = (T)sqrt((double)total);
    for (int j = threadIdx.x; j < cnt; j += SH_NT) w[j] = (T)sqrt(fabs((double)(T)(total - w[j]))) / rt;
    __syncthreads();
    if (threadIdx.x == 0) w[cnt] = w[cnt - 1];                            // pushfirst!(r, r[1])
    __syncthreads();
    const int m = cnt + 1;
    const int iy = sh_argext<T, true>(m, S.rv, S.ri, [&](int j) { return w[j]; });
    const T ymax = w[iy], xmax = v[cnt - 1];
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += SH_NT) w[j] = w[j] / ymax;
    __syncthreads();
    auto xs = [&](int j) { return j == 0 ? (T)0 / xmax : v[j - 1] / xmax; };
    int last = m - 1;
    for (int e = 0; e < elbows; ++e) {
        const T x1 = xs(0), y1 = w[0];
        T vx = xs(last) - x1, vy = w[last] - y1;
        const T nv = (T)sqrt((double)(T)(vx * vx + vy * vy));
        vx = vx / nv; vy = vy / nv;
        last = sh_argext<T, true>(last + 1, S.rv, S.ri, [&](int j) {
            const T dx = xs(j) - x1, dy = w[j] - y1;
            const T dx2 = dx * dx, dy2 = dy * dy;
            const T H = (T)sqrt((double)(T)(dx2 + dy2));
            const T A = dx * vx + dy * vy;
            const T H2 = H * H, A2 = A * A;
            return (T)sqrt(fabs((double)(T)(H2 - A2)));
        });
    }
    if (threadIdx.x == 0) tout[blockIdx.x] = xs(last) * xmax;
}

int need_device()
{
    if (wx_device_count() < 1) return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    return WX_OK;
}

// kind 0: surethreshold, 1: relerrorthreshold
template <typename T>
int api_shrink(int kind, const T *X, int64_t n, int64_t k, int64_t batch, const uint8_t *colmask, int elbows, T *t, void *stream)
{
    WX_REQUIRE(n >= 1 && k >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    WX_REQUIRE(kind == 0 || elbows >= 1, WX_EASSERT, "@assert elbows >= 1 (Denoising.jl:291)");
    std::vector<int> cols;
    if (colmask) for (int64_t c = 0; c < k; ++c) if (colmask[c]) cols.push_back((int)c);
    const int64_t ncols = colmask ? (int64_t)cols.size() : k;
    const int64_t cnt = n * ncols;
    WX_REQUIRE(cnt >= 1, WX_EARG, "no coefficient selected");
    // one workgroup sorts one signal's selection in LDS (up to 8192 Float64 values) or in a global scratch window (WX_SHRINK_WG_MAX, a knob: nothing by default);
    // larger selections -- redundant tables with many leaf columns -- sort with launches over the whole chip (round 4: until then
    // 2^20 coefficients per signal were the limit, and the window path above 2^16 took seconds per signal)
    WX_REQUIRE(cnt <= ((int64_t)1 << 27), WX_EUNSUPPORTED, "threshold selection over more than 2^27 coefficients per signal");
    int rc;
    if ((rc = need_device())) return rc;
    if (batch == 0) return WX_OK;
    int64_t npad = 2;
    while (npad < cnt) npad <<= 1;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dX = (const T *)io.in(X, sizeof(T) * n * k * batch);
    T *dt = (T *)io.out(t, sizeof(T) * batch);
    if (!dX || !dt) return io.finish(WX_EHIP);
    const int *dcols = nullptr;
    if (colmask) {
        dcols = (const int *)scr.upload(cols.data(), cols.size() * sizeof(int));
        if (!dcols) return io.finish(WX_EHIP);
    }
    const size_t win = (size_t)(2 * npad + 8) * sizeof(T);              // sorted magnitudes + the curve (cnt + 1 points)
    const bool in_lds = win <= 144 * 1024;
    if (in_lds && win > 48 * 1024) {
        const void *f = kind == 0 ? reinterpret_cast<const void *>(k_surethreshold<T>) : reinterpret_cast<const void *>(k_relerrorthreshold<T>);
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)win) != hipSuccess)
            return io.finish(wx_set_error(WX_EHIP, "hipFuncSetAttribute(LDS)"));
    }
    // signals per launch when the windows live in global memory: at most 1 GiB of scratch
    int64_t per = batch;
    T *gs = nullptr;
    if (!in_lds) {
        per = ((int64_t)1 << 30) / (int64_t)win;
        if (per < 1) per = 1;
        if (per > batch) per = batch;
        gs = (T *)scr.alloc(win * per);
        if (!gs) return io.finish(WX_EHIP);
    }
    static const int64_t wg_max = wx_getenv("WX_SHRINK_WG_MAX") ? atoll(wx_getenv("WX_SHRINK_WG_MAX")) : ((int64_t)1 << 13);     // (2^16 values, 4 signals: 7.7 ms in the window, under 3 ms with launches)
    const bool chip_sort = !in_lds && npad > wg_max && npad >= 2 * SH_CH;
    for (int64_t b0 = 0; b0 < batch; b0 += per) {
        const int64_t nb = batch - b0 < per ? batch - b0 : per;
        if (chip_sort) {
            const unsigned gx = (unsigned)((npad / 2 + SH_NT - 1) / SH_NT < 65535 ? (npad / 2 + SH_NT - 1) / SH_NT : 65535);
            for (int64_t s0 = 0; s0 < nb; s0 += 65535) {                  // gridDim.y limit
                const unsigned ns = (unsigned)(nb - s0 < 65535 ? nb - s0 : 65535);
                T *g0 = gs + s0 * (2 * npad + 8);
                hipLaunchKernelGGL(k_sh_stage_g<T>, dim3(gx, ns), dim3(SH_NT), 0, st, dX + (b0 + s0) * n * k, n * k, (int)n, dcols, (int)cnt,
                                   (int)npad, g0);
                hipLaunchKernelGGL((k_sh_bitonic_lds<T, true>), dim3((unsigned)(npad / SH_CH), ns), dim3(SH_NT), 0, st, g0, (int)npad, 0);
                for (int64_t kk = 2 * SH_CH; kk <= npad; kk <<= 1) {
                    for (int64_t j = kk >> 1; j >= SH_CH; j >>= 1)
                        hipLaunchKernelGGL(k_sh_bitonic_g<T>, dim3(gx, ns), dim3(SH_NT), 0, st, g0, (int)npad, (int)kk, (int)j);
                    hipLaunchKernelGGL((k_sh_bitonic_lds<T, false>), dim3((unsigned)(npad / SH_CH), ns), dim3(SH_NT), 0, st, g0, (int)npad, (int)kk);
                }
            }
            if (kind == 0)
                hipLaunchKernelGGL((k_surethreshold<T, true>), dim3((unsigned)nb), dim3(SH_NT), 0, st, dX + b0 * n * k, n * k, (int)n,
                                   dcols, (int)ncols, (int)cnt, (int)npad, gs, dt + b0);
            else
                hipLaunchKernelGGL((k_relerrorthreshold<T, true>), dim3((unsigned)nb), dim3(SH_NT), 0, st, dX + b0 * n * k, n * k, (int)n,
                                   dcols, (int)ncols, (int)cnt, (int)npad, elbows, gs, dt + b0);
            if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "threshold selection kernel failed to launch"));
            continue;
        }
        if (kind == 0)
            hipLaunchKernelGGL(k_surethreshold<T>, dim3((unsigned)nb), dim3(SH_NT), in_lds ? win : 0, st, dX + b0 * n * k, n * k, (int)n,
                               dcols, (int)ncols, (int)cnt, (int)npad, gs, dt + b0);
        else
            hipLaunchKernelGGL(k_relerrorthreshold<T>, dim3((unsigned)nb), dim3(SH_NT), in_lds ? win : 0, st, dX + b0 * n * k, n * k, (int)n,
                               dcols, (int)ncols, (int)cnt, (int)npad, elbows, gs, dt + b0);
        if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "threshold selection kernel failed to launch"));
    }
    return io.finish(WX_OK);
}

}  // namespace

extern "C" {
int wx_surethreshold_f64(const double *X, int64_t n, int64_t k, int64_t batch, const uint8_t *colmask, double *t, void *stream)
{ return api_shrink<double>(0, X, n, k, batch, colmask, 1, t, stream); }
int wx_surethreshold_f32(const float *X, int64_t n, int64_t k, int64_t batch, const uint8_t *colmask, float *t, void *stream)
{ return api_shrink<float>(0, X, n, k, batch, colmask, 1, t, stream); }
int wx_relerrorthreshold_f64(const double *X, int64_t n, int64_t k, int64_t batch, const uint8_t *colmask, int elbows, double *t,
                             void *stream)
{ return api_shrink<double>(1, X, n, k, batch, colmask, elbows, t, stream); }
int wx_relerrorthreshold_f32(const float *X, int64_t n, int64_t k, int64_t batch, const uint8_t *colmask, int elbows, float *t,
                             void *stream)
{ return api_shrink<float>(1, X, n, k, batch, colmask, elbows, t, stream); }
}
