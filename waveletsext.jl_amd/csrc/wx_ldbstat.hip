// wx_ldbstat.hip -- order statistics over the signal axis for Local Discriminant Basis: the robust Fisher power and the
// earth mover's distance between class signatures (the measures round 1 left on the host).
//
// Reference (paths relative to /root/reference/src/mod):
//   discriminant_power(coefs, y, RobustFishersClassSeparability())   ldb/ldb_measures.jl:481-519
//       per coefficient and class: median and mad(normalize = false) over the class's signals
//   energy_map(Xw, y, Signatures(:equal))                            ldb/ldb_energymap.jl:186-238 (coefficients by class, weight 1/Nc)
//   discriminant_measure(Gamma, EarthMoverDistance())                ldb/ldb_measures.jl:185-201, 254-285, 327-360
//       per coefficient, summed over the class pairs: sum_i |W_p(r_i) - W_q(r_i)| (r_{i+1} - r_i) / (sum w_p + sum w_q),
//       r = the sorted union of the two classes' values, W = the weight of the values <= r_i
// Both need, for every coefficient of the (n, levels) table, the values of every class SORTED along the signal axis.
// One workgroup takes TC consecutive coefficients (so that the loads of a signal are one contiguous run), stages the
// rows class by class in LDS (+Inf padding to a power of two), sorts all rows at once with a bitonic network and then
//   * medians: the middle element(s) of the row (middle(a, b) = a/2 + b/2 as Statistics.median), the absolute
//     deviations written back in place, sorted again, median again -- exact order statistics, equal to the reference's;
//   * earth mover's distance: every element x of the union contributes |W_p(x) - W_q(x)| (succ(x) - x), where succ is its
//     successor in the merged order and the cumulative weights are upper-bound ranks (binary searches in the two sorted
//     rows) -- no merged array is built, every element is independent, the sum is a workgroup reduction (the reference
//     adds the gaps in merged order: the difference is rounding, 1e-12 in the tests).
// Workgroups that share cache lines of the table (consecutive coefficients) are placed on the same XCD.
#include "wx_common.h"
#include "wx_kernels.h"
#include "wx_host.h"
#include <vector>

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

extern "C" int wx_device_count(void);

namespace {

constexpr int LS_NT = 256;
constexpr int LS_MAXC = 64;

template <typename T> __device__ __forceinline__ T ls_inf();
template <> __device__ __forceinline__ double ls_inf<double>() { return __longlong_as_double(0x7ff0000000000000LL); }
template <> __device__ __forceinline__ float ls_inf<float>() { return __int_as_float(0x7f800000); }

struct LsClasses {
    int nc;
    int cnt[LS_MAXC];       // signals per class
    int npad[LS_MAXC];      // power of two >= cnt
    int rowoff[LS_MAXC];    // offset of the class's TC rows in the window (elements)
    int sigoff[LS_MAXC];    // offset of the class in `order`
};

template <typename T>
__device__ void ls_stage(T *win, const T *__restrict__ X, int64_t nk, int64_t e0, int tc, int TC, const int *__restrict__ order,
                         const LsClasses &C)
{
    for (int c = 0; c < C.nc; ++c) {
        T *rows = win + C.rowoff[c];
        const int np = C.npad[c];
        for (int idx = threadIdx.x; idx < TC * np; idx += LS_NT) {
            const int te = idx % TC, k = idx / TC;
            T v = ls_inf<T>();
            if (te < tc && k < C.cnt[c]) v = X[e0 + te + nk * (int64_t)order[C.sigoff[c] + k]];
            rows[te * np + k] = v;
        }
    }
    __syncthreads();
}

// all TC rows of all classes at once; rows of one class have the same padded length
template <typename T> __device__ void ls_sort(T *win, int TC, const LsClasses &C)
{
    int npmax = 2;
    for (int c = 0; c < C.nc; ++c) npmax = max(npmax, C.npad[c]);
    for (int k = 2; k <= npmax; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int c = 0; c < C.nc; ++c) {
                const int np = C.npad[c];
                if (k > np) continue;
                T *rows = win + C.rowoff[c];
                const int half = np >> 1;
                for (int idx = threadIdx.x; idx < TC * half; idx += LS_NT) {
                    const int te = idx / half, i = idx - te * half;
                    const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
                    const bool up = (lo & k) == 0;
                    T *v = rows + te * np;
                    const T a = v[lo], b = v[hi];
                    if ((a > b) == up) { v[lo] = b; v[hi] = a; }
                }
            }
            __syncthreads();
        }
}

template <typename T> __device__ __forceinline__ T ls_median_sorted(const T *v, int cnt)
{
    if (cnt & 1) return v[cnt >> 1];
    const T a = v[(cnt >> 1) - 1], b = v[cnt >> 1];
    return (T)((T)(a / (T)2) + (T)(b / (T)2));
}

template <typename T>
__global__ __launch_bounds__(LS_NT) void k_class_median_mad(const T *__restrict__ X, int64_t nk, const int *__restrict__ order, LsClasses C,
                                                            int TC, T *__restrict__ med, T *__restrict__ mad)
{
    extern __shared__ __attribute__((aligned(16))) char ls_smem[];
    T *win = reinterpret_cast<T *>(ls_smem);
    __shared__ T meds[LS_MAXC * 16];
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int64_t e0 = (int64_t)bid * TC;
    const int tc = (int)min((int64_t)TC, nk - e0);
    ls_stage<T>(win, X, nk, e0, tc, TC, order, C);
    ls_sort<T>(win, TC, C);
    for (int idx = threadIdx.x; idx < C.nc * TC; idx += LS_NT) {
        const int c = idx / TC, te = idx - c * TC;
        const T m = ls_median_sorted<T>(win + C.rowoff[c] + te * C.npad[c], C.cnt[c]);
        meds[idx] = m;
        if (te < tc) med[e0 + te + nk * c] = m;
    }
    __syncthreads();
    for (int c = 0; c < C.nc; ++c) {
        T *rows = win + C.rowoff[c];
        const int np = C.npad[c];
        for (int idx = threadIdx.x; idx < TC * C.cnt[c]; idx += LS_NT) {
            const int te = idx / C.cnt[c], k = idx - te * C.cnt[c];
            rows[te * np + k] = (T)fabs((double)(T)(rows[te * np + k] - meds[c * TC + te]));
        }
    }
    __syncthreads();
    ls_sort<T>(win, TC, C);
    for (int idx = threadIdx.x; idx < C.nc * TC; idx += LS_NT) {
        const int c = idx / TC, te = idx - c * TC;
        if (te < tc) mad[e0 + te + nk * c] = ls_median_sorted<T>(win + C.rowoff[c] + te * C.npad[c], C.cnt[c]);
    }
}

// first index with v[i] >= x / v[i] > x in the sorted v[0, cnt)
template <typename T> __device__ __forceinline__ int ls_lower(const T *v, int cnt, T x)
{
    int lo = 0, hi = cnt;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (v[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
template <typename T> __device__ __forceinline__ int ls_upper(const T *v, int cnt, T x)
{
    int lo = 0, hi = cnt;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (v[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}

template <typename T> __device__ T ls_block_sum(T v, T *red)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = LS_NT >> 1; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const T r = red[0];
    __syncthreads();
    return r;
}

// equal weights (Signatures(:equal)): w_c = 1 / N_c
template <typename T>
__global__ __launch_bounds__(LS_NT) void k_emd_equal(const T *__restrict__ X, int64_t nk, const int *__restrict__ order, LsClasses C, int TC,
                                                     T *__restrict__ D)
{
    extern __shared__ __attribute__((aligned(16))) char ls_smem[];
    T *win = reinterpret_cast<T *>(ls_smem);
    __shared__ T red[LS_NT];
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int64_t e0 = (int64_t)bid * TC;
    const int tc = (int)min((int64_t)TC, nk - e0);
    ls_stage<T>(win, X, nk, e0, tc, TC, order, C);
    ls_sort<T>(win, TC, C);
    for (int te = 0; te < tc; ++te) {
        T total = 0;
        for (int c1 = 0; c1 < C.nc; ++c1)
            for (int c2 = c1 + 1; c2 < C.nc; ++c2) {
                const T *p = win + C.rowoff[c1] + te * C.npad[c1], *q = win + C.rowoff[c2] + te * C.npad[c2];
                const int n1 = C.cnt[c1], n2 = C.cnt[c2];
                const T wp = (T)1 / (T)n1, wq = (T)1 / (T)n2;
                T acc = 0;
                for (int i = threadIdx.x; i < n1 + n2; i += LS_NT) {
                    T x, succ;
                    if (i < n1) {                                       // element of p: p's tie with q sorts first
                        x = p[i];
                        const int lb = ls_lower<T>(q, n2, x);
                        succ = i + 1 < n1 ? p[i + 1] : ls_inf<T>();
                        if (lb < n2 && q[lb] < succ) succ = q[lb];
                    } else {
                        const int j = i - n1;
                        x = q[j];
                        const int ub = ls_upper<T>(p, n1, x);
                        succ = j + 1 < n2 ? q[j + 1] : ls_inf<T>();
                        if (ub < n1 && p[ub] < succ) succ = p[ub];
                    }
                    if (succ < ls_inf<T>()) {
                        const T Fp = wp * (T)ls_upper<T>(p, n1, x), Fq = wq * (T)ls_upper<T>(q, n2, x);
                        acc += (T)fabs((double)(T)(Fp - Fq)) * (succ - x);
                    }
                }
                const T s = ls_block_sum<T>(acc, red);
                total += s / (wp * (T)n1 + wq * (T)n2);
            }
        if (threadIdx.x == 0) D[e0 + te] = total;
    }
}

int need_device()
{
    if (wx_device_count() < 1) return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    return WX_OK;
}

// kind 0: class medians + mads, kind 1: earth mover's distance with equal weights
template <typename T>
int api_class_rows(int kind, const T *X, int64_t nk, int64_t N, const int32_t *cls, int nc, T *out1, T *out2, void *stream)
{
    WX_REQUIRE(nk >= 1 && N >= 1, WX_EARG, "bad dimensions");
    WX_REQUIRE(cls != nullptr, WX_EARG, "NULL labels");
    WX_REQUIRE(nc > 1, WX_EASSERT, "@assert nc > 1");
    WX_REQUIRE(nc <= LS_MAXC, WX_EUNSUPPORTED, "more than 64 classes");
    WX_REQUIRE(N < ((int64_t)1 << 31), WX_EUNSUPPORTED, "too many signals");
    LsClasses C;
    C.nc = nc;
    std::vector<int> order((size_t)N), offs((size_t)nc + 1, 0);
    for (int64_t i = 0; i < N; ++i) {
        WX_REQUIRE(cls[i] >= 0 && cls[i] < nc, WX_EARG, "class index outside [0, nc)");
        offs[(size_t)cls[i] + 1]++;
    }
    for (int c = 0; c < nc; ++c) {
        WX_REQUIRE(offs[(size_t)c + 1] > 0, WX_EARG, "a class has no signal");
        offs[(size_t)c + 1] += offs[(size_t)c];
    }
    {
        std::vector<int> pos(offs.begin(), offs.end() - 1);
        for (int64_t i = 0; i < N; ++i) order[(size_t)pos[(size_t)cls[i]]++] = (int)i;
    }
    int64_t row_elems = 0;                                              // one row of every class
    for (int c = 0; c < nc; ++c) {
        C.cnt[c] = offs[(size_t)c + 1] - offs[(size_t)c];
        C.sigoff[c] = offs[(size_t)c];
        int np = 2;
        while (np < C.cnt[c]) np <<= 1;
        C.npad[c] = np;
        row_elems += np;
    }
    const int64_t budget = (int64_t)(128 * 1024) / (int64_t)sizeof(T);
    WX_REQUIRE(row_elems <= budget, WX_EUNSUPPORTED, "the signals of one coefficient (padded per class) exceed the 128 KiB LDS window");
    int TC = (int)(budget / row_elems);
    if (TC > 16) TC = 16;
    if (TC > nk) TC = (int)nk;
    {
        int off = 0;
        for (int c = 0; c < nc; ++c) { C.rowoff[c] = off; off += TC * C.npad[c]; }
    }
    int rc;
    if ((rc = need_device())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dX = (const T *)io.in(X, sizeof(T) * nk * N);
    T *d1 = (T *)io.out(out1, sizeof(T) * nk * (kind == 0 ? nc : 1));
    T *d2 = kind == 0 ? (T *)io.out(out2, sizeof(T) * nk * nc) : nullptr;
    if (!dX || !d1 || (kind == 0 && !d2)) return io.finish(WX_EHIP);
    const int *dorder = (const int *)scr.upload(order.data(), order.size() * sizeof(int));
    if (!dorder) return io.finish(WX_EHIP);
    const size_t lds = (size_t)TC * row_elems * sizeof(T);
    const void *f = kind == 0 ? reinterpret_cast<const void *>(k_class_median_mad<T>) : reinterpret_cast<const void *>(k_emd_equal<T>);
    if (lds > 48 * 1024 && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return io.finish(wx_set_error(WX_EHIP, "hipFuncSetAttribute(LDS)"));
    const unsigned grid = (unsigned)((nk + TC - 1) / TC);
    if (kind == 0)
        hipLaunchKernelGGL(k_class_median_mad<T>, dim3(grid), dim3(LS_NT), lds, st, dX, nk, dorder, C, TC, d1, d2);
    else
        hipLaunchKernelGGL(k_emd_equal<T>, dim3(grid), dim3(LS_NT), lds, st, dX, nk, dorder, C, TC, d1);
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "LDB order-statistics kernel failed to launch"));
    return io.finish(WX_OK);
}

}  // namespace

extern "C" {
int wx_class_median_mad_f64(const double *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *med, double *mad, void *stream)
{ return api_class_rows<double>(0, X, nk, N, cls, nc, med, mad, stream); }
int wx_class_median_mad_f32(const float *X, int64_t nk, int64_t N, const int32_t *cls, int nc, float *med, float *mad, void *stream)
{ return api_class_rows<float>(0, X, nk, N, cls, nc, med, mad, stream); }
int wx_emd_measure_f64(const double *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *D, void *stream)
{ return api_class_rows<double>(1, X, nk, N, cls, nc, D, nullptr, stream); }
int wx_emd_measure_f32(const float *X, int64_t nk, int64_t N, const int32_t *cls, int nc, float *D, void *stream)
{ return api_class_rows<float>(1, X, nk, N, cls, nc, D, nullptr, stream); }
}
