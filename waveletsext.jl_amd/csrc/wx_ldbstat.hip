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
#include "../../include/waveletsext_hip.h"     // the definitions below must match the public prototypes
#include "wx_common.h"
#include "wx_kernels.h"
#include "wx_host.h"
#include <vector>

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

extern "C" int wx_device_count(void);

namespace {

constexpr int LS_NT = 256;

template <typename T> __device__ __forceinline__ T ls_inf();
template <> __device__ __forceinline__ double ls_inf<double>() { return __longlong_as_double(0x7ff0000000000000LL); }
template <> __device__ __forceinline__ float ls_inf<float>() { return __int_as_float(0x7f800000); }

// per class, in device memory (any number of classes; 64 was the limit while the tables travelled as a kernel argument)
struct LsClasses {
    int nc;
    const int *cnt;         // signals per class
    const int *npad;        // power of two >= cnt
    const int *rowoff;      // offset of the class's TC rows in the window (elements)
    const int *sigoff;      // offset of the class in `order`
};
// The window of a workgroup -- the rows it sorts, the values it bins -- is LDS when it fits (GM = false) and a slice of a global
// scratch array when it does not (GM = true: more than about 10^4 signals per coefficient; the reference has no limit,
// ldb/ldb_measures.jl:481-519).  With GM a workgroup walks several coefficient groups (grid-stride), its window reused.
template <bool GM> __device__ __forceinline__ char *ls_window(char *lds, char *gwork, size_t wbytes)
{
    return GM ? gwork + (size_t)blockIdx.x * wbytes : lds;
}

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

template <typename T, bool GM>
__global__ __launch_bounds__(LS_NT) void k_class_median_mad(const T *__restrict__ X, int64_t nk, const int *__restrict__ order, LsClasses C,
                                                            int TC, int64_t row_elems, T *__restrict__ med, T *__restrict__ mad, char *gwork,
                                                            size_t wbytes)
{
    extern __shared__ __attribute__((aligned(16))) char ls_smem[];
    T *win = reinterpret_cast<T *>(ls_window<GM>(ls_smem, gwork, wbytes));
    T *meds = win + (int64_t)TC * row_elems;                // nc * TC medians behind the rows
    const int64_t ngroups = (nk + TC - 1) / TC;
    for (int64_t g0 = blockIdx.x; g0 < ngroups; g0 += gridDim.x) {
        int64_t bid = g0;
        if (!GM && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
        const int64_t e0 = bid * TC;
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
        __syncthreads();
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
template <typename T, bool GM>
__global__ __launch_bounds__(LS_NT) void k_emd_equal(const T *__restrict__ X, int64_t nk, const int *__restrict__ order, LsClasses C, int TC,
                                                     T *__restrict__ D, char *gwork, size_t wbytes)
{
    extern __shared__ __attribute__((aligned(16))) char ls_smem[];
    T *win = reinterpret_cast<T *>(ls_window<GM>(ls_smem, gwork, wbytes));
    __shared__ T red[LS_NT];
    const int64_t ngroups = (nk + TC - 1) / TC;
    for (int64_t g0 = blockIdx.x; g0 < ngroups; g0 += gridDim.x) {
    int64_t bid = g0;
    if (!GM && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int64_t e0 = bid * TC;
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
    __syncthreads();
    }
}

int need_device()
{
    if (wx_device_count() < 1) return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    return WX_OK;
}

// host copy of the class tables and their upload
struct LsHost {
    int nc = 0;
    std::vector<int> cnt, npad, rowoff, sigoff;
};
static bool ls_upload(const LsHost &H, WxScratch &scr, LsClasses *C)
{
    std::vector<int> all;
    all.reserve((size_t)4 * H.nc);
    all.insert(all.end(), H.cnt.begin(), H.cnt.end());
    all.insert(all.end(), H.npad.begin(), H.npad.end());
    all.insert(all.end(), H.rowoff.begin(), H.rowoff.end());
    all.insert(all.end(), H.sigoff.begin(), H.sigoff.end());
    const int *d = (const int *)scr.upload(all.data(), all.size() * sizeof(int));
    if (!d) return false;
    C->nc = H.nc;
    C->cnt = d; C->npad = d + H.nc; C->rowoff = d + 2 * (size_t)H.nc; C->sigoff = d + 3 * (size_t)H.nc;
    return true;
}
// signals class by class (stable) + the per-class counts, padded lengths and offsets (rowoff for one row per class)
static int ls_classes(const int32_t *cls, int64_t N, int nc, LsHost *H, std::vector<int> *order)
{
    WX_REQUIRE(cls != nullptr, WX_EARG, "NULL labels");
    WX_REQUIRE(nc > 1, WX_EASSERT, "@assert nc > 1");
    WX_REQUIRE(N >= 1 && N < ((int64_t)1 << 30), WX_EUNSUPPORTED, "2^30 signals or more");
    H->nc = nc;
    order->assign((size_t)N, 0);
    std::vector<int> offs((size_t)nc + 1, 0);
    for (int64_t i = 0; i < N; ++i) {
        WX_REQUIRE(cls[i] >= 0 && cls[i] < nc, WX_EARG, "class index outside [0, nc)");
        offs[(size_t)cls[i] + 1]++;
    }
    for (int c = 0; c < nc; ++c) {
        WX_REQUIRE(offs[(size_t)c + 1] > 0, WX_EARG, "a class has no signal");
        offs[(size_t)c + 1] += offs[(size_t)c];
    }
    std::vector<int> pos(offs.begin(), offs.end() - 1);
    for (int64_t i = 0; i < N; ++i) (*order)[(size_t)pos[(size_t)cls[i]]++] = (int)i;
    H->cnt.resize(nc); H->npad.resize(nc); H->rowoff.resize(nc); H->sigoff.resize(nc);
    int64_t off = 0;
    for (int c = 0; c < nc; ++c) {
        H->cnt[c] = offs[(size_t)c + 1] - offs[(size_t)c];
        H->sigoff[c] = offs[(size_t)c];
        int np = 2;
        while (np < H->cnt[c]) np <<= 1;
        H->npad[c] = np;
        H->rowoff[c] = (int)off;
        off += np;
        WX_REQUIRE(off < ((int64_t)1 << 31), WX_EUNSUPPORTED, "too many (padded) signals per coefficient");
    }
    return WX_OK;
}
// a global window per resident workgroup when the LDS window does not fit: at most 2 GiB of them
static int ls_global_windows(size_t wbytes, int64_t ngroups, WxScratch &scr, char **gwork, unsigned *grid)
{
    int64_t g = ((int64_t)2 << 30) / (int64_t)wbytes;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    if (g > ngroups) g = ngroups;
    *gwork = (char *)scr.alloc((size_t)g * wbytes);
    if (!*gwork) return WX_EHIP;
    *grid = (unsigned)g;
    return WX_OK;
}

// kind 0: class medians + mads, kind 1: earth mover's distance with equal weights
template <typename T>
int api_class_rows(int kind, const T *X, int64_t nk, int64_t N, const int32_t *cls, int nc, T *out1, T *out2, void *stream)
{
    WX_REQUIRE(nk >= 1 && N >= 1, WX_EARG, "bad dimensions");
    WX_REQUIRE(nk < ((int64_t)1 << 31), WX_EUNSUPPORTED, "more than 2^31 coefficients per signal");
    LsHost H;
    std::vector<int> order;
    int rc = ls_classes(cls, N, nc, &H, &order);
    if (rc) return rc;
    int64_t row_elems = 0;                                              // one row of every class
    for (int c = 0; c < nc; ++c) row_elems += H.npad[c];
    // LDS window: TC rows of every class + nc * TC medians; a global window (one coefficient per step) when one row does not fit
    const int64_t budget = (int64_t)(128 * 1024) / (int64_t)sizeof(T);
    const bool gm = row_elems + nc > budget;
    int TC = gm ? 1 : (int)(budget / (row_elems + nc));
    if (TC > 16) TC = 16;
    if (TC > nk) TC = (int)nk;
    {
        int64_t off = 0;
        for (int c = 0; c < nc; ++c) { H.rowoff[c] = (int)off; off += (int64_t)TC * H.npad[c]; }
    }
    if ((rc = need_device())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dX = (const T *)io.in(X, sizeof(T) * nk * N);
    T *d1 = (T *)io.out(out1, sizeof(T) * nk * (kind == 0 ? nc : 1));
    T *d2 = kind == 0 ? (T *)io.out(out2, sizeof(T) * nk * nc) : nullptr;
    if (!dX || !d1 || (kind == 0 && !d2)) return io.finish(WX_EHIP);
    const int *dorder = (const int *)scr.upload(order.data(), order.size() * sizeof(int));
    LsClasses C;
    if (!dorder || !ls_upload(H, scr, &C)) return io.finish(WX_EHIP);
    // (a multiple of 16: block b's global-memory window starts at gwork + b * wbytes and is used as T / double arrays)
    const size_t wbytes = ((((size_t)TC * row_elems + (size_t)nc * TC) * sizeof(T)) + 15) & ~(size_t)15;
    const int64_t ngroups = (nk + TC - 1) / TC;
    if (gm) {
        char *gwork = nullptr;
        unsigned grid = 0;
        if ((rc = ls_global_windows(wbytes, ngroups, scr, &gwork, &grid))) return io.finish(rc);
        if (kind == 0)
            hipLaunchKernelGGL((k_class_median_mad<T, true>), dim3(grid), dim3(LS_NT), 0, st, dX, nk, dorder, C, TC, row_elems, d1, d2, gwork, wbytes);
        else
            hipLaunchKernelGGL((k_emd_equal<T, true>), dim3(grid), dim3(LS_NT), 0, st, dX, nk, dorder, C, TC, d1, gwork, wbytes);
    } else {
        const void *f = kind == 0 ? reinterpret_cast<const void *>(k_class_median_mad<T, false>) : reinterpret_cast<const void *>(k_emd_equal<T, false>);
        if (wbytes > 48 * 1024 && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wbytes) != hipSuccess)
            return io.finish(wx_set_error(WX_EHIP, "hipFuncSetAttribute(LDS)"));
        if (kind == 0)
            hipLaunchKernelGGL((k_class_median_mad<T, false>), dim3((unsigned)ngroups), dim3(LS_NT), wbytes, st, dX, nk, dorder, C, TC, row_elems, d1, d2,
                               (char *)nullptr, (size_t)0);
        else
            hipLaunchKernelGGL((k_emd_equal<T, false>), dim3((unsigned)ngroups), dim3(LS_NT), wbytes, st, dX, nk, dorder, C, TC, d1, (char *)nullptr, (size_t)0);
    }
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

// ------------------------------------------------------------------------------------------------------------------
// Average shifted histograms: the ProbabilityDensity energy map and the weights of Signatures(:pdf)
//   energy_map(Xw, y, ProbabilityDensity())   ldb/ldb_energymap.jl:143-184
//       per coefficient j: z = the coefficient of ALL signals; sigma = std(z); nbins = ceil((30 Nx)^(1/5)),
//       mbins = ceil(100 / nbins), pdf_len = (nbins + 1) mbins; delta = (max z - min z + sigma) / (pdf_len - 1);
//       rng = range(min z - sigma / 2, step = delta, length = pdf_len); Gamma[j, :, c] = density of
//       ash(z of class c, rng = rng, m = mbins, kernel = Kernels.triangular)
//   energy_map(Xw, y, Signatures(:pdf))       ldb/ldb_energymap.jl:216-232: the same construction per class (sigma, min, max
//       of the class's own values), weight of signal k = pdf(epdf, z[k])
// `ash`, `xy` and `pdf` belong to AverageShiftedHistograms.jl (Project.toml compat "0.8, 0.9"), which is NOT in the
// reference tree.  Its published algorithm, restated: bin index of an observation ki = floor((y - first(rng)) / step + 1.5)
// (1-based; the points of rng are bin centres), counted if 1 <= ki <= length(rng); density[i] = sum over the non-empty
// bins k with |i - k| < m of counts[k] * kernel((i - k) / m), triangular kernel 1 - |u|; scaled by 1 / (sum(density) *
// step) so that it integrates to one; pdf(o, x) interpolates the density linearly between the two points of rng around x
// (0 outside).  Parity unpinned (no copy of that package and no Julia here): pinned by the oracle's restatement of the
// same text and by the defining properties (unit integral, weights inside the density's range).
// One workgroup per coefficient: its N values are staged in LDS (the loads of neighbouring coefficients share cache
// lines: XCD-aware order), mean / deviation / extrema are workgroup reductions, the class histograms are LDS atomics.
// ------------------------------------------------------------------------------------------------------------------
struct LsAsh { int nbins, mbins, len; };

template <typename T> __device__ T ls_block_red(T v, T *red, int op)          // op 0 sum, 1 min, 2 max
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = LS_NT >> 1; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const T a = red[threadIdx.x], b = red[threadIdx.x + s];
            red[threadIdx.x] = op == 0 ? a + b : op == 1 ? (b < a ? b : a) : (b > a ? b : a);
        }
        __syncthreads();
    }
    const T r = red[0];
    __syncthreads();
    return r;
}

// sigma (corrected, two passes), min, max of z[0, cnt)
__device__ void ls_moments(const double *z, int cnt, double *red, double &sigma, double &zmin, double &zmax)
{
    double s = 0, mn = ls_inf<double>(), mx = -ls_inf<double>();
    for (int i = threadIdx.x; i < cnt; i += LS_NT) { const double v = z[i]; s += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    const double mean = ls_block_red<double>(s, red, 0) / cnt;
    zmin = ls_block_red<double>(mn, red, 1);
    zmax = ls_block_red<double>(mx, red, 2);
    double q = 0;
    for (int i = threadIdx.x; i < cnt; i += LS_NT) { const double d = z[i] - mean; q += d * d; }
    sigma = sqrt(ls_block_red<double>(q, red, 0) / (cnt - 1));
}

// counts -> density (in place semantics of _ash!): dens[i], i in [0, len)
__device__ void ls_ash_density(const int *counts, double *dens, int len, int m, double delta, double *red)
{
    double part = 0;
    for (int i = threadIdx.x; i < len; i += LS_NT) {
        double d = 0;
        const int k0 = max(0, i - m + 1), k1 = min(len - 1, i + m - 1);
        for (int k = k0; k <= k1; ++k)
            if (counts[k]) d += counts[k] * (1.0 - fabs((double)(i - k) / m));
        dens[i] = d;
        part += d;
    }
    const double denom = 1.0 / (ls_block_red<double>(part, red, 0) * delta);
    for (int i = threadIdx.x; i < len; i += LS_NT) dens[i] *= denom;
    __syncthreads();
}

template <typename T, bool GM>
__global__ __launch_bounds__(LS_NT) void k_pdf_energy_map(const T *__restrict__ X, int64_t nk, int N, const int *__restrict__ cls, int nc,
                                                          LsAsh A, double *__restrict__ Gamma, char *gwork, size_t wbytes)
{
    extern __shared__ __attribute__((aligned(16))) char ls_smem[];
    double *z = reinterpret_cast<double *>(ls_window<GM>(ls_smem, gwork, wbytes));   // N values
    double *dens = z + N;                                                // len
    int *counts = reinterpret_cast<int *>(dens + A.len);                 // len
    __shared__ double red[LS_NT];
    for (int64_t g0 = blockIdx.x; g0 < nk; g0 += gridDim.x) {
    int64_t bid = g0;
    if (!GM && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int64_t e = bid;
    for (int i = threadIdx.x; i < N; i += LS_NT) z[i] = (double)X[e + nk * (int64_t)i];
    __syncthreads();
    double sigma, zmin, zmax;
    ls_moments(z, N, red, sigma, zmin, zmax);
    const double delta = (zmax - zmin + sigma) / (A.len - 1);
    const double a = zmin - 0.5 * sigma, dinv = 1.0 / delta;
    for (int c = 0; c < nc; ++c) {
        for (int i = threadIdx.x; i < A.len; i += LS_NT) counts[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += LS_NT)
            if (cls[i] == c) {
                const int ki = (int)floor((z[i] - a) * dinv + 1.5);
                if (ki >= 1 && ki <= A.len) atomicAdd(&counts[ki - 1], 1);
            }
        __syncthreads();
        ls_ash_density(counts, dens, A.len, A.mbins, delta, red);
        for (int i = threadIdx.x; i < A.len; i += LS_NT) Gamma[e + nk * ((int64_t)i + (int64_t)A.len * c)] = dens[i];
        __syncthreads();
    }
    }
}

// W[e, signal] = pdf(ash of the signal's class at coefficient e, X[e, signal])
template <typename T, bool GM>
__global__ __launch_bounds__(LS_NT) void k_signature_weights(const T *__restrict__ X, int64_t nk, int N, const int *__restrict__ order,
                                                             LsClasses C, LsAsh A, T *__restrict__ W, char *gwork, size_t wbytes)
{
    extern __shared__ __attribute__((aligned(16))) char ls_smem[];
    double *z = reinterpret_cast<double *>(ls_window<GM>(ls_smem, gwork, wbytes));   // the class's values (<= N)
    double *dens = z + N;
    int *counts = reinterpret_cast<int *>(dens + A.len);
    __shared__ double red[LS_NT];
    for (int64_t g0 = blockIdx.x; g0 < nk; g0 += gridDim.x) {
    int64_t bid = g0;
    if (!GM && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int64_t e = bid;
    for (int c = 0; c < C.nc; ++c) {
        const int cnt = C.cnt[c];
        const int *ord = order + C.sigoff[c];
        for (int i = threadIdx.x; i < cnt; i += LS_NT) z[i] = (double)X[e + nk * (int64_t)ord[i]];
        for (int i = threadIdx.x; i < A.len; i += LS_NT) counts[i] = 0;
        __syncthreads();
        double sigma, zmin, zmax;
        ls_moments(z, cnt, red, sigma, zmin, zmax);
        const double delta = (zmax - zmin + sigma) / (A.len - 1);
        const double a = zmin - 0.5 * sigma, dinv = 1.0 / delta;
        for (int i = threadIdx.x; i < cnt; i += LS_NT) {
            const int ki = (int)floor((z[i] - a) * dinv + 1.5);
            if (ki >= 1 && ki <= A.len) atomicAdd(&counts[ki - 1], 1);
        }
        __syncthreads();
        ls_ash_density(counts, dens, A.len, A.mbins, delta, red);
        for (int i = threadIdx.x; i < cnt; i += LS_NT) {
            // searchsortedlast(rng, x) with rng[j] = a + (j - 1) delta (1-based), then linear interpolation
            const double x = z[i];
            int j = (int)floor((x - a) * dinv) + 1;
            while (j >= 1 && a + (j - 1) * delta > x) --j;
            while (j < A.len && a + j * delta <= x) ++j;
            double w = 0.0;
            if (j >= 1 && j < A.len) {
                const double r0 = a + (j - 1) * delta, r1 = a + j * delta;
                w = dens[j - 1] + (dens[j] - dens[j - 1]) * (x - r0) / (r1 - r0);
            }
            W[e + nk * (int64_t)ord[i]] = (T)w;
        }
        __syncthreads();
    }
    }
}

// earth mover's distance with one weight per (coefficient, signal): rows of weights travel with the keys
template <typename T, bool GM>
__global__ __launch_bounds__(LS_NT) void k_emd_weighted(const T *__restrict__ X, const T *__restrict__ Wt, int64_t nk, const int *__restrict__ order,
                                                        LsClasses C, int64_t rows, T *__restrict__ D, char *gwork, size_t wbytes)
{
    extern __shared__ __attribute__((aligned(16))) char ls_smem[];
    T *win = reinterpret_cast<T *>(ls_window<GM>(ls_smem, gwork, wbytes));   // keys: one row per class (TC = 1)
    T *wts = win + rows;                                                 // weights, then their inclusive prefix sums
    __shared__ T red[LS_NT];
    for (int64_t g0 = blockIdx.x; g0 < nk; g0 += gridDim.x) {
    int64_t bid = g0;
    if (!GM && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int64_t e = bid;
    for (int c = 0; c < C.nc; ++c) {
        const int np = C.npad[c];
        for (int k = threadIdx.x; k < np; k += LS_NT) {
            T v = ls_inf<T>(), w = 0;
            if (k < C.cnt[c]) { const int64_t a = e + nk * (int64_t)order[C.sigoff[c] + k]; v = X[a]; w = Wt[a]; }
            win[C.rowoff[c] + k] = v; wts[C.rowoff[c] + k] = w;
        }
    }
    __syncthreads();
    {   // key-value bitonic sort of every row
        int npmax = 2;
        for (int c = 0; c < C.nc; ++c) npmax = max(npmax, C.npad[c]);
        for (int k = 2; k <= npmax; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int c = 0; c < C.nc; ++c) {
                    const int np = C.npad[c];
                    if (k > np) continue;
                    T *v = win + C.rowoff[c], *w = wts + C.rowoff[c];
                    for (int i = threadIdx.x; i < (np >> 1); i += LS_NT) {
                        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
                        const bool up = (lo & k) == 0;
                        const T a = v[lo], b = v[hi];
                        if ((a > b) == up) { v[lo] = b; v[hi] = a; const T t = w[lo]; w[lo] = w[hi]; w[hi] = t; }
                    }
                }
                __syncthreads();
            }
    }
    for (int c = threadIdx.x; c < C.nc; c += LS_NT) {                     // prefix sums of the sorted weights (sequential per row)
        T *w = wts + C.rowoff[c];
        T run = 0;
        for (int k = 0; k < C.cnt[c]; ++k) { run += w[k]; w[k] = run; }
    }
    __syncthreads();
    T total = 0;
    for (int c1 = 0; c1 < C.nc; ++c1)
        for (int c2 = c1 + 1; c2 < C.nc; ++c2) {
            const T *p = win + C.rowoff[c1], *q = win + C.rowoff[c2], *Pw = wts + C.rowoff[c1], *Qw = wts + C.rowoff[c2];
            const int n1 = C.cnt[c1], n2 = C.cnt[c2];
            T acc = 0;
            for (int i = threadIdx.x; i < n1 + n2; i += LS_NT) {
                T x, succ;
                if (i < n1) {
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
                    const int up = ls_upper<T>(p, n1, x), uq = ls_upper<T>(q, n2, x);
                    const T Fp = up ? Pw[up - 1] : (T)0, Fq = uq ? Qw[uq - 1] : (T)0;
                    acc += (T)fabs((double)(T)(Fp - Fq)) * (succ - x);
                }
            }
            const T s = ls_block_sum<T>(acc, red);
            total += s / (Pw[n1 - 1] + Qw[n2 - 1]);
        }
    if (threadIdx.x == 0) D[e] = total;
    __syncthreads();
    }
}

namespace {

LsAsh ls_ash_params(int64_t Nx)
{
    LsAsh A;
    A.nbins = (int)ceil(pow(30.0 * (double)Nx, 0.2));                    // ceil(Int, (30 Nx)^(1/5)), ldb_energymap.jl:158
    A.mbins = (int)ceil(100.0 / A.nbins);
    A.len = (A.nbins + 1) * A.mbins;
    return A;
}

template <typename T>
int api_pdf_map(const T *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *Gamma, void *stream)
{
    WX_REQUIRE(nk >= 1 && N >= 2, WX_EARG, "bad dimensions");
    WX_REQUIRE(nk < ((int64_t)1 << 31), WX_EUNSUPPORTED, "more than 2^31 coefficients per signal");
    LsHost H;
    std::vector<int> order;
    int rc = ls_classes(cls, N, nc, &H, &order);
    if (rc) return rc;
    const LsAsh A = ls_ash_params(N);
    const size_t wbytes = (sizeof(double) * ((size_t)N + A.len) + sizeof(int) * (size_t)A.len + 16 + 15) & ~(size_t)15;   // 16-byte windows
    const bool gm = wbytes > 150 * 1024;                    // more than about 19000 signals: the values of a coefficient in a global window
    if ((rc = need_device())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dX = (const T *)io.in(X, sizeof(T) * nk * N);
    double *dG = (double *)io.out(Gamma, sizeof(double) * nk * A.len * nc);
    if (!dX || !dG) return io.finish(WX_EHIP);
    const int *dcls = (const int *)scr.upload(cls, sizeof(int32_t) * (size_t)N);
    if (!dcls) return io.finish(WX_EHIP);
    if (gm) {
        char *gwork = nullptr;
        unsigned grid = 0;
        if ((rc = ls_global_windows(wbytes, nk, scr, &gwork, &grid))) return io.finish(rc);
        hipLaunchKernelGGL((k_pdf_energy_map<T, true>), dim3(grid), dim3(LS_NT), 0, st, dX, nk, (int)N, dcls, nc, A, dG, gwork, wbytes);
    } else {
        if (wbytes > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(k_pdf_energy_map<T, false>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)wbytes) != hipSuccess)
            return io.finish(wx_set_error(WX_EHIP, "hipFuncSetAttribute(LDS)"));
        hipLaunchKernelGGL((k_pdf_energy_map<T, false>), dim3((unsigned)nk), dim3(LS_NT), wbytes, st, dX, nk, (int)N, dcls, nc, A, dG, (char *)nullptr,
                           (size_t)0);
    }
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "density map kernel failed to launch"));
    return io.finish(WX_OK);
}

// kind 0: W = signature weights; kind 1: D = weighted earth mover's distance summed over the class pairs
template <typename T>
int api_signature(int kind, const T *X, const T *Win, int64_t nk, int64_t N, int64_t Ntot, const int32_t *cls, int nc, T *out, void *stream)
{
    WX_REQUIRE(nk >= 1 && N >= 2, WX_EARG, "bad dimensions");
    WX_REQUIRE(nk < ((int64_t)1 << 31), WX_EUNSUPPORTED, "more than 2^31 coefficients per signal");
    LsHost H;
    std::vector<int> order;
    int rc = ls_classes(cls, N, nc, &H, &order);
    if (rc) return rc;
    const LsAsh A = ls_ash_params(Ntot);
    int64_t rows = 0;
    for (int c = 0; c < nc; ++c) { WX_REQUIRE(kind == 1 || H.cnt[c] >= 2, WX_EARG, "a class needs two signals for its deviation"); rows += H.npad[c]; }
    const size_t wbytes = ((kind == 0 ? sizeof(double) * ((size_t)N + A.len) + sizeof(int) * (size_t)A.len : sizeof(T) * 2 * (size_t)rows) + 16 + 15) & ~(size_t)15;
    const bool gm = wbytes > 150 * 1024;
    if ((rc = need_device())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dX = (const T *)io.in(X, sizeof(T) * nk * N);
    const T *dW = kind == 1 ? (const T *)io.in(Win, sizeof(T) * nk * N) : nullptr;
    T *dout = (T *)io.out(out, sizeof(T) * nk * (kind == 0 ? N : 1));
    if (!dX || !dout || (kind == 1 && !dW)) return io.finish(WX_EHIP);
    const int *dorder = (const int *)scr.upload(order.data(), order.size() * sizeof(int));
    LsClasses C;
    if (!dorder || !ls_upload(H, scr, &C)) return io.finish(WX_EHIP);
    if (gm) {
        char *gwork = nullptr;
        unsigned grid = 0;
        if ((rc = ls_global_windows(wbytes, nk, scr, &gwork, &grid))) return io.finish(rc);
        if (kind == 0)
            hipLaunchKernelGGL((k_signature_weights<T, true>), dim3(grid), dim3(LS_NT), 0, st, dX, nk, (int)N, dorder, C, A, dout, gwork, wbytes);
        else
            hipLaunchKernelGGL((k_emd_weighted<T, true>), dim3(grid), dim3(LS_NT), 0, st, dX, dW, nk, dorder, C, rows, dout, gwork, wbytes);
    } else {
        const void *f = kind == 0 ? reinterpret_cast<const void *>(k_signature_weights<T, false>) : reinterpret_cast<const void *>(k_emd_weighted<T, false>);
        if (wbytes > 48 * 1024 && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wbytes) != hipSuccess)
            return io.finish(wx_set_error(WX_EHIP, "hipFuncSetAttribute(LDS)"));
        if (kind == 0)
            hipLaunchKernelGGL((k_signature_weights<T, false>), dim3((unsigned)nk), dim3(LS_NT), wbytes, st, dX, nk, (int)N, dorder, C, A, dout,
                               (char *)nullptr, (size_t)0);
        else
            hipLaunchKernelGGL((k_emd_weighted<T, false>), dim3((unsigned)nk), dim3(LS_NT), wbytes, st, dX, dW, nk, dorder, C, rows, dout, (char *)nullptr,
                               (size_t)0);
    }
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "signature kernel failed to launch"));
    return io.finish(WX_OK);
}

}  // namespace

extern "C" {
int wx_pdf_energy_map_f64(const double *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *Gamma, void *stream)
{ return api_pdf_map<double>(X, nk, N, cls, nc, Gamma, stream); }
int wx_pdf_energy_map_f32(const float *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *Gamma, void *stream)
{ return api_pdf_map<float>(X, nk, N, cls, nc, Gamma, stream); }
int wx_signature_weights_f64(const double *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *W, void *stream)
{ return api_signature<double>(0, X, nullptr, nk, N, N, cls, nc, W, stream); }
int wx_signature_weights_f32(const float *X, int64_t nk, int64_t N, const int32_t *cls, int nc, float *W, void *stream)
{ return api_signature<float>(0, X, nullptr, nk, N, N, cls, nc, W, stream); }
int wx_emd_measure_weighted_f64(const double *X, const double *W, int64_t nk, int64_t N, const int32_t *cls, int nc, double *D, void *stream)
{ return api_signature<double>(1, X, W, nk, N, N, cls, nc, D, stream); }
int wx_emd_measure_weighted_f32(const float *X, const float *W, int64_t nk, int64_t N, const int32_t *cls, int nc, float *D, void *stream)
{ return api_signature<float>(1, X, W, nk, N, N, cls, nc, D, stream); }
}
