// wx_denoise.hip -- denoising core on the device: SURVEY section 8(f) row 1 (threshold between the forward and
// the inverse transform; both transforms are the batch entry points of the other translation units).
//   noisest(x, redundant, tree)      Denoising.jl:214-232  = Wavelets.Threshold.mad!(dr) / 0.6745
//   threshold!(x, TH, t)             Wavelets.jl Threshold (HardTH / SoftTH / SemiSoftTH / SteinTH), applied as
//                                    in denoise(), Denoising.jl:483-599, to the rows / columns it selects
// Wavelets.jl is not vendored in the reference tree: mad! and the four threshold loops are restated from its
// published source (parity unpinned, like the filter tables).
#include "wx_common.h"
#include "wx_host.h"
#include "wx_kernels.h"
#include <vector>

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

extern "C" int wx_device_count(void);

namespace {

// X (n, k, batch): rows [row_lo, n) of the selected columns of every signal, in place
template <typename T>
__global__ __launch_bounds__(256) void k_threshold(T *__restrict__ X, int n, int k, int64_t batch, int th_kind,
                                                   const T *__restrict__ t, int per_signal, int row_lo,
                                                   const int *__restrict__ cols, int ncols)
{
    const int rows = n - row_lo;
    const int64_t per_sig = (int64_t)rows * ncols;
    const int64_t total = per_sig * batch;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t sig = g / per_sig;
        const int64_t rem = g - sig * per_sig;
        const int c = (int)(rem / rows), r = (int)(rem - (int64_t)c * rows) + row_lo;
        T *p = X + (sig * k + (cols ? cols[c] : c)) * (int64_t)n + r;
        const T v = *p, tt = per_signal ? t[sig] : t[0];
        T out;
        if (th_kind == 0) out = (T)fabs((double)v) <= tt ? (T)0 : v;
        else {
            const T sg = v > (T)0 ? (T)1 : (v < (T)0 ? (T)-1 : v);
            if (th_kind == 1) { const T sh = (T)((T)fabs((double)v) - tt); out = sh < (T)0 ? (T)0 : (T)(sg * sh); }
            else if (th_kind == 2) {
                const T sh = (T)((T)(v * v) - (T)(tt * tt));
                out = sh < (T)0 ? (T)0 : (T)(sg * (T)sqrt((double)sh));
            } else {
                const T sh = (T)((T)1 - (T)((T)(tt * tt) / (T)(v * v)));
                out = sh < (T)0 ? (T)0 : (T)(v * sh);
            }
        }
        if (out != v || th_kind != 0) *p = out;
    }
}

template <typename T> __device__ __forceinline__ void bitonic_sort_lds(T *v, int P)
{
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < (P >> 1); i += blockDim.x) {
                const int lo = 2 * i - (i & (stride - 1));         // index with the `stride` bit clear
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const T a = v[lo], b = v[hi];
                if ((a > b) == up) { v[lo] = b; v[hi] = a; }
            }
        }
    __syncthreads();
}

// one workgroup per signal: median, absolute deviations, median again -- exact order statistics, so the result
// equals the reference's partial sorts bit for bit
template <typename T>
__global__ __launch_bounds__(1024) void k_mad(const T *__restrict__ X, int64_t sig_stride, int64_t off, int cnt, int P,
                                              T *__restrict__ sigma)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    T *v = reinterpret_cast<T *>(wx_smem);
    const T *x = X + (int64_t)blockIdx.x * sig_stride + off;
    const T inf = (T)INFINITY;
    for (int i = threadIdx.x; i < P; i += blockDim.x) v[i] = i < cnt ? x[i] : inf;
    bitonic_sort_lds<T>(v, P);
    const T m = (cnt & 1) ? v[cnt / 2] : (T)((T)(v[cnt / 2 - 1] / (T)2) + (T)(v[cnt / 2] / (T)2));
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) v[i] = (T)fabs((double)(T)(v[i] - m));
    bitonic_sort_lds<T>(v, P);
    if (threadIdx.x == 0) {
        const T r = (cnt & 1) ? v[cnt / 2] : (T)((T)(v[cnt / 2 - 1] / (T)2) + (T)(v[cnt / 2] / (T)2));
        sigma[blockIdx.x] = (T)(r / (T)0.6745);
    }
}

int need_device()
{
    if (wx_device_count() < 1) return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    return WX_OK;
}

template <typename T>
int api_noisest(const T *X, int64_t n, int64_t k, int64_t batch, int64_t row_lo, int64_t col, T *sigma, void *stream)
{
    WX_REQUIRE(n >= 1 && k >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    WX_REQUIRE(wx_isdyadic(n), WX_EASSERT, "@assert isdyadic(size(x,1)) (Denoising.jl:218)");
    WX_REQUIRE(0 <= row_lo && row_lo < n && 0 <= col && col < k, WX_EBOUNDS, "detail range outside the array");
    const int64_t cnt = n - row_lo;
    int64_t P = 2;
    while (P < cnt) P <<= 1;
    WX_REQUIRE((size_t)P * sizeof(T) <= 128 * 1024, WX_EUNSUPPORTED, "noisest: more detail coefficients than fit the LDS of one CU");
    int rc;
    if ((rc = need_device())) return rc;
    if (batch == 0) return WX_OK;
    hipStream_t st = wx_stream(stream);
    WxIO io(st);
    const T *dX = (const T *)io.in(X, sizeof(T) * n * k * batch);
    T *ds = (T *)io.out(sigma, sizeof(T) * batch);
    if (!dX || !ds) return io.finish(WX_EHIP);
    const size_t lds = (size_t)P * sizeof(T);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mad<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "hipFuncSetAttribute(LDS)"));
    }
    const int nt = P >= 2048 ? 1024 : (P >= 512 ? 256 : 64);
    hipLaunchKernelGGL(k_mad<T>, dim3((unsigned)batch), dim3(nt), lds, st, dX, n * k, col * n + row_lo, (int)cnt, (int)P, ds);
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "noisest kernel failed to launch"));
    return io.finish(WX_OK);
}

template <typename T>
int api_threshold(T *X, int64_t n, int64_t k, int64_t batch, int th_kind, const T *t, int64_t nt, int64_t row_lo,
                  const uint8_t *colmask, void *stream)
{
    WX_REQUIRE(n >= 1 && k >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    WX_REQUIRE(th_kind >= 0 && th_kind <= 3, WX_EARG, "th_kind: 0 HardTH, 1 SoftTH, 2 SemiSoftTH, 3 SteinTH");
    WX_REQUIRE(t != nullptr && (nt == 1 || nt == batch), WX_EARG, "one threshold, or one per signal");
    WX_REQUIRE(0 <= row_lo && row_lo <= n, WX_EBOUNDS, "row range outside the array");
    WX_REQUIRE(n < ((int64_t)1 << 31) && k < ((int64_t)1 << 31), WX_EUNSUPPORTED, "array too large");
    int rc;
    if ((rc = need_device())) return rc;
    std::vector<int> cols;
    if (colmask) for (int64_t c = 0; c < k; ++c) if (colmask[c]) cols.push_back((int)c);
    const int ncols = colmask ? (int)cols.size() : (int)k;
    if (batch == 0 || ncols == 0 || row_lo == n) return WX_OK;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    T *dX = (T *)io.in(X, sizeof(T) * n * k * batch);
    for (auto &it : io.items) if (it.user == X) it.copy_out = true;
    const T *dt = (const T *)io.in(t, sizeof(T) * nt);
    if (!dX || !dt) return io.finish(WX_EHIP);
    const int *dcols = nullptr;
    if (colmask) {
        dcols = (const int *)scr.upload(cols.data(), cols.size() * sizeof(int));
        if (!dcols) return io.finish(WX_EHIP);
    }
    const int64_t total = (n - row_lo) * ncols * batch;
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(k_threshold<T>, dim3((unsigned)grid), dim3(256), 0, st, dX, (int)n, (int)k, batch, th_kind, dt,
                       nt == batch && batch > 1 ? 1 : (nt == batch ? 1 : 0), (int)row_lo, dcols, ncols);
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "threshold kernel failed to launch"));
    return io.finish(WX_OK);
}

}  // namespace

extern "C" {
int wx_noisest_f64(const double *X, int64_t n, int64_t k, int64_t batch, int64_t row_lo, int64_t col, double *sigma, void *stream)
{ return api_noisest<double>(X, n, k, batch, row_lo, col, sigma, stream); }
int wx_noisest_f32(const float *X, int64_t n, int64_t k, int64_t batch, int64_t row_lo, int64_t col, float *sigma, void *stream)
{ return api_noisest<float>(X, n, k, batch, row_lo, col, sigma, stream); }
int wx_threshold_f64(double *X, int64_t n, int64_t k, int64_t batch, int th_kind, const double *t, int64_t nt, int64_t row_lo,
                     const uint8_t *colmask, void *stream)
{ return api_threshold<double>(X, n, k, batch, th_kind, t, nt, row_lo, colmask, stream); }
int wx_threshold_f32(float *X, int64_t n, int64_t k, int64_t batch, int th_kind, const float *t, int64_t nt, int64_t row_lo,
                     const uint8_t *colmask, void *stream)
{ return api_threshold<float>(X, n, k, batch, th_kind, t, nt, row_lo, colmask, stream); }
}
