// wx_ldb.hip -- Local Discriminant Basis, the batch-sized steps: SURVEY section 8(f) row 2.
//   energy_map(Xw, y, TimeFrequency())   ldb/ldb_energymap.jl:109-141: per class c,
//        Gamma[e, c] = sum_{i in c} Xw[e, i]^2 / sum_{i in c} norm(x_i)^2,   x_i = root column/slice of signal i
//   class-wise mean / variance of the basis coefficients for FishersClassSeparability
//        ldb/ldb_measures.jl:441-479 (mean(dims), var(dims): two passes, n-1 denominator)
// Everything else of fitdec! (discriminant measure on the small (n, k, classes) map, node costs with top_k,
// tree selection, ordering) is host logic on small arrays in the Python / Julia layer; the feature gather is
// wx_getbasiscoef* + an index selection.
#include "../../include/waveletsext_hip.h"     // the definitions below must match the public prototypes
#include "wx_common.h"
#include "wx_host.h"
#include "wx_kernels.h"

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

extern "C" int wx_device_count(void);

namespace {

constexpr int LDB_MAXC = 65535;          // gridDim.y = classes x chunks

// squared 2-norm of the root of every signal, as norm(x, 2)^2: sqrt then square
template <typename T>
__global__ __launch_bounds__(256) void k_ldb_root_norm2(const T *__restrict__ X, int64_t cnt, int64_t sig_stride,
                                                        T *__restrict__ nrm2)
{
    __shared__ double red[256];
    const T *x = X + (int64_t)blockIdx.x * sig_stride;
    const double acc = wx_sumsq_strided<T>(x, cnt);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) { const T r = (T)sqrt(red[0]); nrm2[blockIdx.x] = (T)(r * r); }
}

// per class: sum of the per-signal values and member count (one workgroup per class, fixed reduction tree)
template <typename T>
__global__ __launch_bounds__(256) void k_ldb_class_sum(const T *__restrict__ v, const int *__restrict__ cls, int64_t N,
                                                       int nc, T *__restrict__ out, T *__restrict__ count)
{
    __shared__ double red[256], redn[256];
    const int c = blockIdx.x;
    double acc = 0.0, cn = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += blockDim.x)
        if (cls[i] == c) { acc += (double)v[i]; cn += 1.0; }
    red[threadIdx.x] = acc; redn[threadIdx.x] = cn;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { red[threadIdx.x] += red[threadIdx.x + w]; redn[threadIdx.x] += redn[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[c] = (T)red[0]; count[c] = (T)redn[0]; }
}

// `order` lists the signals class by class (stable), `offs[c]..offs[c+1]` is class c's range; workgroup
// (blockIdx.y = c * nch + ch) sums chunk ch of class c:
// mode 0: partial[c][ch][e] = sum of X[e, i]^2,  mode 1: of X[e, i],  mode 2: of (X[e, i] - mean[c][e])^2
template <typename T>
__global__ __launch_bounds__(256) void k_ldb_class_partial(const T *__restrict__ X, int64_t nk,
                                                           const int *__restrict__ order, const int *__restrict__ offs,
                                                           int nch, int mode, const T *__restrict__ mean,
                                                           T *__restrict__ partial)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nk) return;
    const int c = blockIdx.y / nch, ch = blockIdx.y - c * nch;
    const int j0 = offs[c], cnt = offs[c + 1] - j0;
    const int per = (cnt + nch - 1) / nch;
    int a = j0 + ch * per, b = a + per;
    if (b > j0 + cnt) b = j0 + cnt;
    const T mu = mode == 2 ? mean[(int64_t)c * nk + e] : (T)0;
    T acc = 0;
    int j = a;
    for (; j + 4 <= b; j += 4) {                              // four independent loads in flight
        const T x0 = X[(int64_t)order[j] * nk + e], x1 = X[(int64_t)order[j + 1] * nk + e];
        const T x2 = X[(int64_t)order[j + 2] * nk + e], x3 = X[(int64_t)order[j + 3] * nk + e];
        if (mode == 1) { acc = (T)(acc + x0); acc = (T)(acc + x1); acc = (T)(acc + x2); acc = (T)(acc + x3); }
        else {
            const T d0 = (T)(x0 - mu), d1 = (T)(x1 - mu), d2 = (T)(x2 - mu), d3 = (T)(x3 - mu);
            acc = (T)(acc + (T)(d0 * d0)); acc = (T)(acc + (T)(d1 * d1));
            acc = (T)(acc + (T)(d2 * d2)); acc = (T)(acc + (T)(d3 * d3));
        }
    }
    for (; j < b; ++j) {
        const T x = X[(int64_t)order[j] * nk + e];
        if (mode == 1) acc = (T)(acc + x);
        else { const T d = (T)(x - mu); acc = (T)(acc + (T)(d * d)); }
    }
    partial[((int64_t)c * nch + ch) * nk + e] = acc;
}

// out[c][e] = (sum of the chunk partials in order) * scale, scale = 1/den[c] or 1/(den[c]-1) (variance)
template <typename T>
__global__ __launch_bounds__(256) void k_ldb_class_combine(const T *__restrict__ partial, int64_t nk, int nc, int nchunks,
                                                           const T *__restrict__ den, int minus_one,
                                                           T *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nk) return;
    for (int c = 0; c < nc; ++c) {
        T acc = 0;
        for (int j = 0; j < nchunks; ++j) acc = (T)(acc + partial[((int64_t)c * nchunks + j) * nk + e]);
        const T d = minus_one ? (T)(den[c] - (T)1) : den[c];
        out[(int64_t)c * nk + e] = (T)(acc / d);
    }
}

int need_device()
{
    if (wx_device_count() < 1) return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    return WX_OK;
}

int check_labels(const int32_t *cls, int64_t N, int nc, bool allow_empty)
{
    WX_REQUIRE(cls != nullptr, WX_EARG, "class labels are NULL");
    WX_REQUIRE(nc > 1, WX_EASSERT, "@assert nc > 1 (ldb_energymap.jl:122)");
    WX_REQUIRE(nc <= LDB_MAXC, WX_EUNSUPPORTED, "more than 65535 classes");
    std::vector<char> seen((size_t)nc, 0);
    for (int64_t i = 0; i < N; ++i) {
        WX_REQUIRE(cls[i] >= 0 && cls[i] < nc, WX_EARG, "class index outside [0, nc)");
        seen[(size_t)cls[i]] = 1;
    }
    if (!allow_empty) for (int c = 0; c < nc; ++c) WX_REQUIRE(seen[(size_t)c], WX_EARG, "a class has no signal");
    return WX_OK;
}

int pick_chunks(int64_t nk, int64_t N)
{
    // enough workgroups to fill the chip: ceil(nk/256) * nchunks >= ~2048, chunks of at least 64 signals
    int64_t gx = (nk + 255) / 256;
    int64_t want = (2048 + gx - 1) / gx;
    int64_t maxc = (N + 63) / 64;
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    return (int)want;
}

// mode 0: energy map (needs nroot), mode 1: class means, mode 2: class variances (needs mean)
template <typename T>
int api_class_reduce(const T *X, int64_t nk, int64_t nroot, int64_t N, const int32_t *cls, int nc, int mode,
                     const T *mean, T *out, T *den_out, void *stream)
{
    WX_REQUIRE(nk >= 1 && N >= 1, WX_EARG, "bad dimensions");
    int rc = check_labels(cls, N, nc, den_out != nullptr);      // a shard of a multi-GPU batch may miss a class
    if (rc) return rc;
    if ((rc = need_device())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dX = (const T *)io.in(X, sizeof(T) * nk * N);
    T *dout = (T *)io.out(out, sizeof(T) * nk * nc);
    const T *dmean = mode == 2 ? (const T *)io.in(mean, sizeof(T) * nk * nc) : nullptr;
    T *dden = den_out ? (T *)io.out(den_out, sizeof(T) * nc) : nullptr;
    if (!dX || !dout || (mode == 2 && !dmean) || (den_out && !dden)) return io.finish(WX_EHIP);
    // signals class by class in index order (the order the reference sums them in) + class offsets
    std::vector<int> order((size_t)N), offs((size_t)nc + 1, 0);
    for (int64_t i = 0; i < N; ++i) offs[(size_t)cls[i] + 1]++;
    for (int c = 0; c < nc; ++c) offs[(size_t)c + 1] += offs[(size_t)c];
    {
        std::vector<int> pos(offs.begin(), offs.end() - 1);
        for (int64_t i = 0; i < N; ++i) order[(size_t)pos[(size_t)cls[i]]++] = (int)i;
    }
    WX_REQUIRE(N < ((int64_t)1 << 31), WX_EUNSUPPORTED, "too many signals");
    const int *dcls = (const int *)scr.alloc(sizeof(int32_t) * (size_t)N);
    int *dorder = (int *)scr.alloc(sizeof(int) * (size_t)N);
    int *doffs = (int *)scr.alloc(sizeof(int) * ((size_t)nc + 1));
    if (!dcls || !dorder || !doffs) return io.finish(WX_EHIP);
    if (hipMemcpyAsync((void *)dcls, cls, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(dorder, order.data(), sizeof(int) * (size_t)N, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(doffs, offs.data(), sizeof(int) * ((size_t)nc + 1), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)                    // the host vectors go out of scope
        return io.finish(wx_set_error(WX_EHIP, "upload of the class index tables"));
    int nchunks = pick_chunks(nk, (N + nc - 1) / nc);
    if ((int64_t)nchunks * nc > 65535) nchunks = 65535 / nc;       // gridDim.y
    T *partial = (T *)scr.alloc(sizeof(T) * (size_t)nchunks * nc * nk);
    T *per_sig = (T *)scr.alloc(sizeof(T) * (size_t)N);
    T *den = (T *)scr.alloc(sizeof(T) * 2 * (size_t)nc);
    if (!partial || !per_sig || !den) return io.finish(WX_EHIP);
    T *count = den + nc;
    if (mode == 0) {
        WX_REQUIRE(nroot >= 1 && nroot <= nk, WX_EARG, "bad root size");
        hipLaunchKernelGGL(k_ldb_root_norm2<T>, dim3((unsigned)N), dim3(256), 0, st, dX, nroot, nk, per_sig);
    } else {
        if (hipMemsetAsync(per_sig, 0, sizeof(T) * (size_t)N, st) != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "memset"));
    }
    hipLaunchKernelGGL(k_ldb_class_sum<T>, dim3((unsigned)nc), dim3(256), 0, st, (const T *)per_sig, dcls, N, nc, den, count);
    hipLaunchKernelGGL(k_ldb_class_partial<T>, dim3((unsigned)((nk + 255) / 256), (unsigned)(nchunks * nc)), dim3(256), 0, st, dX,
                       nk, (const int *)dorder, (const int *)doffs, nchunks, mode, dmean, partial);
    hipLaunchKernelGGL(k_ldb_class_combine<T>, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, (const T *)partial, nk, nc,
                       nchunks, (const T *)(mode == 0 ? den : count), mode == 2 ? 1 : 0, dout);
    if (dden && hipMemcpyAsync(dden, mode == 0 ? den : count, sizeof(T) * nc, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return io.finish(wx_set_error(WX_EHIP, "copy of the class denominators"));
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "LDB reduction kernels failed to launch"));
    return io.finish(WX_OK);
}

}  // namespace

extern "C" {
int wx_energy_map_f64(const double *Xw, int64_t nk, int64_t nroot, int64_t N, const int32_t *cls, int nc, double *Gamma,
                      double *norm_sum, void *stream)
{ return api_class_reduce<double>(Xw, nk, nroot, N, cls, nc, 0, nullptr, Gamma, norm_sum, stream); }
int wx_energy_map_f32(const float *Xw, int64_t nk, int64_t nroot, int64_t N, const int32_t *cls, int nc, float *Gamma,
                      float *norm_sum, void *stream)
{ return api_class_reduce<float>(Xw, nk, nroot, N, cls, nc, 0, nullptr, Gamma, norm_sum, stream); }
int wx_class_mean_f64(const double *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *mean, void *stream)
{ return api_class_reduce<double>(X, nk, 0, N, cls, nc, 1, nullptr, mean, nullptr, stream); }
int wx_class_mean_f32(const float *X, int64_t nk, int64_t N, const int32_t *cls, int nc, float *mean, void *stream)
{ return api_class_reduce<float>(X, nk, 0, N, cls, nc, 1, nullptr, mean, nullptr, stream); }
int wx_class_var_f64(const double *X, int64_t nk, int64_t N, const int32_t *cls, int nc, const double *mean, double *var, void *stream)
{ return api_class_reduce<double>(X, nk, 0, N, cls, nc, 2, mean, var, nullptr, stream); }
int wx_class_var_f32(const float *X, int64_t nk, int64_t N, const int32_t *cls, int nc, const float *mean, float *var, void *stream)
{ return api_class_reduce<float>(X, nk, 0, N, cls, nc, 2, mean, var, nullptr, stream); }
}
