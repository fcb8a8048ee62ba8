// wx_common.h -- shared declarations of the MI355X (gfx950) wavelet-packet kernels.
// CDNA4 only: wave64, 160 KiB LDS per CU, no MFMA (bandwidth-bound stencils).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

// Tuning / diagnostic knobs.  The library reads WX_* environment variables ONLY when the process also sets WX_KNOBS=1
// (tools/, the A/B tests of tests/): a production process's dispatch does not depend on its environment, and the header's
// statement about library state holds.  Each knob is read once, at its first use.
static inline const char *wx_getenv(const char *name)
{
    static const bool on = [] { const char *k = getenv("WX_KNOBS"); return k && atoi(k) != 0; }();
    return on ? getenv(name) : nullptr;
}

#define WX_MAXF 64            // longest supported QMF (even length)

// Rotation stages the lattice kernels are BUILT for: 1, 2, 4, 6, 8, 10 (round 6; every count 1 ... 10 until then).  A filter of 3, 5, 7 or
// 9 stages (db3 / coif2, db5, db7, db9 / coif6 ...) runs on the kernel of the next even count: wx_lattice_factor fills the stages a
// filter does not have with p = kap = 0, and a rotation by zero followed by the unit advance of the odd channel is undone exactly by the
// realignment at the end of the level -- fma(0, x, y) = y: the results are bit-identical to a kernel of the filter's own length, at the
// cost of one idle stage (db7 runs like db8).  40 % of the lattice instantiations, which are most of the library's build time and size.
static constexpr int wx_lat_stages_of(int ns) { return ns <= 2 ? ns : ((ns + 1) & ~1); }
static constexpr bool wx_lat_built(int ns) { return wx_lat_stages_of(ns) == ns; }
static inline int wx_lat_stages(int F) { return wx_lat_stages_of(F / 2); }
#define WX_WAVE 64

// status codes of the C ABI (include/waveletsext_hip.h)
#define WX_OK 0
#define WX_EASSERT (-1)       // the reference would fail an @assert (AssertionError)
#define WX_EARG (-2)          // the reference would throw ArgumentError
#define WX_EBOUNDS (-3)       // the reference would throw BoundsError
#define WX_EHIP (-10)         // HIP runtime failure (see wx_last_error)
#define WX_EUNSUPPORTED (-11) // valid for the reference, not implemented by this library

// QMF passed by value as a kernel argument: lands in SGPRs via s_load, so every tap is a
// scalar operand of v_fma_f64 / v_fma_f32.
struct WxFilt {
    double q[WX_MAXF];
    int F;
};

// autocorrelation-shell half filter a_l/(2*sqrt(2)), l = 1..F-1 (acwt_utils.jl:7-48)
struct WxAcFilt {
    double b[WX_MAXF];        // b[l-1] = c2 * a_l
    double c1;                // 1/sqrt(2)
    int F;                    // length of the generating QMF; AC filter has 2F-1 taps
};

struct WxStreamScratch;       // opaque

template <typename T> struct WxVec2;
template <> struct WxVec2<double> { typedef double2 type; };
template <> struct WxVec2<float> { typedef float2 type; };

// Wavelets.Threshold rules (HardTH 0, SoftTH 1, SemiSoftTH 2, SteinTH 3), see wx_denoise.hip
template <typename T> __device__ __forceinline__ T wx_thresh(T v, T tt, int th_kind)
{
    if (th_kind == 0) return (T)fabs((double)v) <= tt ? (T)0 : v;
    const T sg = v > (T)0 ? (T)1 : (v < (T)0 ? (T)-1 : v);
    if (th_kind == 1) { const T sh = (T)((T)fabs((double)v) - tt); return sh < (T)0 ? (T)0 : (T)(sg * sh); }
    if (th_kind == 2) {
        // semisoft (Gao-Bruce, upper knee at 2t): 0 below t, sign(x) * 2(|x| - t) up to 2t, x above
        const T av = (T)fabs((double)v);
        if (av > (T)((T)2 * tt)) return v;
        const T tmp = (T)((T)((T)2 * av) - (T)((T)2 * tt));
        return tmp < (T)0 ? (T)0 : (T)(sg * tmp);
    }
    const T sh = (T)((T)1 - (T)((T)(tt * tt) / (T)(v * v)));
    return sh < (T)0 ? (T)0 : (T)(v * sh);
}


// optional threshold applied by an inverse transform while it loads the coefficients (denoise: threshold + inverse in one
// pass over the table): rows [lo, n) of every signal, t[0] or t[signal]
struct WxThreshArg {
    const void *t;
    int kind, lo, per_signal;
    double scale;               // threshold = t[...] * scale (sigma_i * dnt.t with sigma left on the device)
    const void *head = nullptr; // idwt of a pyramid: samples 0 .. 63 of signal s come from head + 64 s (wx_dwttail.hip)
};

static __device__ __forceinline__ int wx_modn(int x, int n)
{
    int r = x % n;
    return r < 0 ? r + n : r;
}
static __device__ __forceinline__ int64_t wx_modn64(int64_t x, int64_t n)
{
    int64_t r = x % n;
    return r < 0 ? r + n : r;
}

// global -> LDS (or register-staged global -> global) copy of n elements by the whole workgroup with NB loads of a
// lane in flight: a plain `dst[i] = src[i]` loop waits for every load before issuing the next (measured: tree
// selection 1.0 -> 0.5 ms, 2-D tile levels 3.0 -> 2.3 ms from this alone)
template <typename T, int NB = 8>
static __device__ __forceinline__ void wx_stage(T *__restrict__ dst, const T *__restrict__ src, int n)
{
    const int st = blockDim.x;
    int i = threadIdx.x;
    for (; i + (NB - 1) * st < n; i += NB * st) {
        T v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) v[u] = src[i + u * st];
#pragma unroll
        for (int u = 0; u < NB; ++u) dst[i + u * st] = v[u];
    }
    for (; i < n; i += st) dst[i] = src[i];
}

// per-lane partial of sum x^2 over x[tid], x[tid + 256], ...: eight loads in flight, same order of additions
template <typename T>
static __device__ __forceinline__ double wx_sumsq_strided(const T *__restrict__ x, int64_t cnt)
{
    double acc = 0.0;
    int64_t i = threadIdx.x;
    for (; i + 7 * 256 < cnt; i += 8 * 256) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const double d = (double)v[u]; acc = fma(d, d, acc); }
    }
    for (; i < cnt; i += 256) { const double d = (double)x[i]; acc = fma(d, d, acc); }
    return acc;
}

#define WX_HIP_CHECK(expr)                                   \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return wx_set_hip_error(_e, #expr, __FILE__, __LINE__); \
    } while (0)

int wx_set_hip_error(hipError_t e, const char *what, const char *file, int line);
int wx_set_error(int code, const char *msg);
