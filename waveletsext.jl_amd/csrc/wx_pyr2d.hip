// wx_pyr2d.hip -- dwt / idwt (the pyramid) of SMALL images, a whole image per workgroup in LDS.
//   2-D dwt_step! / idwt_step!   dwt/dwt_one_level.jl:319-354, 401-436 (columns then rows; the inverse rows then columns)
//   dwt / idwt / dwtall / idwtall of images: Wavelets.jl's 2-D pyramid as called by dwt/dwt_all.jl:39-110, i.e. wpt / iwpt along
//   maketree(m, n, L, :dwt) -- level d decomposes the top-left (m >> d) x (n >> d) block only.
// The tile kernels of wx_dwt2d.hip make one pass over the batch per level: a pyramid of 64 x 64 images read and wrote every image
// L times and ran at 0.15 of the HBM peak.  A pyramid's levels shrink by four, so all of them together cost 4/3 of the first one:
// here an image is loaded once, every level runs between two LDS arrays (lanes down the rows in the row pass, across the columns of odd
// pitch in the column pass: conflict-free), and the image is stored once.  Images of up to 2 x 64 KiB (128 x 128 Float32, 64 x 128 Float64).
#include "wx_common.h"
#include "wx_kernels.h"
#include "wx_host.h"

namespace {

// FT: filter length at compile time (0 = any even length up to WX_MAXF, taps in a loop)
template <typename T, int FT, bool INVERSE>
__global__ __launch_bounds__(1024) void k_pyr2d_small(const T *__restrict__ x, T *__restrict__ y, int lm, int ln, int L, int64_t batch, WxFilt filt)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem_p[];
    const int m = 1 << lm, n = 1 << ln, P = m + 1;                 // odd pitch: the row pass walks columns
    T *A = reinterpret_cast<T *>(wx_smem_p), *B = A + (size_t)n * P;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int F = FT ? FT : filt.F;
    T q[FT ? FT : WX_MAXF];
    if (FT) {
#pragma unroll
        for (int k = 0; k < (FT ? FT : 1); ++k) q[k] = (T)filt.q[k];
    } else {
        for (int k = 0; k < F; ++k) q[k] = (T)filt.q[k];
    }
    const int64_t mn = (int64_t)m * n;
    for (int64_t img = blockIdx.x; img < batch; img += gridDim.x) {
        const T *xs = x + img * mn;
        T *ys = y + img * mn;
        for (int e = tid; e < (int)mn; e += NT) A[(e >> lm) * P + (e & (m - 1))] = xs[e];
        __syncthreads();
        for (int s = 0; s < L; ++s) {
            const int d = INVERSE ? L - 1 - s : s;
            const int lmp = lm - d, lnp = ln - d, mp = 1 << lmp, np = 1 << lnp, h1 = mp >> 1, h2 = np >> 1;
            if (!INVERSE) {
                // columns (dim 1): item = (column c < np, pair i < h1), lanes ACROSS the columns (odd pitch: conflict-free; along i the
                // reads are two elements apart)
                for (int e = tid; e < np * h1; e += NT) {
                    const int c = e & (np - 1), i = e >> lnp;
                    const T *v = A + c * P;
                    // one window for two outputs: a[i] reads v[2i .. 2i+F-1], and so does d[i + F/2 - 1] (its taps run backwards)
                    const int id = (i + F / 2 - 1) & (h1 - 1);
                    T a = 0, dd = 0;
                    if (FT) {
                        T w[FT ? FT : 1];
#pragma unroll
                        for (int k = 0; k < (FT ? FT : 1); ++k) w[k] = v[(2 * i + k) & (mp - 1)];
#pragma unroll
                        for (int k = 0; k < (FT ? FT : 1); ++k) {
                            a = fma(q[k], w[k], a);
                            dd = fma((k & 1) ? -q[k] : q[k], w[(FT ? FT : 1) - 1 - k], dd);
                        }
                    } else {
                        for (int k = 0; k < F; ++k) {
                            a = fma(q[k], v[(2 * i + k) & (mp - 1)], a);
                            dd = fma((k & 1) ? -q[k] : q[k], v[(2 * i + F - 1 - k) & (mp - 1)], dd);
                        }
                    }
                    B[c * P + i] = a;
                    B[c * P + h1 + id] = dd;
                }
                __syncthreads();
                // rows (dim 2): item = (row r < mp, pair j < h2), lanes along r
                for (int e = tid; e < mp * h2; e += NT) {
                    const int j = e >> lmp, r = e & (mp - 1);
                    const T *v = B + r;
                    const int jd = (j + F / 2 - 1) & (h2 - 1);
                    T a = 0, dd = 0;
                    if (FT) {
                        T w[FT ? FT : 1];
#pragma unroll
                        for (int k = 0; k < (FT ? FT : 1); ++k) w[k] = v[((2 * j + k) & (np - 1)) * P];
#pragma unroll
                        for (int k = 0; k < (FT ? FT : 1); ++k) {
                            a = fma(q[k], w[k], a);
                            dd = fma((k & 1) ? -q[k] : q[k], w[(FT ? FT : 1) - 1 - k], dd);
                        }
                    } else {
                        for (int k = 0; k < F; ++k) {
                            a = fma(q[k], v[((2 * j + k) & (np - 1)) * P], a);
                            dd = fma((k & 1) ? -q[k] : q[k], v[((2 * j + F - 1 - k) & (np - 1)) * P], dd);
                        }
                    }
                    A[j * P + r] = a;
                    A[(h2 + jd) * P + r] = dd;
                }
                __syncthreads();
            } else {
                // rows first (dim 2): x[2j] = sum q[2t] a[j-t] - q[2t+1] d[j+t], x[2j+1] = sum q[2t+1] a[j-t] + q[2t] d[j+t]
                for (int e = tid; e < mp * h2; e += NT) {
                    const int j = e >> lmp, r = e & (mp - 1);
                    const T *v = A + r;
                    T x0 = 0, x1 = 0;
#pragma unroll
                    for (int t = 0; t < (FT ? FT / 2 : 1); ++t) {
                        if (FT) {
                            const T av = v[((j - t) & (h2 - 1)) * P], dv = v[(h2 + ((j + t) & (h2 - 1))) * P];
                            x0 = fma(q[2 * t], av, x0); x0 = fma(-q[2 * t + 1], dv, x0);
                            x1 = fma(q[2 * t + 1], av, x1); x1 = fma(q[2 * t], dv, x1);
                        }
                    }
                    if (!FT)
                        for (int t = 0; t < F / 2; ++t) {
                            const T av = v[((j - t) & (h2 - 1)) * P], dv = v[(h2 + ((j + t) & (h2 - 1))) * P];
                            x0 = fma(q[2 * t], av, x0); x0 = fma(-q[2 * t + 1], dv, x0);
                            x1 = fma(q[2 * t + 1], av, x1); x1 = fma(q[2 * t], dv, x1);
                        }
                    B[(2 * j) * P + r] = x0;
                    B[(2 * j + 1) * P + r] = x1;
                }
                __syncthreads();
                // then columns (dim 1), lanes across the columns
                for (int e = tid; e < np * h1; e += NT) {
                    const int c = e & (np - 1), i = e >> lnp;
                    const T *v = B + c * P;
                    T x0 = 0, x1 = 0;
#pragma unroll
                    for (int t = 0; t < (FT ? FT / 2 : 1); ++t) {
                        if (FT) {
                            const T av = v[(i - t) & (h1 - 1)], dv = v[h1 + ((i + t) & (h1 - 1))];
                            x0 = fma(q[2 * t], av, x0); x0 = fma(-q[2 * t + 1], dv, x0);
                            x1 = fma(q[2 * t + 1], av, x1); x1 = fma(q[2 * t], dv, x1);
                        }
                    }
                    if (!FT)
                        for (int t = 0; t < F / 2; ++t) {
                            const T av = v[(i - t) & (h1 - 1)], dv = v[h1 + ((i + t) & (h1 - 1))];
                            x0 = fma(q[2 * t], av, x0); x0 = fma(-q[2 * t + 1], dv, x0);
                            x1 = fma(q[2 * t + 1], av, x1); x1 = fma(q[2 * t], dv, x1);
                        }
                    A[c * P + 2 * i] = x0;
                    A[c * P + 2 * i + 1] = x1;
                }
                __syncthreads();
            }
        }
        for (int e = tid; e < (int)mn; e += NT) ys[e] = A[(e >> lm) * P + (e & (m - 1))];
        __syncthreads();
    }
}

template <typename T, int FT>
int launch(bool inverse, const T *x, T *y, int lm, int ln, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    const int m = 1 << lm, n = 1 << ln;
    const size_t lds = (size_t)2 * n * (m + 1) * sizeof(T);
    auto kf = k_pyr2d_small<T, FT, false>;
    auto ki = k_pyr2d_small<T, FT, true>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(inverse ? ki : kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return wx_set_error(WX_EHIP, "small-image pyramid: LDS attribute");
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    int nt = m * n / 8;                                            // eight items of the first level per lane
    if (nt < 256) nt = 256;
    if (nt > 1024) nt = 1024;
    while (nt * per_cu > 2048) { if (nt > 256) nt >>= 1; else --per_cu; }
    int64_t grid = (int64_t)256 * per_cu;
    if (grid > batch) grid = batch;
    if (inverse) hipLaunchKernelGGL(ki, dim3((unsigned)grid), dim3(nt), lds, st, x, y, lm, ln, L, batch, filt);
    else hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(nt), lds, st, x, y, lm, ln, L, batch, filt);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "small-image pyramid launch", __FILE__, __LINE__);
    return WX_OK;
}

}  // namespace

// images whose two LDS copies fit 128 KiB; dyadic sides of at least 4
template <typename T> bool wx_pyr2d_small_ok(int64_t m, int64_t n, int L, int F)
{
    static const bool off = wx_getenv("WX_PYR2D_SMALL") && atoi(wx_getenv("WX_PYR2D_SMALL")) == 0;
    static const int64_t maxb = wx_getenv("WX_PYR2D_SMALL_MAXKB") ? atoll(wx_getenv("WX_PYR2D_SMALL_MAXKB")) * 1024 : 128 * 1024;
    if (off || m < 4 || n < 4 || (m & (m - 1)) || (n & (n - 1)) || L < 1 || F < 2 || (F & 1) || F > WX_MAXF) return false;
    if (((m < n ? m : n) >> L) < 1) return false;
    return (int64_t)2 * n * (m + 1) * (int64_t)sizeof(T) <= maxb;
}
template bool wx_pyr2d_small_ok<double>(int64_t, int64_t, int, int);
template bool wx_pyr2d_small_ok<float>(int64_t, int64_t, int, int);

template <typename T>
int wx_dev_pyr2d_small(bool inverse, const T *x, T *y, int64_t m, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    if (batch == 0) return WX_OK;
    int lm = 0, ln = 0;
    while (((int64_t)1 << lm) < m) ++lm;
    while (((int64_t)1 << ln) < n) ++ln;
    switch (filt.F) {
    case 2: return launch<T, 2>(inverse, x, y, lm, ln, L, batch, filt, st);
    case 4: return launch<T, 4>(inverse, x, y, lm, ln, L, batch, filt, st);
    case 6: return launch<T, 6>(inverse, x, y, lm, ln, L, batch, filt, st);
    case 8: return launch<T, 8>(inverse, x, y, lm, ln, L, batch, filt, st);
    default: return launch<T, 0>(inverse, x, y, lm, ln, L, batch, filt, st);
    }
}
template int wx_dev_pyr2d_small<double>(bool, const double *, double *, int64_t, int64_t, int, int64_t, const WxFilt &, hipStream_t);
template int wx_dev_pyr2d_small<float>(bool, const float *, float *, int64_t, int64_t, int, int64_t, const WxFilt &, hipStream_t);
