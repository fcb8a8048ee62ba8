// wx_dwt3d.hip -- 3-D discrete wavelet transform of a batch of cubes (the 3-D case).
//
// Reference: dwtall / idwtall (src/mod/dwt/dwt_all.jl:39-54, 95-110) call Wavelets.jl's dwt! / idwt! on every slice of
// the last dimension; "dwt is currently available for 1-D, 2-D, and 3-D signals" (dwt_all.jl:8-9).  Wavelets.jl is not
// vendored in the reference tree; its 3-D filter transform is the separable pyramid restated here from its published
// source: the array is a cube with dyadic sides, and level l applies the one-level analysis step (the same step as 1-D,
// dwt/dwt_one_level.jl:79-107: [approximation | detail] halves) along dimension 1, 2 and 3 of the low-pass sub-cube of
// side n >> l; the inverse undoes the levels from the coarsest, dimensions in reverse.  The three passes of a level act
// on different axes and commute exactly in exact arithmetic; in floating point their order only moves the last bits
// (the tolerance of the path is 1e-10).  Parity unpinned (no Wavelets.jl source, no Julia here): pinned by the oracle's
// restatement out of the same 1-D step, by perfect reconstruction and by the energy identity of the orthonormal transform.
//
// One thread per output pair of one line; the fastest thread index is the dimension-1 coordinate (or the pair index when
// the line runs along dimension 1), so every pass reads and writes whole cache lines.  A level is three passes through one
// scratch cube plus the copy of the sub-cube back (HBM-bound, 8 sub-cube traversals per level: the row is breadth, the
// batched 1-D / 2-D kernels are where the bytes of this library go).
#include "../../include/waveletsext_hip.h"     // the definitions below must match the public prototypes
#include "wx_common.h"
#include "wx_kernels.h"
#include "wx_host.h"

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

extern "C" int wx_device_count(void);

namespace {

// lines of length ns along `axis` inside the sub-cube [0, ns)^3 of every n^3 cube
template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void k_dwt3d_axis(const T *__restrict__ src, T *__restrict__ dst, int n, int ns, int axis,
                                                    int64_t batch, WxFilt filt)
{
    const int h = ns >> 1;
    const int64_t per = (int64_t)h * ns * ns, total = per * batch;
    const int64_t n2 = (int64_t)n * n, n3 = n2 * n;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = g / per;
        int64_t li = g - b * per;
        int t;
        int64_t base, es;
        if (axis == 0) {
            t = (int)(li % h); li /= h;
            const int i2 = (int)(li % ns), i3 = (int)(li / ns);
            base = (int64_t)i2 * n + (int64_t)i3 * n2; es = 1;
        } else if (axis == 1) {
            const int i1 = (int)(li % ns); li /= ns;
            t = (int)(li % h);
            const int i3 = (int)(li / h);
            base = i1 + (int64_t)i3 * n2; es = n;
        } else {
            const int i1 = (int)(li % ns); li /= ns;
            const int i2 = (int)(li % ns);
            t = (int)(li / ns);
            base = i1 + (int64_t)i2 * n; es = n2;
        }
        const T *v = src + b * n3 + base;
        T *o = dst + b * n3 + base;
        if (!INVERSE) {
            double a = 0.0, dd = 0.0;
            int k1 = 2 * t, k2 = 2 * t + 1;
            if (k1 >= ns) k1 -= ns;
            if (k2 >= ns) k2 -= ns;
            for (int k = 0; k < filt.F; ++k) {
                a = fma(filt.q[k], (double)v[k1 * es], a);
                dd = fma((k & 1) ? -filt.q[k] : filt.q[k], (double)v[k2 * es], dd);
                k1 = k1 + 1 == ns ? 0 : k1 + 1;
                k2 = k2 == 0 ? ns - 1 : k2 - 1;
            }
            o[t * es] = (T)a;
            o[(h + t) * es] = (T)dd;
        } else {
            double v0 = 0.0, v1 = 0.0;
            int k1 = t, k2 = t;
            for (int mm = 0; mm < filt.F / 2; ++mm) {
                const double av = (double)v[k1 * es], dv = (double)v[(h + k2) * es];
                v0 = fma(filt.q[2 * mm], av, v0);
                v0 = fma(-filt.q[2 * mm + 1], dv, v0);
                v1 = fma(filt.q[2 * mm + 1], av, v1);
                v1 = fma(filt.q[2 * mm], dv, v1);
                k1 = k1 == 0 ? h - 1 : k1 - 1;
                k2 = k2 + 1 == h ? 0 : k2 + 1;
            }
            o[(2 * t) * es] = (T)v0;
            o[(2 * t + 1) * es] = (T)v1;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_copy_subcube(const T *__restrict__ src, T *__restrict__ dst, int n, int ns, int64_t batch)
{
    const int64_t per = (int64_t)ns * ns * ns, total = per * batch;
    const int64_t n2 = (int64_t)n * n, n3 = n2 * n;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = g / per;
        int64_t li = g - b * per;
        const int i1 = (int)(li % ns); li /= ns;
        const int i2 = (int)(li % ns);
        const int i3 = (int)(li / ns);
        const int64_t e = b * n3 + i1 + (int64_t)i2 * n + (int64_t)i3 * n2;
        dst[e] = src[e];
    }
}

unsigned grid_for(int64_t total)
{
    int64_t g = (total + 255) / 256;
    if (g > 256 * 64) g = 256 * 64;
    if (g < 1) g = 1;
    return (unsigned)g;
}

template <typename T>
int api_dwt3d(const T *x, T *y, int64_t n1, int64_t n2, int64_t n3, int L, int64_t batch, const double *qmf, int F, bool inverse,
              void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(n1 >= 1 && batch >= 0, WX_EARG, "dwt 3-D: bad dimensions");
    WX_REQUIRE(n1 == n2 && n2 == n3, WX_EASSERT, "3-D dwt: the array must be a cube (Wavelets.jl 3-D transform)");
    WX_REQUIRE(wx_isdyadic(n1) && 0 <= L && L <= wx_maxtransformlevels(n1), WX_EASSERT,
               "3-D dwt: dyadic sides and 0 <= L <= maxtransformlevels(x)");
    // (sides beyond 1024 were refused until round 4; the kernels index with 64 bits, what bounds the side is the memory of the
    // device: a 2048^3 Float32 cube is 32 GiB, plus as much scratch)
    WX_REQUIRE(n1 < ((int64_t)1 << 20), WX_EUNSUPPORTED, "3-D dwt: side of 2^20 or more");
    if (wx_device_count() < 1) return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    const int64_t cube = n1 * n1 * n1;
    if (batch == 0) return WX_OK;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dx = (const T *)io.in(x, sizeof(T) * cube * batch);
    T *dy = (T *)io.out(y, sizeof(T) * cube * batch);
    if (!dx || !dy) return io.finish(WX_EHIP);
    // The level on the whole cube moves all the data: out of place it reads x itself and its three passes end in y (x -> y -> tmp -> y) -- no
    // copy of the cube before, none after (round 5: the forward transform took a copy + three passes + a copy: 3.3 ms per GiB, 8 % of the roofline)
    const bool direct = dy != dx && L >= 1 && (!inverse || L == 1);
    if (dy != dx && !direct) WX_HIP_CHECK(hipMemcpyAsync(dy, dx, sizeof(T) * cube * batch, hipMemcpyDeviceToDevice, st));
    if (L == 0) return io.finish(WX_OK);                 // (direct implies L >= 1: the copy above has been made)
    T *tmp = (T *)scr.alloc(sizeof(T) * cube * batch);
    if (!tmp) return io.finish(WX_EHIP);
    const int n = (int)n1;
    for (int l = 0; l < L; ++l) {
        const int ns = inverse ? n >> (L - 1 - l) : n >> l;
        const int64_t pairs = (int64_t)(ns >> 1) * ns * ns * batch;
        const unsigned g = grid_for(pairs);
        if (direct && ns == n) {
            if (!inverse) {
                hipLaunchKernelGGL((k_dwt3d_axis<T, false>), dim3(g), dim3(256), 0, st, dx, dy, n, ns, 0, batch, filt);
                hipLaunchKernelGGL((k_dwt3d_axis<T, false>), dim3(g), dim3(256), 0, st, dy, tmp, n, ns, 1, batch, filt);
                hipLaunchKernelGGL((k_dwt3d_axis<T, false>), dim3(g), dim3(256), 0, st, tmp, dy, n, ns, 2, batch, filt);
            } else {
                hipLaunchKernelGGL((k_dwt3d_axis<T, true>), dim3(g), dim3(256), 0, st, dx, dy, n, ns, 2, batch, filt);
                hipLaunchKernelGGL((k_dwt3d_axis<T, true>), dim3(g), dim3(256), 0, st, dy, tmp, n, ns, 1, batch, filt);
                hipLaunchKernelGGL((k_dwt3d_axis<T, true>), dim3(g), dim3(256), 0, st, tmp, dy, n, ns, 0, batch, filt);
            }
            WX_HIP_CHECK(hipGetLastError());
            continue;
        }
        if (!inverse) {
            hipLaunchKernelGGL((k_dwt3d_axis<T, false>), dim3(g), dim3(256), 0, st, dy, tmp, n, ns, 0, batch, filt);
            hipLaunchKernelGGL((k_dwt3d_axis<T, false>), dim3(g), dim3(256), 0, st, tmp, dy, n, ns, 1, batch, filt);
            hipLaunchKernelGGL((k_dwt3d_axis<T, false>), dim3(g), dim3(256), 0, st, dy, tmp, n, ns, 2, batch, filt);
        } else {
            hipLaunchKernelGGL((k_dwt3d_axis<T, true>), dim3(g), dim3(256), 0, st, dy, tmp, n, ns, 2, batch, filt);
            hipLaunchKernelGGL((k_dwt3d_axis<T, true>), dim3(g), dim3(256), 0, st, tmp, dy, n, ns, 1, batch, filt);
            hipLaunchKernelGGL((k_dwt3d_axis<T, true>), dim3(g), dim3(256), 0, st, dy, tmp, n, ns, 0, batch, filt);
        }
        hipLaunchKernelGGL(k_copy_subcube<T>, dim3(grid_for(2 * pairs)), dim3(256), 0, st, tmp, dy, n, ns, batch);
        WX_HIP_CHECK(hipGetLastError());
    }
    return io.finish(WX_OK);
}

}  // namespace

extern "C" {
int wx_dwt3d_f64(const double *x, double *y, int64_t n1, int64_t n2, int64_t n3, int L, int64_t batch, const double *qmf, int F,
                 void *stream)
{ return api_dwt3d<double>(x, y, n1, n2, n3, L, batch, qmf, F, false, stream); }
int wx_dwt3d_f32(const float *x, float *y, int64_t n1, int64_t n2, int64_t n3, int L, int64_t batch, const double *qmf, int F,
                 void *stream)
{ return api_dwt3d<float>(x, y, n1, n2, n3, L, batch, qmf, F, false, stream); }
int wx_idwt3d_f64(const double *x, double *y, int64_t n1, int64_t n2, int64_t n3, int L, int64_t batch, const double *qmf, int F,
                  void *stream)
{ return api_dwt3d<double>(x, y, n1, n2, n3, L, batch, qmf, F, true, stream); }
int wx_idwt3d_f32(const float *x, float *y, int64_t n1, int64_t n2, int64_t n3, int L, int64_t batch, const double *qmf, int F,
                  void *stream)
{ return api_dwt3d<float>(x, y, n1, n2, n3, L, batch, qmf, F, true, stream); }
}
