// wx_swt2d.hip -- batched 2-D redundant (undecimated) transforms for gfx950: stationary family
// (sdwt / swpt / swpd 2-D and inverses) and autocorrelation family (acdwt / acwpt / acwpd 2-D).
//
// Reference semantics (paths relative to /root/reference/src/mod):
//   sdwt_step! 2-D   swt/swt_one_level.jl:334-370   1-D step down every column into temp[:,:,1|2], then
//                    along every row: w1,w2 <- temp1 (low,high along dim 2), w3,w4 <- temp2
//   isdwt_step! 2-D  swt/swt_one_level.jl:395-431 (average), :433-469 (shift): rows first, then columns
//   acdwt_step! 2-D  acwt/acwt_one_level.jl:240-276;  iacdwt_step! 2-D :288-322
//   containers       SWT.jl:132-158 (sdwt (n,m,3L+1)), :474-513 (swpt (n,m,4^L)), :870-902 (swpd, quad heap),
//                    inverses :286-358, :648-758, :1095-1199; ACWT.jl:131-157, 306-329, 462-501, 612-648,
//                    761-793, 970-1000
// Images are (m rows, n cols) column-major; every pass has consecutive lanes on consecutive rows
// (coalesced).  A forward level is one pass along dim 1 into scratch and one along dim 2; the
// parent slice is fully consumed by the first pass, which resolves the reference's aliasing of
// the parent with child 1.
#include "wx_common.h"
#include "wx_kernels.h"
#include <cstdlib>

enum { WX2_DWT = 0, WX2_WPT = 1, WX2_WPD = 2 };

struct WxRed2d {
    int layout, L, d;
    int m, n;               // rows, columns
    int64_t ncols;          // slices per image in the coefficient array
    int64_t batch;
};

static __device__ __forceinline__ int64_t wx_quad_start(int d)   // 1-based heap index of the first node of depth d
{
    int64_t s = 1;
    for (int t = 0; t < d; ++t) s = 4 * s - 2;
    return s;
}

// 0-based slice indices of the parent (pv) and of the four children of node b at depth d
static __device__ __forceinline__ void wx_red2d_slices(const WxRed2d &D, int b, int64_t &pv, int64_t (&pc)[4])
{
    if (D.layout == WX2_DWT) {
        pv = 3 * (D.L - D.d);
        pc[0] = pv - 3; pc[1] = pv - 2; pc[2] = pv - 1; pc[3] = pv;
    } else if (D.layout == WX2_WPT) {
        const int64_t nc = (int64_t)1 << (2 * (D.L - D.d - 1));
        pv = 4 * (int64_t)b * nc;
        for (int c = 0; c < 4; ++c) pc[c] = (4 * (int64_t)b + c) * nc;
    } else {
        const int64_t i = wx_quad_start(D.d) + b;
        pv = i - 1;
        for (int c = 0; c < 4; ++c) pc[c] = 4 * i - 2 + c - 1;
    }
}

// undecimated analysis pair at index i along a line of `len` samples with element stride `es`
template <typename T, bool AC>
static __device__ __forceinline__ void wx_red_point(const T *line, int64_t es, int len, int i, int s,
                                                    const WxFilt &filt, const WxAcFilt &ac, T &lo, T &hi)
{
    if (!AC) {
        double a = 0.0, dd = 0.0;
        int k1 = i - s; if (k1 < 0) k1 += len;
        int k2 = i;
        // taps in blocks of four: eight loads in flight before the first multiply-add; same order of the sums
        for (int j0 = 0; j0 < filt.F; j0 += 4) {
            T xa[4], xd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j0 + u < filt.F) {
                    xa[u] = line[k1 * es]; xd[u] = line[k2 * es];
                    k1 += s; if (k1 >= len) k1 -= len;
                    k2 -= s; if (k2 < 0) k2 += len;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j0 + u < filt.F) {
                    const int j = j0 + u;
                    a = fma(filt.q[j], (double)xa[u], a);
                    dd = fma((j & 1) ? -filt.q[j] : filt.q[j], (double)xd[u], dd);
                }
            }
        }
        lo = (T)a; hi = (T)dd;
    } else {
        double S = 0.0;
        const int s2 = (2 * s) % len;
        int km = i - s; if (km < 0) km += len;
        int kp = i + s; if (kp >= len) kp -= len;
        for (int l0 = 1; l0 < ac.F; l0 += 8) {                 // four odd lags per block: eight loads in flight
            T xm[4], xp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (l0 + 2 * u < ac.F) {
                    xm[u] = line[km * es]; xp[u] = line[kp * es];
                    km -= s2; if (km < 0) km += len;
                    kp += s2; if (kp >= len) kp -= len;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (l0 + 2 * u < ac.F) S = fma(ac.b[l0 + 2 * u - 1], (double)xm[u] + (double)xp[u], S);
        }
        const double c = ac.c1 * (double)line[i * es];
        lo = (T)(c + S); hi = (T)(c - S);
    }
}

// pass 1 (dim 1): parent slice -> scratch t[job][0|1]
template <typename T, bool AC>
__global__ __launch_bounds__(256) void k_red2d_fwd_dim1(const T *__restrict__ x, T *__restrict__ xw, T *__restrict__ tmp,
                                                        WxRed2d D, int nodes, WxFilt filt, WxAcFilt ac)
{
    // grid.y walks the jobs (node x signal), grid.x the elements of one image: the per-element index arithmetic is
    // one 32-bit division; with a flat 64-bit index the four 64-bit divisions per element cost more than the filter
    const int64_t mn = (int64_t)D.m * D.n;
    const int64_t njobs = D.batch * nodes;
    const int s = (1 << D.d) % D.m;
    for (int64_t job = blockIdx.y; job < njobs; job += gridDim.y) {
      const int b = (int)(job % nodes);
      const int64_t sig = job / nodes;
      int64_t pv, pc[4];
      wx_red2d_slices(D, b, pv, pc);
      const T *src = (D.d == 0) ? x + sig * mn : xw + (sig * D.ncols + pv) * mn;
      for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < (unsigned)mn; e += gridDim.x * 256u) {
        const unsigned c = e / (unsigned)D.m;
        const int r = (int)(e - c * (unsigned)D.m);
        if (D.d == 0 && D.layout == WX2_WPD) xw[(sig * D.ncols) * mn + e] = src[e];
        T lo, hi;
        wx_red_point<T, AC>(src + (int64_t)c * D.m, 1, D.m, r, s, filt, ac, lo, hi);
        tmp[(job * 2) * mn + e] = lo;
        tmp[(job * 2 + 1) * mn + e] = hi;
      }
    }
}

// pass 2 (dim 2): t[job][which] -> children (2*which, 2*which+1)
template <typename T, bool AC>
__global__ __launch_bounds__(256) void k_red2d_fwd_dim2(T *__restrict__ xw, const T *__restrict__ tmp, WxRed2d D,
                                                        int nodes, WxFilt filt, WxAcFilt ac)
{
    const int64_t mn = (int64_t)D.m * D.n;
    const int64_t njobs = D.batch * nodes * 2;
    const int s = (1 << D.d) % D.n;
    for (int64_t jw = blockIdx.y; jw < njobs; jw += gridDim.y) {       // job*2 + which
      const int which = (int)(jw & 1);
      const int64_t job = jw >> 1;
      const int b = (int)(job % nodes);
      const int64_t sig = job / nodes;
      int64_t pv, pc[4];
      wx_red2d_slices(D, b, pv, pc);
      for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < (unsigned)mn; e += gridDim.x * 256u) {
        const int c = (int)(e / (unsigned)D.m);
        const int r = (int)(e - (unsigned)c * (unsigned)D.m);
        T lo, hi;
        wx_red_point<T, AC>(tmp + jw * mn + r, D.m, D.n, c, s, filt, ac, lo, hi);
        xw[(sig * D.ncols + pc[2 * which]) * mn + e] = lo;
        xw[(sig * D.ncols + pc[2 * which + 1]) * mn + e] = hi;
      }
    }
}

// ------------------------------------------------------------------------------------------
// inverse
// ------------------------------------------------------------------------------------------
struct WxInv2d {
    WxRed2d R;
    const void *in;          // caller's coefficients (read only)
    void *cur;               // computed nodes of depth d+1 (slice = node index in level)
    void *out;               // destination for depth d (or final x)
    int64_t cur_cols, out_cols;
    const uint8_t *tree;     // WPD only
    int64_t ntree;
};

// base pointer of child c (0..3) of node b; DWT: child 0 is the running reconstruction
template <typename T>
static __device__ __forceinline__ const T *wx_inv2d_child(const WxInv2d &D, int64_t sig, int b, int c)
{
    const int64_t mn = (int64_t)D.R.m * D.R.n;
    const T *in = reinterpret_cast<const T *>(D.in) + sig * D.R.ncols * mn;
    const T *cur = reinterpret_cast<const T *>(D.cur) + sig * D.cur_cols * mn;
    if (D.R.layout == WX2_DWT) {
        if (c == 0) return (D.R.d == D.R.L - 1) ? in : cur;
        return in + (int64_t)(3 * (D.R.L - D.R.d) - 3 + c) * mn;
    }
    const int64_t cb = 4 * (int64_t)b + c;
    if (D.R.layout == WX2_WPT) return (D.R.d == D.R.L - 1) ? in + cb * mn : cur + cb * mn;
    const int64_t heap = wx_quad_start(D.R.d + 1) + cb;
    const bool computed = D.tree ? (heap <= D.ntree && D.tree[heap - 1]) : (D.R.d + 1 < D.R.L);
    return computed ? cur + cb * mn : in + (heap - 1) * mn;
}

// one parent sample of isdwt_step! along a line: slot u of residue class cls (position cls + u*s)
template <typename T>
static __device__ __forceinline__ double wx_isdwt_point(const T *w1, const T *w2, int64_t es, int len, int d, int cls,
                                                        int u, bool shifted, const WxFilt &filt)
{
    const int s = 1 << d;
    const int np = len >> d, nc = np >> 1;
    const int t0 = shifted ? u : (u + 1 == np ? 0 : u + 1);
    const int swc = shifted ? cls + s : cls;
    const int tau = t0 >> 1;
    const bool odd = t0 & 1;
    int k1 = tau, k2 = tau;
    double v = 0.0;
    // tap pairs in blocks of four: eight loads in flight before the first multiply-add, same order of the sum
    for (int m0 = 0; m0 < filt.F / 2; m0 += 4) {
        T xa[4], xc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (m0 + q < filt.F / 2) {
                xa[q] = w1[(swc + (int64_t)k1 * 2 * s) * es];
                xc[q] = w2[(swc + (int64_t)k2 * 2 * s) * es];
                k1 = k1 == 0 ? nc - 1 : k1 - 1;
                k2 = k2 + 1 == nc ? 0 : k2 + 1;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (m0 + q < filt.F / 2) {
                const int m = m0 + q;
                const double a = (double)xa[q], c = (double)xc[q];
                if (!odd) { v = fma(filt.q[2 * m], a, v); v = fma(-filt.q[2 * m + 1], c, v); }
                else { v = fma(filt.q[2 * m + 1], a, v); v = fma(filt.q[2 * m], c, v); }
            }
        }
    }
    return v;
}

// rows pass (dim 2): children (2*which, 2*which+1) -> tmp[job][which]
//   shift mode: only rows of class sw (mod 2s) and columns of class sv (mod s) are produced
template <typename T>
__global__ __launch_bounds__(256) void k_red2d_inv_dim2(WxInv2d D, T *__restrict__ tmp, int nodes, int sm_mode,
                                                        int sv, int sw, WxFilt filt)
{
    const int m = D.R.m, n = D.R.n, d = D.R.d, s = 1 << d;
    const int64_t mn = (int64_t)m * n;
    const int nr = sm_mode ? (m >> (d + 1)) : m;          // rows handled per job
    const int ncl = sm_mode ? (n >> d) : n;               // columns handled per job
    const int64_t per = (int64_t)nr * ncl;
    const int64_t njobs = D.R.batch * nodes * 2;
    for (int64_t jw = blockIdx.y; jw < njobs; jw += gridDim.y) {
      const int which = (int)(jw & 1);
      const int64_t job = jw >> 1;
      const int b = (int)(job % nodes);
      const int64_t sig = job / nodes;
      if (D.R.layout == WX2_WPD && D.tree) {
          const int64_t heap = wx_quad_start(d) + b;
          if (!(heap <= D.ntree && D.tree[heap - 1])) continue;
      }
      const T *w1 = wx_inv2d_child<T>(D, sig, b, 2 * which);
      const T *w2 = wx_inv2d_child<T>(D, sig, b, 2 * which + 1);
      for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < (unsigned)per; e += gridDim.x * 256u) {
        const int ci = (int)(e / (unsigned)nr), ri = (int)(e - (unsigned)ci * (unsigned)nr);
        int r, c, cls, u;
        bool single, shifted = false;
        if (sm_mode) { r = sw + ri * 2 * s; cls = sv; u = ci; c = cls + u * s; single = true; shifted = (sw != sv); }
        else { r = ri; c = ci; cls = c & (s - 1); u = c >> d; single = false; }
        double v;
        if (single) v = wx_isdwt_point<T>(w1 + r, w2 + r, m, n, d, cls, u, shifted, filt);
        else v = 0.5 * (wx_isdwt_point<T>(w1 + r, w2 + r, m, n, d, cls, u, false, filt) +
                        wx_isdwt_point<T>(w1 + r, w2 + r, m, n, d, cls, u, true, filt));
        tmp[jw * mn + r + (int64_t)c * m] = (T)v;
      }
    }
}

// columns pass (dim 1): tmp[job][0], tmp[job][1] -> parent
template <typename T>
__global__ __launch_bounds__(256) void k_red2d_inv_dim1(WxInv2d D, const T *__restrict__ tmp, int nodes, int sm_mode,
                                                        int sv, int sw, WxFilt filt)
{
    const int m = D.R.m, n = D.R.n, d = D.R.d, s = 1 << d;
    const int64_t mn = (int64_t)m * n;
    const int nr = sm_mode ? (m >> d) : m;
    const int ncl = sm_mode ? (n >> d) : n;
    const int64_t per = (int64_t)nr * ncl;
    const int64_t njobs = D.R.batch * nodes;
    for (int64_t job = blockIdx.y; job < njobs; job += gridDim.y) {
      const int b = (int)(job % nodes);
      const int64_t sig = job / nodes;
      if (D.R.layout == WX2_WPD && D.tree) {
          const int64_t heap = wx_quad_start(d) + b;
          if (!(heap <= D.ntree && D.tree[heap - 1])) continue;
      }
      const T *t1 = tmp + (job * 2) * mn, *t2 = tmp + (job * 2 + 1) * mn;
      T *out = reinterpret_cast<T *>(D.out) + (sig * D.out_cols + ((D.R.layout == WX2_DWT || d == 0) ? 0 : b)) * mn;
      for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < (unsigned)per; e += gridDim.x * 256u) {
        const int ci = (int)(e / (unsigned)nr), ri = (int)(e - (unsigned)ci * (unsigned)nr);
        int r, c, cls, u;
        double v;
        if (sm_mode) {
            cls = sv; u = ri; r = cls + u * s; c = sv + ci * s;
            v = wx_isdwt_point<T>(t1 + (int64_t)c * m, t2 + (int64_t)c * m, 1, m, d, cls, u, sw != sv, filt);
        } else {
            r = ri; c = ci; cls = r & (s - 1); u = r >> d;
            v = 0.5 * (wx_isdwt_point<T>(t1 + (int64_t)c * m, t2 + (int64_t)c * m, 1, m, d, cls, u, false, filt) +
                       wx_isdwt_point<T>(t1 + (int64_t)c * m, t2 + (int64_t)c * m, 1, m, d, cls, u, true, filt));
        }
        out[r + (int64_t)c * m] = (T)v;
      }
    }
}

// iacdwt_step! 2-D: v = ((w1+w2)/sqrt2 + (w3+w4)/sqrt2)/sqrt2, same operation order as the reference
template <typename T>
__global__ __launch_bounds__(256) void k_red2d_iac(WxInv2d D, int nodes)
{
    const double sqrt2 = 1.4142135623730951;
    const int64_t mn = (int64_t)D.R.m * D.R.n;
    const int64_t njobs = D.R.batch * nodes;
    const int d = D.R.d;
    for (int64_t job = blockIdx.y; job < njobs; job += gridDim.y) {
      const int b = (int)(job % nodes);
      const int64_t sig = job / nodes;
      if (D.R.layout == WX2_WPD && D.tree) {
          const int64_t heap = wx_quad_start(d) + b;
          if (!(heap <= D.ntree && D.tree[heap - 1])) continue;
      }
      const T *w1 = wx_inv2d_child<T>(D, sig, b, 0), *w2 = wx_inv2d_child<T>(D, sig, b, 1);
      const T *w3 = wx_inv2d_child<T>(D, sig, b, 2), *w4 = wx_inv2d_child<T>(D, sig, b, 3);
      T *out = reinterpret_cast<T *>(D.out) + (sig * D.out_cols + ((D.R.layout == WX2_DWT || d == 0) ? 0 : b)) * mn;
      for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < (unsigned)mn; e += gridDim.x * 256u) {
        const T t1 = (T)(((double)w1[e] + (double)w2[e]) / sqrt2);
        const T t2 = (T)(((double)w3[e] + (double)w4[e]) / sqrt2);
        out[e] = (T)(((double)t1 + (double)t2) / sqrt2);
      }
    }
}

// ---- forward level in one pass ----------------------------------------------------------------------------------
// The two-pass level moves 9 images per node (parent read, 2 written + 2 read through scratch, 4 children written).
// Here a workgroup owns a strip of R rows of one node over ALL columns: the dim-1 step of the strip is computed
// straight from the parent (taps along the contiguous dimension: coalesced, neighbouring rows hit in cache) into two
// LDS images [column][row-in-strip]; the dim-2 step then runs out of LDS (whole rows resident: the periodic wrap and
// the 2^d dilation are plain index arithmetic) and stores the four children: 5 images per node.  The reference's
// in-place containers alias a child with its parent (swpt: child 0, sdwt: the next approximation), which a fused
// level cannot write while other workgroups still read the parent -- so a child that is decomposed again travels
// through a scratch image of its depth's parity (src_s / dst_s) and only final coefficients are written to xw.
template <typename T, bool AC>
__global__ __launch_bounds__(256) void k_red2d_fwd_fused(const T *__restrict__ x, T *__restrict__ xw,
                                                         const T *__restrict__ src_s, T *__restrict__ dst_s, WxRed2d D,
                                                         int nodes, int R, int CT, int hb, int W, WxFilt filt, WxAcFilt ac)
{
    // CT = n, hb = 0, W = n: the strip holds whole rows (wrap in LDS).  Wider images: tiles of CT columns with the
    // taps' reach as halo (hb before, W - CT - hb after), fetched with wrap from the parent; no wrap in LDS then.
    extern __shared__ __attribute__((aligned(16))) char wx_smem4[];
    const int m = D.m, n = D.n, d = D.d;
    const int64_t mn = (int64_t)m * n;
    T *lo1 = reinterpret_cast<T *>(wx_smem4), *hi1 = lo1 + (size_t)W * R;
    const int strips = m / R, ctiles = (n + CT - 1) / CT;
    const int s1 = (1 << d) % m, s2 = (1 << d) % n;
    const bool last = d + 1 == D.L;
    const int64_t total = D.batch * nodes * strips * ctiles;
    for (int64_t bs0 = blockIdx.x; bs0 < total; bs0 += gridDim.x) {
        const int64_t bs = bs0 / ctiles;
        const int c0 = (int)(bs0 - bs * ctiles) * CT;
        const int cw = n - c0 < CT ? n - c0 : CT;             // columns of this tile
        const int64_t job = bs / strips;
        const int r0 = (int)(bs - job * strips) * R;
        const int b = (int)(job % nodes);
        const int64_t sig = job / nodes;
        int64_t pv, pc[4];
        wx_red2d_slices(D, b, pv, pc);
        const T *src;
        T *dst[4];
        if (D.layout == WX2_WPD) {
            src = d == 0 ? x + sig * mn : xw + (sig * D.ncols + pv) * mn;
            for (int c = 0; c < 4; ++c) dst[c] = xw + (sig * D.ncols + pc[c]) * mn;
        } else if (D.layout == WX2_WPT) {
            src = d == 0 ? x + sig * mn : src_s + job * mn;
            for (int c = 0; c < 4; ++c)
                dst[c] = last ? xw + (sig * D.ncols + pc[c]) * mn : dst_s + (job * 4 + c) * mn;
        } else {                                          // DWT: child 0 is the next level's parent
            src = d == 0 ? x + sig * mn : src_s + sig * mn;
            dst[0] = last ? xw + (sig * D.ncols + pc[0]) * mn : dst_s + sig * mn;
            for (int c = 1; c < 4; ++c) dst[c] = xw + (sig * D.ncols + pc[c]) * mn;
        }
        const int wl = CT == n ? n : cw + (W - CT);           // staged columns of this tile
        for (int e = threadIdx.x; e < R * wl; e += 256) {
            const int lc = e / R, r = e - lc * R;
            int c = c0 - hb + lc;
            if (c < 0) c += n; else if (c >= n) c -= n;
            T lo, hi;
            wx_red_point<T, AC>(src + (int64_t)c * m, 1, m, r0 + r, s1, filt, ac, lo, hi);
            lo1[e] = lo; hi1[e] = hi;
            if (d == 0 && D.layout == WX2_WPD && lc >= hb && lc < hb + cw)
                xw[(sig * D.ncols) * mn + (int64_t)c * m + r0 + r] = src[(int64_t)c * m + r0 + r];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < R * cw; e += 256) {
            const int cl = e / R, r = e - cl * R;
            const int64_t o = (int64_t)(c0 + cl) * m + r0 + r;
            T lo, hi;
            wx_red_point<T, AC>(lo1 + r, R, wl, cl + hb, s2, filt, ac, lo, hi);
            dst[0][o] = lo; dst[1][o] = hi;
            wx_red_point<T, AC>(hi1 + r, R, wl, cl + hb, s2, filt, ac, lo, hi);
            dst[2][o] = lo; dst[3][o] = hi;
        }
        __syncthreads();
    }
}

// geometry of the one-pass level at dilation s: R rows per strip, tiles of CT columns with hb halo columns before and
// W - CT - hb after (CT = n: whole rows).  R = 0: not applicable (the caller takes the two passes).
struct WxRedTile { int R, CT, hb, W; };
template <typename T> static WxRedTile wx_red2d_fused_geom(int64_t m, int64_t n, int s, int F, bool ac)
{
    static const bool off = wx_getenv("WX_RED2D_FUSED") && atoi(wx_getenv("WX_RED2D_FUSED")) == 0;
    // LDS budget of a strip: 32 KiB (4 workgroups per CU hide the tap loads' latency; measured 64 / 32 / 16 KiB:
    // sdwt 2.11 / 1.65 / 2.47 ms, swpt 10.5 / 8.2 / 16.5 ms); the autocorrelation step has half the taps and is
    // indifferent (6.6 / 6.9 ms)
    static const size_t kib_env = wx_getenv("WX_RED2D_LDS_KIB") ? (size_t)atoi(wx_getenv("WX_RED2D_LDS_KIB")) : 0;
    const size_t kib = kib_env ? kib_env : 32;
    WxRedTile g = {0, 0, 0, 0};
    if (off || s >= n) return g;
    // store runs of a strip are R rows: below 64 bytes the one-pass level loses to the two passes
    // whole rows first (no redundant dim-1 work): 256 images of 256 x 256 Float64, swpt: 7.6 ms against 9.0 tiled
    for (int R = 128 / (int)sizeof(T); R * (int)sizeof(T) >= 64; R >>= 1) {
        if (m % R) continue;
        const int wmax = (int)(kib * 1024 / ((size_t)2 * R * sizeof(T)));
        if (n <= wmax) { g.R = R; g.CT = (int)n; g.hb = 0; g.W = (int)n; return g; }
    }
    // column tiles for wider Float64 images (Float32 tiles lose to the two passes: 1024 x 1024, swpt 13.7 ms against 11.8)
    if (sizeof(T) != 8) return g;
    for (int R = 128 / (int)sizeof(T); R * (int)sizeof(T) >= 64; R >>= 1) {
        if (m % R) continue;
        const int wmax = (int)(kib * 1024 / ((size_t)2 * R * sizeof(T)));
        const int hb = (F - 1) * s, ha = ac ? (F - 1) * s : (F - 2) * s;
        const int ct = wmax - hb - ha;
        if (ct >= hb + ha && ct >= 16) { g.R = R; g.CT = ct; g.hb = hb; g.W = wmax; return g; }   // at most 2x the dim-1 work
    }
    return g;
}

// grid.x over the elements of one job, grid.y over the jobs
static dim3 wx_grid2r(int64_t per, int64_t njobs)
{
    int64_t gx = (per + 255) / 256;
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    int64_t gy = njobs < 65535 ? njobs : 65535;
    if (gy < 1) gy = 1;
    return dim3((unsigned)gx, (unsigned)gy);
}

// ---- average-based inverse level in one pass ---------------------------------------------------------------------
// The mirror of k_red2d_fwd_fused for isdwt_step! 2-D (average of the two shifts in each dimension).  The two 1-D
// synthesis steps act on different axes and commute, so the strip first merges along dim 1 -- (w1, w3) and (w2, w4),
// taps along the contiguous dimension, straight from the children -- into two LDS images over all columns, then
// merges those along dim 2 out of LDS.  (The reference goes rows first; the results agree to rounding.)  5 images
// per node instead of 9.  Whole-row strips only; the shift-based variants and wider images keep the two passes.
template <typename T>
__global__ __launch_bounds__(256) void k_red2d_inv_fused(WxInv2d D, int nodes, int R, WxFilt filt)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem5[];
    const int m = D.R.m, n = D.R.n, d = D.R.d, s = 1 << d;
    const int64_t mn = (int64_t)m * n;
    T *ulo = reinterpret_cast<T *>(wx_smem5), *uhi = ulo + (size_t)n * R;
    const int strips = m / R;
    const int64_t total = D.R.batch * nodes * strips;
    for (int64_t bs = blockIdx.x; bs < total; bs += gridDim.x) {
        const int64_t job = bs / strips;
        const int r0 = (int)(bs - job * strips) * R;
        const int b = (int)(job % nodes);
        const int64_t sig = job / nodes;
        if (D.R.layout == WX2_WPD && D.tree) {
            const int64_t heap = wx_quad_start(d) + b;
            if (!(heap <= D.ntree && D.tree[heap - 1])) continue;      // (uniform for the workgroup)
        }
        const T *w1 = wx_inv2d_child<T>(D, sig, b, 0), *w2 = wx_inv2d_child<T>(D, sig, b, 1);
        const T *w3 = wx_inv2d_child<T>(D, sig, b, 2), *w4 = wx_inv2d_child<T>(D, sig, b, 3);
        T *out = reinterpret_cast<T *>(D.out) + (sig * D.out_cols + ((D.R.layout == WX2_DWT || d == 0) ? 0 : b)) * mn;
        for (int e = threadIdx.x; e < R * n; e += 256) {
            const int c = e / R, r = r0 + (e - c * R);
            const int cls = r & (s - 1), u = r >> d;
            const int64_t co = (int64_t)c * m;
            ulo[e] = (T)(0.5 * (wx_isdwt_point<T>(w1 + co, w3 + co, 1, m, d, cls, u, false, filt) +
                                wx_isdwt_point<T>(w1 + co, w3 + co, 1, m, d, cls, u, true, filt)));
            uhi[e] = (T)(0.5 * (wx_isdwt_point<T>(w2 + co, w4 + co, 1, m, d, cls, u, false, filt) +
                                wx_isdwt_point<T>(w2 + co, w4 + co, 1, m, d, cls, u, true, filt)));
        }
        __syncthreads();
        for (int e = threadIdx.x; e < R * n; e += 256) {
            const int c = e / R, rl = e - c * R;
            const int cls = c & (s - 1), u = c >> d;
            const double v = 0.5 * (wx_isdwt_point<T>(ulo + rl, uhi + rl, R, n, d, cls, u, false, filt) +
                                    wx_isdwt_point<T>(ulo + rl, uhi + rl, R, n, d, cls, u, true, filt));
            out[(int64_t)c * m + r0 + rl] = (T)v;
        }
        __syncthreads();
    }
}

static int64_t wx_red2d_ncols(int layout, int L)
{
    if (layout == WX2_DWT) return 3 * L + 1;
    if (layout == WX2_WPT) return (int64_t)1 << (2 * L);
    return ((((int64_t)1 << (2 * (L + 1))) - 1) / 3);
}

// tmp: 2 * 4^(L-1) * batch * m*n elements (1 node per level for the dwt layout)
template <typename T>
int wx_dev_red2d_fwd(const T *x, T *xw, int64_t m, int64_t n, int L, int layout, int64_t batch, const WxFilt &filt,
                     const WxAcFilt *ac, T *tmp, hipStream_t st)
{
    if (batch == 0 || m * n == 0) return WX_OK;
    WxAcFilt acz;
    if (ac) acz = *ac; else { acz.F = 0; acz.c1 = 0; }
    bool all_fused = L > 0;
    for (int d = 0; d < L; ++d) all_fused = all_fused && wx_red2d_fused_geom<T>(m, n, 1 << d, filt.F, ac != nullptr).R > 0;
    if (all_fused) {
        // scratch images of the nodes that are decomposed again: depth d in buf[d & 1] (each half of tmp holds the
        // 4^(L-1) nodes of the deepest intermediate depth)
        const int64_t half = (layout == WX2_DWT ? 1 : ((int64_t)1 << (2 * (L > 1 ? L - 1 : 0)))) * batch * m * n;
        T *buf[2] = {tmp, tmp + half};
        for (int d = 0; d < L; ++d) {
            WxRed2d D;
            D.layout = layout; D.L = L; D.d = d; D.m = (int)m; D.n = (int)n; D.ncols = wx_red2d_ncols(layout, L); D.batch = batch;
            const int nodes = layout == WX2_DWT ? 1 : (1 << (2 * d));
            const WxRedTile t = wx_red2d_fused_geom<T>(m, n, 1 << d, filt.F, ac != nullptr);
            const size_t lds = (size_t)2 * t.W * t.R * sizeof(T);
            int64_t g = batch * nodes * (m / t.R) * ((n + t.CT - 1) / t.CT);
            if (g > 256 * 8) g = 256 * 8;
            if (ac)
                hipLaunchKernelGGL((k_red2d_fwd_fused<T, true>), dim3((unsigned)g), dim3(256), lds, st, x, xw, (const T *)buf[d & 1],
                                   buf[(d + 1) & 1], D, nodes, t.R, t.CT, t.hb, t.W, filt, acz);
            else
                hipLaunchKernelGGL((k_red2d_fwd_fused<T, false>), dim3((unsigned)g), dim3(256), lds, st, x, xw, (const T *)buf[d & 1],
                                   buf[(d + 1) & 1], D, nodes, t.R, t.CT, t.hb, t.W, filt, acz);
        }
        WX_HIP_CHECK(hipGetLastError());
        return WX_OK;
    }
    for (int d = 0; d < L; ++d) {
        WxRed2d D;
        D.layout = layout; D.L = L; D.d = d; D.m = (int)m; D.n = (int)n; D.ncols = wx_red2d_ncols(layout, L); D.batch = batch;
        const int nodes = layout == WX2_DWT ? 1 : (1 << (2 * d));
        if (ac) {
            hipLaunchKernelGGL((k_red2d_fwd_dim1<T, true>), wx_grid2r(m * n, batch * nodes), dim3(256), 0, st, x, xw, tmp, D, nodes, filt, acz);
            hipLaunchKernelGGL((k_red2d_fwd_dim2<T, true>), wx_grid2r(m * n, 2 * batch * nodes), dim3(256), 0, st, xw, (const T *)tmp, D, nodes, filt, acz);
        } else {
            hipLaunchKernelGGL((k_red2d_fwd_dim1<T, false>), wx_grid2r(m * n, batch * nodes), dim3(256), 0, st, x, xw, tmp, D, nodes, filt, acz);
            hipLaunchKernelGGL((k_red2d_fwd_dim2<T, false>), wx_grid2r(m * n, 2 * batch * nodes), dim3(256), 0, st, xw, (const T *)tmp, D, nodes, filt, acz);
        }
    }
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

// sm < 0: average based.  ac: autocorrelation inverse (no filter).  s0/s1: level buffers, tmp: 2 slices per job
template <typename T>
int wx_dev_red2d_inv(const T *xw, T *x, int64_t m, int64_t n, int L, int layout, int64_t ncols, int64_t batch, int64_t sm,
                     bool ac, const uint8_t *dtree, int64_t ntree, const WxFilt &filt, T *s0, T *s1, T *tmp,
                     hipStream_t st)
{
    if (batch == 0 || m * n == 0) return WX_OK;
    const int64_t mn = m * n;
    if (L == 0) {
        WX_HIP_CHECK(hipMemcpy2DAsync(x, mn * sizeof(T), xw, (size_t)mn * ncols * sizeof(T), mn * sizeof(T), batch,
                                      hipMemcpyDeviceToDevice, st));
        return WX_OK;
    }
    int64_t sd[40];
    sd[0] = 0;
    if (sm >= 0) { int64_t acc = 0; for (int d = 0; d < L; ++d) { acc += ((sm >> d) & 1) << d; sd[d + 1] = acc; } }
    T *bufs[2] = {s0, s1};
    for (int d = L - 1; d >= 0; --d) {
        WxInv2d D;
        D.R.layout = layout; D.R.L = L; D.R.d = d; D.R.m = (int)m; D.R.n = (int)n; D.R.ncols = ncols; D.R.batch = batch;
        D.in = xw; D.tree = dtree; D.ntree = ntree;
        const int nodes_d = layout == WX2_DWT ? 1 : (1 << (2 * d));
        const int64_t nodes_c = layout == WX2_DWT ? 1 : ((int64_t)1 << (2 * (d + 1)));
        D.cur = bufs[(d + 1) & 1]; D.cur_cols = nodes_c;
        if (d == 0) { D.out = x; D.out_cols = 1; } else { D.out = bufs[d & 1]; D.out_cols = nodes_d; }
        const int64_t jobs = batch * nodes_d;
        if (ac) {
            hipLaunchKernelGGL(k_red2d_iac<T>, wx_grid2r(mn, jobs), dim3(256), 0, st, D, nodes_d);
        } else {
            const int sm_mode = sm >= 0 ? 1 : 0;
            const int sv = sm >= 0 ? (int)sd[d] : 0, sw = sm >= 0 ? (int)sd[d + 1] : 0;
            if (!sm_mode && (m & (m - 1)) == 0 && (n & (n - 1)) == 0) {
                const WxRedTile t = wx_red2d_fused_geom<T>(m, n, 1 << d, filt.F, false);
                // whole-row strips; measured against the two passes (256 images, L = 3 iswpt): 128 x 128 Float64
                // 9.4 ms vs 12.3, 256 x 256 Float32 8.6 vs 9.8, but 256 x 256 Float64 (strips of 8 rows) 12.7 vs 11.3
                if (t.R && t.CT == (int)n && (size_t)n * sizeof(T) <= 1024) {
                    int64_t g = jobs * (m / t.R);
                    if (g > 256 * 8) g = 256 * 8;
                    hipLaunchKernelGGL(k_red2d_inv_fused<T>, dim3((unsigned)g), dim3(256), (size_t)2 * n * t.R * sizeof(T), st, D,
                                       nodes_d, t.R, filt);
                    continue;
                }
            }
            const int64_t per2 = sm_mode ? (m >> (d + 1)) * (n >> d) : mn;
            const int64_t per1 = sm_mode ? (m >> d) * (n >> d) : mn;
            hipLaunchKernelGGL(k_red2d_inv_dim2<T>, wx_grid2r(per2, jobs * 2), dim3(256), 0, st, D, tmp, nodes_d, sm_mode, sv, sw, filt);
            hipLaunchKernelGGL(k_red2d_inv_dim1<T>, wx_grid2r(per1, jobs), dim3(256), 0, st, D, (const T *)tmp, nodes_d, sm_mode, sv, sw, filt);
        }
    }
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

#define WX_INST(T)                                                                                                    \
    template int wx_dev_red2d_fwd<T>(const T *, T *, int64_t, int64_t, int, int, int64_t, const WxFilt &, const WxAcFilt *, \
                                     T *, hipStream_t);                                                               \
    template int wx_dev_red2d_inv<T>(const T *, T *, int64_t, int64_t, int, int, int64_t, int64_t, int64_t, bool,      \
                                     const uint8_t *, int64_t, const WxFilt &, T *, T *, T *, hipStream_t);
WX_INST(double)
WX_INST(float)
