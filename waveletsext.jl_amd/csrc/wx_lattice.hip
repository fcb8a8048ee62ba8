// wx_lattice.hip -- host side of the lattice kernels (device code: wx_lattice_dev.h): factorisation, launchers for
// 4096-sample signals; the interleaved shorter signals are launched from wx_lattice_sh.hip (a separate translation unit so
// that the two sets of kernel instantiations compile in parallel)
#include "wx_lattice_dev.h"

// Lattice factorisation of the polyphase matrix (long double): G(w) = [[Qe, Qo], [-w^J Qo(1/w), w^J Qe(1/w)]]
// = R_J Lambda(w) R_{J-1} ... Lambda(w) R_0 with R_j = c_j [[1, t_j], [-t_j, 1]], Lambda = diag(1, w).
// Returns false when a rotation is too close to a quarter turn (|t| huge) or the rebuilt filter misses q.
bool wx_lattice_factor(const WxFilt &filt, int L, bool inverse, WxLat *out)
{
    const int F = filt.F;
    if (F < 2 || (F & 1) || F / 2 > WX_LAT_MAXS) return false;
    const int J = F / 2 - 1;
    long double G[WX_LAT_MAXS][2][2] = {};
    for (int m = 0; m <= J; ++m) {
        G[m][0][0] = filt.q[2 * m];
        G[m][0][1] = filt.q[2 * m + 1];
        G[J - m][1][0] = -(long double)filt.q[2 * m + 1];
        G[J - m][1][1] = filt.q[2 * m];
    }
    long double t[WX_LAT_MAXS], cs[WX_LAT_MAXS];
    for (int deg = J; deg >= 1; --deg) {
        const int k = (fabsl(G[deg][0][0]) + fabsl(G[deg][1][0]) >= fabsl(G[deg][0][1]) + fabsl(G[deg][1][1])) ? 0 : 1;
        const long double a0 = G[deg][0][k], a1 = G[deg][1][k], r = hypotl(a0, a1);
        if (!(r > 0)) return false;
        const long double c = a1 / r, s = a0 / r;
        long double Hm[WX_LAT_MAXS][2][2];
        for (int m = 0; m <= deg; ++m)
            for (int col = 0; col < 2; ++col) {
                Hm[m][0][col] = c * G[m][0][col] - s * G[m][1][col];
                Hm[m][1][col] = s * G[m][0][col] + c * G[m][1][col];
            }
        for (int col = 0; col < 2; ++col)
            if (fabsl(Hm[deg][0][col]) > 1e-13L || fabsl(Hm[0][1][col]) > 1e-13L) return false;   // not paraunitary
        for (int m = 0; m < deg; ++m)
            for (int col = 0; col < 2; ++col) {
                G[m][0][col] = Hm[m][0][col];
                G[m][1][col] = Hm[m + 1][1][col];
            }
        if (fabsl(c) < 1e-3L) return false;
        t[deg] = s / c;
        cs[deg] = c;
    }
    {
        const long double c = G[0][0][0], s = G[0][0][1];
        if (fabsl(G[0][1][0] + s) > 1e-13L || fabsl(G[0][1][1] - c) > 1e-13L || fabsl(c) < 1e-3L) return false;
        t[0] = s / c;
        cs[0] = c;
    }
    long double gain1 = 1;
    double tr[WX_LAT_MAXS];
    for (int j = 0; j <= J; ++j) {
        tr[j] = (double)t[j];
        gain1 *= cs[j];
    }
    // rebuild the filter from the rounded tangents: row 0 of the product must be (Qe, Qo)
    long double P[WX_LAT_MAXS][2][2] = {};
    P[0][0][0] = 1; P[0][0][1] = tr[0]; P[0][1][0] = -(long double)tr[0]; P[0][1][1] = 1;
    for (int j = 1; j <= J; ++j) {
        long double Q[WX_LAT_MAXS][2][2] = {};
        const long double tj = tr[j];
        for (int m = 0; m < j; ++m)
            for (int col = 0; col < 2; ++col) {
                // Lambda: row 1 advances (w^1), then the rotation
                Q[m][0][col] += P[m][0][col];
                Q[m + 1][0][col] += tj * P[m][1][col];
                Q[m][1][col] += -tj * P[m][0][col];
                Q[m + 1][1][col] += P[m][1][col];
            }
        for (int m = 0; m <= j; ++m)
            for (int r2 = 0; r2 < 2; ++r2)
                for (int col = 0; col < 2; ++col) P[m][r2][col] = Q[m][r2][col];
    }
    for (int m = 0; m <= J; ++m)
        if (fabsl(gain1 * P[m][0][0] - filt.q[2 * m]) > 1e-14L || fabsl(gain1 * P[m][0][1] - filt.q[2 * m + 1]) > 1e-14L)
            return false;
    // shear form
    long double sig = 1;
    for (int j = 0; j < WX_LAT_MAXS; ++j) {
        if (j <= J) {
            const long double c2 = 1 / (1 + t[j] * t[j]);
            out->p[j] = (double)(t[j] / sig);
            out->kap[j] = (double)(sig * t[j] * c2);
            sig *= c2;
        } else {
            out->p[j] = out->kap[j] = 0.0;
        }
    }
    const long double gL = powl(gain1, L);
    out->g0 = (double)(inverse ? 1 / gL : gL);
    out->g2 = (double)(inverse ? gain1 * gain1 : 1 / (gain1 * gain1));
    // the intermediate values range over g^(+-L): keep them far from the limits of Float64
    return std::isfinite(out->g0) && fabsl(gL) > 1e-60L && fabsl(gL) < 1e60L;
}

// the factorisation for other translation units (wx_lattice2d.hip): p[j], kap[j] (j < F/2), g0 = g^(+-L), g2 = g^(-+2)
bool wx_lattice_coeffs(const WxFilt &filt, int L, bool inverse, double *p, double *kap, double *g0, double *g2)
{
    WxLat c;
    if (!wx_lattice_factor(filt, L, inverse, &c)) return false;
    for (int j = 0; j < WX_LAT_MAXS; ++j) { p[j] = c.p[j]; kap[j] = c.kap[j]; }
    *g0 = c.g0;
    *g2 = c.g2;
    return true;
}

// 0 = not applicable (the caller takes the general kernels), 1 = launched, < 0 = HIP error code of the C ABI
int wx_lattice_launch_sh(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, int64_t in_stride,
                         const WxFilt &filt, hipStream_t st);          // wx_lattice_sh.hip
int wx_lattice_launch_g(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, int64_t in_stride,
                        const WxFilt &filt, hipStream_t st);           // wx_lattice_sg.hip
int wx_lattice_wpd_g_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st);   // wx_lattice_sgw.hip
int wx_lattice_launch_lo(bool inverse, const double *x, double *y, int L, int64_t batch, int64_t in_stride, const WxFilt &filt,
                         hipStream_t st);                              // wx_lattice_lo.hip

static int wx_lattice_launch(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, int64_t in_stride,
                             const WxFilt &filt, hipStream_t st)
{
    static const bool off = wx_getenv("WX_LATTICE") && atoi(wx_getenv("WX_LATTICE")) == 0;
    static const bool off_sh = wx_getenv("WX_LATTICE_SH") && atoi(wx_getenv("WX_LATTICE_SH")) == 0;
    if (!off && !off_sh && (n == 2048 || n == 1024))
        return wx_lattice_launch_sh(inverse, x, y, n, L, batch, inverse ? in_stride : n, filt, st);
    if (!off && !off_sh && n >= 64 && n <= 512)
        return wx_lattice_launch_g(inverse, x, y, n, L, batch, inverse ? in_stride : n, filt, st);
    static const bool off_lo = wx_getenv("WX_LATTICE_LO") && atoi(wx_getenv("WX_LATTICE_LO")) == 0;
    if (!off && !off_lo && n == 4096 && L >= 1 && L <= 5) return wx_lattice_launch_lo(inverse, x, y, L, batch, inverse ? in_stride : n, filt, st);
    if (off || n != 4096 || L < 6 || L > 12 || filt.F < 2 || batch <= 0) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    if (inverse && (in_stride & 3)) return 0;
    WxLat cf;
    if (!wx_lattice_factor(filt, L, inverse, &cf)) return 0;
    static const int wg_per_cu = wx_getenv("WX_LATTICE_WG") ? atoi(wx_getenv("WX_LATTICE_WG")) : 0;
    int64_t grid = batch;
    if (wg_per_cu > 0 && grid > (int64_t)256 * wg_per_cu) grid = (int64_t)256 * wg_per_cu;
    if (grid > 0x7fffffff) grid = 0x7fffffff;
    // wavefronts per SIMD the kernels are compiled for (amdgpu_waves_per_eu): the forward fits 3 (166 registers); the
    // inverse needs 194 registers without spills, and its spills at 3 cost 15 % extra HBM traffic (scratch) and 6 % time
    // (db4 L = 10: 0.84 ms at 3, 0.78 ms at 2; the forward built for 2 is 2 % slower than for 3).  WX_LATTICE_INV_WPE = 3
    // selects the other build of the inverse.
    // (the three-wavefront build of the inverse was reachable through WX_LATTICE_INV_WPE=3 only: not built since round 6)
#define WX_GO(NSS)                                                                                                  \
    case NSS:                                                                                                       \
        if (inverse)                                                                                                \
            hipLaunchKernelGGL((k_lat_iwpt_f64<NSS, 2>), dim3((unsigned)grid), dim3(64), 0, st, x, y, L, batch, in_stride, cf); \
        else                                                                                                        \
            hipLaunchKernelGGL((k_lat_wpt_f64<NSS, 3>), dim3((unsigned)grid), dim3(64), 0, st, x, y, L, batch, cf);  \
        break;
    switch (wx_lat_stages(filt.F)) {
        WX_GO(1) WX_GO(2) WX_GO(4) WX_GO(6) WX_GO(8) WX_GO(10)
    default: return 0;
    }
#undef WX_GO
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpt launch", __FILE__, __LINE__);
    return 1;
}

// wpd: 0 = not applicable, 1 = launched, < 0 = error
int wx_lattice_wpd_sh_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st);   // wx_lattice_shw.hip

int wx_lattice_wpd_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    static const bool off = (wx_getenv("WX_LATTICE") && atoi(wx_getenv("WX_LATTICE")) == 0) ||
                            (wx_getenv("WX_LATTICE_WPD") && atoi(wx_getenv("WX_LATTICE_WPD")) == 0);
    static const bool off_sh = wx_getenv("WX_LATTICE_SH") && atoi(wx_getenv("WX_LATTICE_SH")) == 0;
    if (!off && !off_sh && (n == 2048 || n == 1024) && x != (const double *)y) return wx_lattice_wpd_sh_f64(x, y, n, L, batch, filt, st);
    if (!off && !off_sh && n >= 64 && n <= 512 && x != (const double *)y) return wx_lattice_wpd_g_f64(x, y, n, L, batch, filt, st);
    if (off || n != 4096 || L < 1 || L > 12 || filt.F < 2 || batch <= 0 || batch > 0x7fffffff) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, false, &cw.c)) return 0;
    // g^l for every level (g0 = g^L, g2 = g^-2): g = (g2)^(-1/2) with the sign of g0's L-th root
    {
        const long double g2 = cw.c.g2;
        long double g = 1 / sqrtl(g2);
        // the sign of g: recompute the product of the cosines
        WxLat tmp;
        if (!wx_lattice_factor(filt, 1, false, &tmp)) return 0;
        g = tmp.g0;
        long double acc = 1;
        for (int l = 0; l <= 12; ++l) { cw.gl[l] = (double)acc; acc *= g; }
    }
    // built for 2 wavefronts per SIMD (174-178 registers, no spills; the 3-wavefront build spills and is 1.5 % slower)
#define WX_GOW(NSS)                                                                                                 \
    case NSS:                                                                                                       \
        hipLaunchKernelGGL((k_lat_wpd_f64<NSS, 2>), dim3((unsigned)batch), dim3(64), 0, st, x, y, L, batch, cw);     \
        break;
    switch (wx_lat_stages(filt.F)) {
        WX_GOW(1) WX_GOW(2) WX_GOW(4) WX_GOW(6) WX_GOW(8) WX_GOW(10)
    default: return 0;
    }
#undef WX_GOW
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpd launch", __FILE__, __LINE__);
    return 1;
}

bool wx_lattice_applicable_f64(const WxFilt &filt)
{
    static const bool off = wx_getenv("WX_LATTICE") && atoi(wx_getenv("WX_LATTICE")) == 0;
    if (off || filt.F < 2 || filt.F / 2 > WX_LAT_MAXS) return false;
    WxLat tmp;
    return wx_lattice_factor(filt, 6, false, &tmp);
}

// 8192-sample signals in one pass: two wavefronts per signal, the first level in direct form inside the load / store phase (wx_lattice_8k.h)
int wx_lattice_wpt8k_f64(const double *x, double *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st);
int wx_lattice_iwpt8k_f64(const double *xw, double *y, int L, int64_t batch, int64_t in_stride, const WxFilt &filt, hipStream_t st);

int wx_lattice_wpt_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    static const bool off = wx_getenv("WX_LATTICE") && atoi(wx_getenv("WX_LATTICE")) == 0;
    if (n == 8192 && !off) return wx_lattice_wpt8k_f64(x, y, L, batch, filt, st);
    return wx_lattice_launch(false, x, y, n, L, batch, n, filt, st);
}
int wx_lattice_iwpt_f64(const double *xw, double *y, int64_t n, int L, int64_t batch, int64_t in_stride, const WxFilt &filt,
                        hipStream_t st)
{
    static const bool off = wx_getenv("WX_LATTICE") && atoi(wx_getenv("WX_LATTICE")) == 0;
    if (n == 8192 && !off) return wx_lattice_iwpt8k_f64(xw, y, L, batch, in_stride, filt, st);
    return wx_lattice_launch(true, xw, y, n, L, batch, in_stride, filt, st);
}

// ---- tree-driven transforms (wx_lattice_tree.h) -----------------------------------------------------------------
#define WX_TREE_DECL(k)                                                                                              \
    int wx_lattice_tree##k##_f64(bool, const double *, double *, int64_t, int, int64_t, int64_t, int64_t, const WxFilt &, \
                                 const uint8_t *, int64_t, const WxThreshArg *, hipStream_t, int64_t);
WX_TREE_DECL(0f) WX_TREE_DECL(0i) WX_TREE_DECL(1f) WX_TREE_DECL(1i) WX_TREE_DECL(2f) WX_TREE_DECL(2i)
#undef WX_TREE_DECL

// short signals (512 .. 64 samples, wx_lattice_tree_s.h): filters up to 16 taps, no threshold riding on the loads
#define WX_TREES_DECL(k)                                                                                                             \
    int wx_lattice_trees_##k##_f64(const double *, double *, int64_t, int, int64_t, int64_t, int64_t, const WxFilt &, const uint8_t *, int64_t, hipStream_t, int64_t); \
    int wx_lattice_trees_##k##_f32(const float *, float *, int64_t, int, int64_t, int64_t, int64_t, const WxFilt &, const uint8_t *, int64_t, hipStream_t, int64_t);
WX_TREES_DECL(3f) WX_TREES_DECL(3i) WX_TREES_DECL(4f) WX_TREES_DECL(4i) WX_TREES_DECL(5f) WX_TREES_DECL(5i) WX_TREES_DECL(6f) WX_TREES_DECL(6i)
#undef WX_TREES_DECL
static bool wx_lattice_trees_short(int64_t n, const WxFilt &filt)
{
    static const bool off = wx_getenv("WX_LATTICE_TREES") && atoi(wx_getenv("WX_LATTICE_TREES")) == 0;
    return !off && (n == 512 || n == 256 || n == 128 || n == 64) && filt.F <= 16;
}
#define WX_TREES_GO(TS)                                                                                                              \
    switch (n) {                                                                                                                     \
    case 512: return inverse ? wx_lattice_trees_3i_##TS(x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, st, out_stride) \
                             : wx_lattice_trees_3f_##TS(x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, st, out_stride); \
    case 256: return inverse ? wx_lattice_trees_4i_##TS(x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, st, out_stride) \
                             : wx_lattice_trees_4f_##TS(x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, st, out_stride); \
    case 128: return inverse ? wx_lattice_trees_5i_##TS(x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, st, out_stride) \
                             : wx_lattice_trees_5f_##TS(x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, st, out_stride); \
    default: return inverse ? wx_lattice_trees_6i_##TS(x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, st, out_stride) \
                            : wx_lattice_trees_6f_##TS(x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, st, out_stride); \
    }

bool wx_lattice_tree_applicable_f64(int64_t n, const WxFilt &filt)
{
    static const bool off = (wx_getenv("WX_LATTICE") && atoi(wx_getenv("WX_LATTICE")) == 0) ||
                            (wx_getenv("WX_LATTICE_TREE") && atoi(wx_getenv("WX_LATTICE_TREE")) == 0);
    return !off && (n == 4096 || n == 2048 || n == 1024 || wx_lattice_trees_short(n, filt)) && wx_lattice_applicable_f64(filt);
}

int wx_lattice_tree_f64(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, int64_t in_stride,
                        int64_t col_stride, const WxFilt &filt, const uint8_t *dstatus, int64_t nstatus, hipStream_t st,
                        const WxThreshArg *thr, int64_t out_stride)
{
    if (!wx_lattice_tree_applicable_f64(n, filt)) return 0;
    if (n <= 512) {
        if (thr && (thr->t || thr->head)) return 0;
        WX_TREES_GO(f64)
    }
#define WX_TREE_GO(k) (inverse ? wx_lattice_tree##k##i_f64(inverse, x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, thr, st, out_stride) \
                               : wx_lattice_tree##k##f_f64(inverse, x, y, n, L, batch, in_stride, col_stride, filt, dstatus, nstatus, thr, st, out_stride))
    if (n == 4096) return WX_TREE_GO(0);
    if (n == 2048) return WX_TREE_GO(1);
    return WX_TREE_GO(2);
#undef WX_TREE_GO
}

// ---- Float32 signals along a tree (wx_lattice_tree32.h) ------------------------------------------------------------------------
#define WX_T32(k, d) int wx_lattice_tree32_##k##d(bool, const float *, float *, int64_t, int, int64_t, int64_t, const WxFilt &, const uint8_t *, int64_t, \
                                                  const WxThreshArg *, hipStream_t, int64_t);
WX_T32(0, f) WX_T32(0, i) WX_T32(1, f) WX_T32(1, i) WX_T32(2, f) WX_T32(2, i)
#undef WX_T32
bool wx_lattice_tree_applicable_f32(int64_t n, const WxFilt &filt)
{
    static const bool off = wx_getenv("WX_LATTICE_TREE32") && atoi(wx_getenv("WX_LATTICE_TREE32")) == 0;
    return !off && wx_lattice_tree_applicable_f64(n, filt);
}
int wx_lattice_tree_f32(bool inverse, const float *x, float *y, int64_t n, int L, int64_t batch, int64_t in_stride, const WxFilt &filt,
                        const uint8_t *dstatus, int64_t nstatus, hipStream_t st, const WxThreshArg *thr, int64_t out_stride)
{
    if (!wx_lattice_tree_applicable_f32(n, filt)) return 0;
    if (n <= 512) {
        if (thr && (thr->t || thr->head)) return 0;
        const int64_t col_stride = 0;
        WX_TREES_GO(f32)
    }
#define WX_TREE_GO(k) (inverse ? wx_lattice_tree32_##k##i(inverse, x, y, n, L, batch, in_stride, filt, dstatus, nstatus, thr, st, out_stride) \
                               : wx_lattice_tree32_##k##f(inverse, x, y, n, L, batch, in_stride, filt, dstatus, nstatus, thr, st, out_stride))
    if (n == 4096) return WX_TREE_GO(0);
    if (n == 2048) return WX_TREE_GO(1);
    return WX_TREE_GO(2);
#undef WX_TREE_GO
}
