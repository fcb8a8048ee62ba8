/*
 * wx_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU oracle for the WaveletsExt.jl
 * wavelet-packet hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the CPU
 * baseline -- never as the product path.
 *
 * The reference (/root/reference, pure Julia) cannot be compiled or imported in
 * this image (no Julia), and its 1-D wpt/iwpt, filter tables, 1-D maketree /
 * isvalidtree and makereverseqmfpair live in the un-vendored dependency
 * Wavelets.jl (Project.toml:19,30, compat "0.9, 0.10", no Manifest).  This file
 * restates the reference line by line (citations on every function; paths are
 * relative to /root/reference/src/mod) and restates the published Wavelets.jl
 * semantics where the reference only calls into it.
 * Pinned by the reference's own KATs: tests/test_oracle_kat.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Julia mod1(x, n) in 1..n */
static inline int64_t wxo_mod1(int64_t x, int64_t n)
{
    int64_t r = (x - 1) % n;
    if (r < 0) r += n;
    return r + 1;
}

/* Wavelets.jl WT.makereverseqmfpair(f, true) -> (reverse(qmf), mirror(qmf)), bound by the
 * reference as `g, h = ...` (DWT.jl:141).  mirror(q)[i] = (-1)^(i-1) q[i] (1-based). */
void wxo_makereverseqmfpair(const double *qmf, int F, double *g, double *h)
{
    for (int i = 0; i < F; i++) {
        g[i] = qmf[F - 1 - i];
        h[i] = (i & 1) ? -qmf[i] : qmf[i];
    }
}

/* Wavelets.jl Util.maxtransformlevels(n): number of times n is divisible by 2 */
int wxo_maxtransformlevels(int64_t n)
{
    if (n < 2) return 0;
    int tl = 0;
    while ((n & 1) == 0) { n >>= 1; tl++; }
    return tl;
}
int wxo_isdyadic(int64_t n) { return n >= 1 && (n & (n - 1)) == 0; }
int wxo_ndyadicscales(int64_t n) { int s = 0; while (((int64_t)1 << (s + 1)) <= n) s++; return s; }

/* utils/utils_tree.jl:252-263  getdepth(idx, :binary) = floor(log2(idx)) */
int wxo_getdepth_binary(int64_t idx) { int d = 0; while (idx > 1) { idx >>= 1; d++; } return d; }
/* getdepth(idx, :quad) = floor(log(4, 3idx-2)), in integer arithmetic (App. D) */
int wxo_getdepth_quad(int64_t idx)
{
    int64_t t = 3 * idx - 2; int d = 0;
    while (t >= 4) { t >>= 2; d++; }
    return d;
}
/* utils/utils_tree.jl:285-293  gettreelength */
int64_t wxo_gettreelength1d(int64_t n) { return ((int64_t)1 << wxo_maxtransformlevels(n)) - 1; }
int64_t wxo_gettreelength2d(int64_t n, int64_t m)
{
    int L = wxo_maxtransformlevels(n < m ? n : m);
    return (((int64_t)1 << (2 * L)) - 1) / 3;
}

/* Wavelets.jl Util.maketree(n, L, s): BitVector(n-1); :full (s=0) sets 1..2^L-1,
 * :dwt (s=1) sets 1,2,4,..,2^(L-1).  Asserts isdyadic(n), 0<=L<=maxtransformlevels(n). */
int wxo_maketree1d(uint8_t *tree, int64_t n, int L, int s)
{
    if (!wxo_isdyadic(n)) return -1;
    if (!(0 <= L && L <= wxo_maxtransformlevels(n))) return -1;
    memset(tree, 0, (size_t)(n - 1));
    if (s == 0) for (int64_t i = 1; i <= ((int64_t)1 << L) - 1; i++) tree[i - 1] = 1;
    else for (int i = 0; i <= L - 1; i++) tree[((int64_t)1 << i) - 1] = 1;
    return 0;
}
/* utils/utils_tree.jl:193-222  maketree(n, m, L, s) quad tree */
int wxo_maketree2d(uint8_t *tree, int64_t n, int64_t m, int L, int s)
{
    int L0 = wxo_maxtransformlevels(n < m ? n : m);
    if (!(0 <= L && L <= L0)) return -1;
    int64_t nq = wxo_gettreelength2d(n, m);
    memset(tree, 0, (size_t)nq);
    if (s == 0) {
        int64_t tot = 0, p = 1;
        for (int i = 0; i <= L - 1; i++) { tot += p; p *= 4; }
        for (int64_t i = 1; i <= tot; i++) tree[i - 1] = 1;
    } else {
        tree[0] = 1;                                   /* :212 sets the root unconditionally */
        for (int i = 0; i <= L - 2; i++) tree[((((int64_t)1 << (2 * i + 2)) + 2) / 3) - 1] = 1;
    }
    return 0;
}

/* Wavelets.jl Util.isvalidtree(x::Vector, b): length(b)==length(x)-1 and no true node under a
 * false parent (cross-checked by test/utils.jl:9-12) */
int wxo_isvalidtree1d(int64_t n, const uint8_t *b, int64_t nb)
{
    if (nb != n - 1) return 0;
    for (int64_t i = 1; 2 * i + 1 <= nb; i++)
        if (!b[i - 1] && (b[2 * i - 1] || b[2 * i])) return 0;
    return 1;
}
/* utils/utils_tree.jl:13-29  isvalidtree(x::Matrix, b) */
int wxo_isvalidtree2d(int64_t n, int64_t m, const uint8_t *b, int64_t nb)
{
    if (wxo_gettreelength2d(n, m) != nb) return 0;
    int L0 = wxo_getdepth_quad(nb);
    int64_t ns = (((int64_t)1 << (2 * L0)) - 1) / 3;
    for (int64_t i = 1; i <= ns; i++) {
        int isnode = b[i - 1];
        int haschild = b[4 * i - 2 - 1] || b[4 * i - 1 - 1] || b[4 * i - 1] || b[4 * i + 1 - 1];
        if (!isnode && haschild) return 0;
    }
    return 1;
}

/* utils/utils_tree.jl:122-157  getleaf(tree, :binary); result has 2*len+1 entries */
int wxo_getleaf_binary(uint8_t *result, const uint8_t *tree, int64_t nt)
{
    int L0 = wxo_getdepth_binary(nt);
    if ((((int64_t)1 << (L0 + 1)) - 1) != nt) return -1;
    int64_t n = (int64_t)1 << (L0 + 1);
    if (!wxo_isvalidtree1d(n, tree, nt)) return -1;
    memset(result, 0, (size_t)(n + nt));
    result[0] = 1;
    for (int64_t i = 1; i <= nt; i++) {
        if (!tree[i - 1]) continue;
        result[i - 1] = 0;
        result[(i << 1) - 1] = 1;
        result[(i << 1) + 1 - 1] = 1;
    }
    return 0;
}
/* getleaf(tree, :quad); result has 4^(L0+1) + nt entries */
int wxo_getleaf_quad(uint8_t *result, const uint8_t *tree, int64_t nt)
{
    int L0 = wxo_getdepth_quad(nt);
    if (((((int64_t)1 << (2 * L0 + 2)) - 1) / 3) != nt) return -1;
    int64_t n = (int64_t)1 << (2 * L0 + 2);
    int64_t ns = (int64_t)1 << (L0 + 1);
    if (!wxo_isvalidtree2d(ns, ns, tree, nt)) return -1;
    memset(result, 0, (size_t)(n + nt));
    result[0] = 1;
    for (int64_t i = 1; i <= nt; i++) {
        if (!tree[i - 1]) continue;
        result[i - 1] = 0;
        for (int c = 0; c < 4; c++) result[4 * i - 2 + c - 1] = 1;
    }
    return 0;
}

/* Utils.jl:465-490  getrowrange(n, idx) -> [lo, hi] 1-based inclusive; -1 on the @assert */
int wxo_getrowrange(int64_t n, int64_t idx, int64_t *lo, int64_t *hi)
{
    int L0 = wxo_maxtransformlevels(n);
    int64_t k = (((int64_t)1 << (2 * L0 + 2)) - 1) / 3;
    if (!(0 < idx && idx <= k)) return -1;
    if (idx == 1) { *lo = 1; *hi = n; return 0; }
    int64_t parent = (idx + 2) / 4;
    int64_t pl, ph;
    wxo_getrowrange(n, parent, &pl, &ph);
    int64_t mid = (pl + ph) / 2;
    if (idx < 4 * parent) { *lo = pl; *hi = mid; } else { *lo = mid + 1; *hi = ph; }
    return 0;
}
/* Utils.jl:517-542  getcolrange(n, idx) */
int wxo_getcolrange(int64_t n, int64_t idx, int64_t *lo, int64_t *hi)
{
    int L0 = wxo_maxtransformlevels(n);
    int64_t k = (((int64_t)1 << (2 * L0 + 2)) - 1) / 3;
    if (!(0 < idx && idx <= k)) return -1;
    if (idx == 1) { *lo = 1; *hi = n; return 0; }
    int64_t parent = (idx + 2) / 4;
    int64_t pl, ph;
    wxo_getcolrange(n, parent, &pl, &ph);
    int64_t mid = (pl + ph) / 2;
    if ((idx & 1) == 0) { *lo = pl; *hi = mid; } else { *lo = mid + 1; *hi = ph; }
    return 0;
}

/* Utils.jl:297-305  main2depthshift(sm, L) -> sd[0..L]; -1 on @assert sm < 1<<L */
int wxo_main2depthshift(int64_t sm, int L, int64_t *sd)
{
    if (!(sm < ((int64_t)1 << L))) return -1;
    sd[0] = 0;
    int64_t acc = 0;
    for (int d = 0; d < L; d++) {
        int64_t bit = (sm >> d) & 1;          /* digits!(sb, sm, base=2) */
        acc += bit << d;                      /* sb .<< d |> cumsum */
        sd[d + 1] = acc;
    }
    return 0;
}

/* BestBasis.jl:128-140  delete_subtree!(bt, i, type); type 0 binary, 1 quad */
void wxo_delete_subtree(uint8_t *bt, int64_t len, int64_t i, int quad)
{
    bt[i - 1] = 0;
    if (!quad) {
        for (int c = 0; c < 2; c++) {
            int64_t ch = (i << 1) + c;
            if (ch <= len && bt[ch - 1]) wxo_delete_subtree(bt, len, ch, quad);
        }
    } else {
        for (int c = 0; c < 4; c++) {
            int64_t ch = 4 * i - 2 + c;
            if (ch <= len && bt[ch - 1]) wxo_delete_subtree(bt, len, ch, quad);
        }
    }
}

/* acwt/acwt_utils.jl:7-18 autocorr, :27-33 pfilter, :42-48 qfilter, :69-72
 * make_acreverseqmfpair -> (reverse(P), reverse(Q)), each 2F-1 long. */
void wxo_autocorr(const double *H, int l, double *result)
{
    for (int k = 1; k <= l - 1; k++) {
        result[k - 1] = 0.0;
        for (int i = 1; i <= l - k; i++) result[k - 1] += H[i - 1] * H[i + k - 1];
        result[k - 1] *= 2;
    }
}
void wxo_make_acreverseqmfpair(const double *qmf, int F, double *P, double *Q)
{
    double *a = (double *)malloc(sizeof(double) * (F > 1 ? F - 1 : 1));
    wxo_autocorr(qmf, F, a);
    double c1 = 1 / sqrt(2.0);
    double c2 = c1 / 2;
    int AL = 2 * F - 1;
    double *pf = (double *)malloc(sizeof(double) * AL), *qf = (double *)malloc(sizeof(double) * AL);
    for (int k = 0; k < F - 1; k++) {
        double b = c2 * a[k], bq = -c2 * a[k];
        pf[F - 2 - k] = b;  pf[F + k] = b;      /* vcat(reverse(b), c1, b) */
        qf[F - 2 - k] = bq; qf[F + k] = bq;
    }
    pf[F - 1] = c1; qf[F - 1] = c1;
    for (int i = 0; i < AL; i++) { P[i] = pf[AL - 1 - i]; Q[i] = qf[AL - 1 - i]; }
    free(a); free(pf); free(qf);
}

/* Utils.jl:351-371  coarsestscalingrange(n, tree, redundant).  Non-redundant: returns 1:(n>>j)
 * as hi; redundant: returns node index.  -1 on the @assert. */
int64_t wxo_coarsestscalingrange(int64_t n, const uint8_t *tree, int64_t nt, int redundant)
{
    int L = wxo_getdepth_binary(nt);
    if (!(L + 1 == wxo_maxtransformlevels(n))) return -1;
    int64_t i = 1; int j = 0;
    while (i < nt && tree[i - 1]) { i = i << 1; j++; }
    return redundant ? i : (n >> j);
}
/* Utils.jl:416-438  finestdetailrange: non-redundant returns lo of (n-n0+1):n; redundant node idx */
int64_t wxo_finestdetailrange(int64_t n, const uint8_t *tree, int64_t nt, int redundant)
{
    int L = wxo_getdepth_binary(nt);
    if (!(L + 1 == wxo_maxtransformlevels(n))) return -1;
    int64_t i = 1; int j = 0;
    while (i <= nt && tree[i - 1]) { i = (i << 1) + 1; j++; }
    return redundant ? i : (n - (n >> j) + 1);
}

#define WXO_T double
#define WXO_SUF _f64
#include "wx_oracle_impl.h"
#undef WXO_T
#undef WXO_SUF

#define WXO_T float
#define WXO_SUF _f32
#include "wx_oracle_impl.h"
#undef WXO_T
#undef WXO_SUF

/* batch drivers: dwt/dwt_all.jl:260-282 wpdall, :324-342 iwpdall, :152-166 wptall, :210-225
 * iwptall -- a loop over the last (batch) dimension.  Used as the timed CPU baseline
 * ("port", one thread, like the reference). */
void wxo_wpdall1d_f64(double *y, const double *x, int64_t n, int L, int64_t B, const double *qmf, int F)
{
    for (int64_t b = 0; b < B; b++) wxo_wpd1d_f64(y + b * n * (L + 1), x + b * n, n, L, qmf, F);
}
int wxo_iwpdall1d_f64(double *xh, const double *xw, int64_t n, int k, int64_t B, const uint8_t *tree,
                      int64_t ntree, const double *qmf, int F)
{
    int rc = 0;
    for (int64_t b = 0; b < B && rc == 0; b++)
        rc = wxo_iwpd1d_tree_f64(xh + b * n, xw + b * n * k, n, k, tree, ntree, qmf, F);
    return rc;
}
int wxo_wptall1d_f64(double *y, const double *x, int64_t n, int64_t B, const uint8_t *tree, int64_t ntree,
                     const double *qmf, int F)
{
    int rc = 0;
    for (int64_t b = 0; b < B && rc == 0; b++) rc = wxo_wpt1d_tree_f64(y + b * n, x + b * n, n, tree, ntree, qmf, F);
    return rc;
}
int wxo_iwptall1d_f64(double *y, const double *x, int64_t n, int64_t B, const uint8_t *tree, int64_t ntree,
                      const double *qmf, int F)
{
    int rc = 0;
    for (int64_t b = 0; b < B && rc == 0; b++) rc = wxo_iwpt1d_tree_f64(y + b * n, x + b * n, n, tree, ntree, qmf, F);
    return rc;
}

/* "generous" CPU baseline: the same per-signal restatement with OpenMP over the
 * batch on all host cores.  The reference itself has no threading anywhere. */
#ifdef _OPENMP
#include <omp.h>
int wxo_omp_max_threads(void) { return omp_get_max_threads(); }
#else
int wxo_omp_max_threads(void) { return 1; }
#endif
void wxo_wpd_iwpd_roundtrip_omp_f64(double *xh, double *y, const double *x, int64_t n, int L, int64_t B,
                                    const uint8_t *tree, int64_t ntree, const double *qmf, int F)
{
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; b++) {
        wxo_wpd1d_f64(y + b * n * (L + 1), x + b * n, n, L, qmf, F);
        wxo_iwpd1d_tree_f64(xh + b * n, y + b * n * (L + 1), n, L + 1, tree, ntree, qmf, F);
    }
}
/* first touch by the thread that will work on the signal (the same static schedule over the batch as the loops around this): dst = src */
void wxo_copy_omp_f64(double *dst, const double *src, int64_t n, int64_t B)
{
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; b++)
        for (int64_t i = 0; i < n; i++) dst[b * n + i] = src[b * n + i];
}
void wxo_wpt_iwpt_roundtrip_omp_f64(double *xh, double *y, const double *x, int64_t n, int64_t B,
                                    const uint8_t *tree, int64_t ntree, const double *qmf, int F)
{
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; b++) {
        wxo_wpt1d_tree_f64(y + b * n, x + b * n, n, tree, ntree, qmf, F);
        wxo_iwpt1d_tree_f64(xh + b * n, y + b * n, n, tree, ntree, qmf, F);
    }
}

/* sum(X, dims=3) and sum(X.^2, dims=3) of X = acwpdall(x) (acwt/acwt_all.jl:239-259 + bestbasis/bestbasis_tree.jl:153-154)
 * for batches whose (n, 2^(L+1)-1, N) table cannot be held: tables of a window of signals are made in parallel (each by the
 * per-signal restatement wxo_acwpd1d), then added signal by signal IN ORDER -- threads split the coefficient axis, never the
 * signal axis, so every EX[e] / EX2[e] sees the roundings of the sequential sums.  EX / EX2 are (n, 2^(L+1)-1), zeroed here. */
int wxo_acwpd_jbb_sums_f64(double *EX, double *EX2, const double *x, int64_t n, int L, int64_t N, const double *qmf, int F)
{
    const int64_t nl = n * (((int64_t)1 << (L + 1)) - 1);
    int W = wxo_omp_max_threads();
    if (W > 32) W = 32;
    if (W > N) W = (int)N;
    if (W < 1) W = 1;
    double *tab = (double *)malloc(sizeof(double) * nl * W);
    if (!tab) return -1;
    memset(EX, 0, sizeof(double) * nl); memset(EX2, 0, sizeof(double) * nl);
    int rc = 0;
    for (int64_t b0 = 0; b0 < N && rc == 0; b0 += W) {
        const int w = (int)(N - b0 < W ? N - b0 : W);
#pragma omp parallel for schedule(static)
        for (int s = 0; s < w; s++) {
            int r = wxo_acwpd1d_f64(tab + s * nl, x + (b0 + s) * n, n, L, qmf, F);
            if (r) {
#pragma omp atomic write
                rc = r;
            }
        }
        if (rc) break;
#pragma omp parallel for schedule(static)
        for (int64_t e = 0; e < nl; e++) {
            double a = EX[e], q = EX2[e];
            for (int s = 0; s < w; s++) {
                const double xv = tab[s * nl + e];
                a = a + xv;
                q = q + xv * xv;                 /* -ffp-contract=off: the square is rounded on its own, like X.^2 */
            }
            EX[e] = a; EX2[e] = q;
        }
    }
    free(tab);
    return rc;
}
