/*
 * wx_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (CPU oracle, not the product).
 *
 * Scalar restatement of the WaveletsExt.jl hot path, instantiated twice by
 * wx_oracle.c (T = double -> suffix _f64, T = float -> suffix _f32).  Every
 * function cites the reference file:line it follows (paths relative to
 * /root/reference/src/mod).  Loop nests and summation order are kept as in the
 * reference; Julia's 1-based indices are kept through the A1()/A2()/A3() macros
 * so each line can be read beside the Julia line it restates.
 *
 * Element-type rule: data is T, filters are always double;
 * `w[i] += g*v[k]` computes in double and rounds to T at every store.
 *
 * Parity status: pinned by the reference's own known-answer tests (3-digit db4
 * vectors, the 16-digit Haar doctest vector, every integer-helper literal); see
 * tests/test_oracle_kat.py.  JBB trees and Coiflet tables are "parity unpinned"
 * (the reference holds no golden values for them).
 */

#ifndef WXO_T
#error "define WXO_T, WXO_SUF before including"
#endif

#define WXO_CAT_(a, b) a##b
#define WXO_CAT(a, b) WXO_CAT_(a, b)
#define FN(name) WXO_CAT(name, WXO_SUF)
typedef WXO_T FN(wxo_elem);
#define T WXO_T

/* strided 1-based vector element */
#define V1(p, s, i) ((p)[(int64_t)((i) - 1) * (s)])

/* ------------------------------------------------------------------------- */
/* dwt/dwt_one_level.jl:79-107  dwt_step!(w1, w2, v, h, g), 1-D              */
/* ------------------------------------------------------------------------- */
static void FN(dwt_step_s)(T *w1, int64_t s1, T *w2, int64_t s2, const T *v, int64_t sv,
                           int64_t n, const double *h, const double *g, int filtlen)
{
    int64_t n1 = n / 2;
    for (int64_t i = 1; i <= n1; i++) {
        int64_t k1 = 2 * i - 1;
        int64_t k2 = 2 * i;
        V1(w1, s1, i) = (T)(g[filtlen - 1] * (double)V1(v, sv, k1));
        V1(w2, s2, i) = (T)(h[0] * (double)V1(v, sv, k2));
        for (int j = 2; j <= filtlen; j++) {
            k1 = k1 + 1; if (k1 > n) k1 = wxo_mod1(k1, n);
            k2 = k2 - 1; if (k2 <= 0) k2 = wxo_mod1(k2, n);
            V1(w1, s1, i) = (T)((double)V1(w1, s1, i) + g[filtlen - j] * (double)V1(v, sv, k1));
            V1(w2, s2, i) = (T)((double)V1(w2, s2, i) + h[j - 1] * (double)V1(v, sv, k2));
        }
    }
}

/* dwt/dwt_one_level.jl:192-223  idwt_step!(v, w1, w2, h, g), 1-D */
static void FN(idwt_step_s)(T *v, int64_t sv, const T *w1, int64_t s1, const T *w2, int64_t s2,
                            int64_t n, const double *h, const double *g, int filtlen)
{
    int64_t n1 = n / 2;
    for (int64_t i = 1; i <= n; i++) {
        int j0 = (int)wxo_mod1(i, 2);
        int j1 = filtlen - j0 + 1;
        int j2 = (int)wxo_mod1(i + 1, 2);
        int64_t k1 = (i + 1) >> 1;
        int64_t k2 = (i + 1) >> 1;
        V1(v, sv, i) = (T)(g[j1 - 1] * (double)V1(w1, s1, k1) + h[j2 - 1] * (double)V1(w2, s2, k2));
        for (int j = j0 + 2; j <= filtlen; j += 2) {
            j1 = filtlen - j + 1;
            j2 = j + (j & 1) - ((j & 1) == 0);
            k1 = k1 - 1; if (k1 <= 0) k1 = wxo_mod1(k1, n1);
            k2 = k2 + 1; if (k2 > n1) k2 = wxo_mod1(k2, n1);
            V1(v, sv, i) = (T)((double)V1(v, sv, i) +
                               (g[j1 - 1] * (double)V1(w1, s1, k1) + h[j2 - 1] * (double)V1(w2, s2, k2)));
        }
    }
}

/* 2-D views: column-major, element (r,c) 1-based at p[(r-1) + (c-1)*ld] */
#define M2(p, ld, r, c) ((p)[(int64_t)((r) - 1) + (int64_t)((c) - 1) * (ld)])

/* dwt/dwt_one_level.jl:319-354  dwt_step!(w1,w2,w3,w4,v,h,g,temp), 2-D.
 * w* are n x m views, v and temp are 2n x 2m views (each with its own ld). */
static void FN(dwt_step2)(T *w1, int64_t l1, T *w2, int64_t l2, T *w3, int64_t l3, T *w4, int64_t l4,
                          const T *v, int64_t lv, T *temp, int64_t lt, int64_t n, int64_t m,
                          const double *h, const double *g, int filtlen)
{
    for (int64_t j = 1; j <= 2 * m; j++)          /* all columns  :333-340 */
        FN(dwt_step_s)(&M2(temp, lt, 1, j), 1, &M2(temp, lt, n + 1, j), 1, &M2(v, lv, 1, j), 1,
                       2 * n, h, g, filtlen);
    for (int64_t i = 1; i <= n; i++) {             /* all rows     :342-352 */
        FN(dwt_step_s)(&M2(w1, l1, i, 1), l1, &M2(w2, l2, i, 1), l2, &M2(temp, lt, i, 1), lt,
                       2 * m, h, g, filtlen);
        FN(dwt_step_s)(&M2(w3, l3, i, 1), l3, &M2(w4, l4, i, 1), l4, &M2(temp, lt, n + i, 1), lt,
                       2 * m, h, g, filtlen);
    }
}

/* dwt/dwt_one_level.jl:401-436  idwt_step!(v,w1,w2,w3,w4,h,g,temp), 2-D */
static void FN(idwt_step2)(T *v, int64_t lv, const T *w1, int64_t l1, const T *w2, int64_t l2,
                           const T *w3, int64_t l3, const T *w4, int64_t l4, T *temp, int64_t lt,
                           int64_t n, int64_t m, const double *h, const double *g, int filtlen)
{
    for (int64_t i = 1; i <= n; i++) {             /* rows    :417-427 */
        FN(idwt_step_s)(&M2(temp, lt, i, 1), lt, &M2(w1, l1, i, 1), l1, &M2(w2, l2, i, 1), l2,
                        2 * m, h, g, filtlen);
        FN(idwt_step_s)(&M2(temp, lt, n + i, 1), lt, &M2(w3, l3, i, 1), l3, &M2(w4, l4, i, 1), l4,
                        2 * m, h, g, filtlen);
    }
    for (int64_t j = 1; j <= 2 * m; j++)          /* columns :429-434 */
        FN(idwt_step_s)(&M2(v, lv, 1, j), 1, &M2(temp, lt, 1, j), 1, &M2(temp, lt, n + 1, j), 1,
                        2 * n, h, g, filtlen);
}

/* exported single steps (contiguous) for the KAT tests */
void FN(wxo_dwt_step)(T *w1, T *w2, const T *v, int64_t n, const double *h, const double *g, int F)
{ FN(dwt_step_s)(w1, 1, w2, 1, v, 1, n, h, g, F); }
void FN(wxo_idwt_step)(T *v, const T *w1, const T *w2, int64_t n, const double *h, const double *g, int F)
{ FN(idwt_step_s)(v, 1, w1, 1, w2, 1, n, h, g, F); }
/* v is (2n x 2m); w1..w4 are (n x m), all dense column-major */
void FN(wxo_dwt_step2)(T *w1, T *w2, T *w3, T *w4, const T *v, int64_t n, int64_t m,
                       const double *h, const double *g, int F)
{
    T *temp = (T *)malloc(sizeof(T) * 4 * n * m);
    FN(dwt_step2)(w1, n, w2, n, w3, n, w4, n, v, 2 * n, temp, 2 * n, n, m, h, g, F);
    free(temp);
}
void FN(wxo_idwt_step2)(T *v, const T *w1, const T *w2, const T *w3, const T *w4, int64_t n, int64_t m,
                        const double *h, const double *g, int F)
{
    T *temp = (T *)malloc(sizeof(T) * 4 * n * m);
    FN(idwt_step2)(v, 2 * n, w1, n, w2, n, w3, n, w4, n, temp, 2 * n, n, m, h, g, F);
    free(temp);
}

/* ------------------------------------------------------------------------- */
/* DWT.jl:131-161  wpd!(y, x, wt, L), 1-D.  y is (n, L+1).                    */
/* ------------------------------------------------------------------------- */
void FN(wxo_wpd1d)(T *y, const T *x, int64_t n, int L, const double *qmf, int F)
{
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    for (int64_t i = 0; i < n; i++) y[i] = x[i];                     /* y[:,1] = x */
    for (int i = 0; i <= L - 1; i++) {
        int64_t np = n >> i;                                         /* nodelength(n,i) */
        for (int64_t j = 0; j <= ((int64_t)1 << i) - 1; j++) {
            int colp = i + 1;
            const T *v = &M2(y, n, j * np + 1, colp);
            int colr = colp + 1;
            int64_t nr = np / 2;
            T *w1 = &M2(y, n, 2 * j * nr + 1, colr);
            T *w2 = &M2(y, n, (2 * j + 1) * nr + 1, colr);
            FN(dwt_step_s)(w1, 1, w2, 1, v, 1, np, h, g, F);
        }
    }
    free(g); free(h);
}

/* DWT.jl:164-209  wpd!(y, x, wt, L), 2-D.  x is (m, n), y is (m, n, L+1).    */
void FN(wxo_wpd2d)(T *y, const T *x, int64_t m, int64_t n, int L, const double *qmf, int F)
{
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    T *temp = (T *)malloc(sizeof(T) * m * n);
    int64_t mn = m * n;
    for (int64_t i = 0; i < mn; i++) y[i] = x[i];
    for (int i = 0; i <= L - 1; i++) {
        int64_t mp = m >> i, np = n >> i;
        int64_t lr = ((int64_t)1 << i) - 1;
        for (int64_t j = 0; j <= lr; j++)
            for (int64_t k = 0; k <= lr; k++) {
                const T *ys = y + (int64_t)i * mn;        /* slice i+1   */
                T *yr = y + (int64_t)(i + 1) * mn;        /* slice i+2   */
                const T *v = &M2(ys, m, j * mp + 1, k * np + 1);
                int64_t mr = mp / 2, nr = np / 2;
                T *w1 = &M2(yr, m, 2 * j * mr + 1, 2 * k * nr + 1);
                T *w2 = &M2(yr, m, 2 * j * mr + 1, (2 * k + 1) * nr + 1);
                T *w3 = &M2(yr, m, (2 * j + 1) * mr + 1, 2 * k * nr + 1);
                T *w4 = &M2(yr, m, (2 * j + 1) * mr + 1, (2 * k + 1) * nr + 1);
                T *tk = &M2(temp, m, j * mp + 1, k * np + 1);
                FN(dwt_step2)(w1, m, w2, m, w3, m, w4, m, v, m, tk, m, mr, nr, h, g, F);
            }
    }
    free(temp); free(g); free(h);
}

/* Utils.jl:101-134  getbasiscoef(Xw, tree), N==2 branch (1-D signals).
 * Xw is (n, k); returns 0 ok, -2 = ArgumentError("Not enough decomposition levels"),
 * -1 = AssertionError. */
int FN(wxo_getbasiscoef1d)(T *xw, const T *Xw, int64_t n, int k, const uint8_t *tree, int64_t ntree)
{
    int L = wxo_maxtransformlevels(n);
    if (!wxo_isvalidtree1d(n, tree, ntree)) return -1;
    if (!(k - 1 <= L)) return -1;
    int64_t nleaf = ntree + ((int64_t)1 << wxo_getdepth_binary(ntree)) * 2; /* n + nt of getleaf */
    uint8_t *leaf = (uint8_t *)calloc(nleaf, 1);
    if (wxo_getleaf_binary(leaf, tree, ntree) != 0) { free(leaf); return -1; }
    int64_t leaf_len = ((int64_t)1 << (L + 1)) - 1;                  /* gettreelength(1<<(L+1)) */
    if (leaf_len != nleaf) { free(leaf); return -1; }
    for (int64_t i = 1; i <= nleaf; i++) {
        if (leaf[i - 1]) {
            int d = wxo_getdepth_binary(i);
            if (!(d < k)) { free(leaf); return -2; }
            int64_t nn = i - ((int64_t)1 << d);                      /* i-1<<d == i-(1<<d) */
            int64_t n0 = n >> d;
            for (int64_t r = nn * n0 + 1; r <= (nn + 1) * n0; r++) xw[r - 1] = M2(Xw, n, r, d + 1);
        }
    }
    free(leaf);
    return 0;
}

/* 1-D wpt / iwpt by tree.  Source: Wavelets.jl (un-vendored dependency, compat 0.9/0.10),
 * call sites dwt/dwt_all.jl:162,221 and DWT.jl:349.  Restated through the reference's own
 * pin test/transforms.jl:25-33: wpt(x,wt,tree) == getbasiscoef(wpd(x,wt),tree) and
 * iwpt == bottom-up idwt_step!.  Computed here top-down with dwt_step! in a ping-pong pair. */
int FN(wxo_wpt1d_tree)(T *y, const T *x, int64_t n, const uint8_t *tree, int64_t ntree,
                       const double *qmf, int F)
{
    if (!wxo_isvalidtree1d(n, tree, ntree)) return -1;
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    T *cur = (T *)malloc(sizeof(T) * n), *nxt = (T *)malloc(sizeof(T) * n);
    for (int64_t i = 0; i < n; i++) cur[i] = x[i];
    int Lmax = wxo_maxtransformlevels(n);
    for (int d = 0; d < Lmax; d++) {
        int64_t np = n >> d;
        for (int64_t j = 0; j < ((int64_t)1 << d); j++) {
            int64_t node = ((int64_t)1 << d) + j;                    /* heap index, 1-based */
            int alive = node <= ntree && tree[node - 1];
            if (alive)
                FN(dwt_step_s)(nxt + j * np, 1, nxt + j * np + np / 2, 1, cur + j * np, 1, np, h, g, F);
            else
                for (int64_t r = 0; r < np; r++) nxt[j * np + r] = cur[j * np + r];
        }
        T *t = cur; cur = nxt; nxt = t;
    }
    for (int64_t i = 0; i < n; i++) y[i] = cur[i];
    free(cur); free(nxt); free(g); free(h);
    return 0;
}

int FN(wxo_iwpt1d_tree)(T *xh, const T *xw, int64_t n, const uint8_t *tree, int64_t ntree,
                        const double *qmf, int F)
{
    if (!wxo_isvalidtree1d(n, tree, ntree)) return -1;
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    T *cur = (T *)malloc(sizeof(T) * n), *nxt = (T *)malloc(sizeof(T) * n);
    for (int64_t i = 0; i < n; i++) cur[i] = xw[i];
    int Lmax = wxo_maxtransformlevels(n);
    for (int d = Lmax - 1; d >= 0; d--) {
        int64_t np = n >> d;
        for (int64_t j = 0; j < ((int64_t)1 << d); j++) {
            int64_t node = ((int64_t)1 << d) + j;
            int alive = node <= ntree && tree[node - 1];
            if (alive)
                FN(idwt_step_s)(nxt + j * np, 1, cur + j * np, 1, cur + j * np + np / 2, 1, np, h, g, F);
            else
                for (int64_t r = 0; r < np; r++) nxt[j * np + r] = cur[j * np + r];
        }
        T *t = cur; cur = nxt; nxt = t;
    }
    for (int64_t i = 0; i < n; i++) xh[i] = cur[i];
    free(cur); free(nxt); free(g); free(h);
    return 0;
}

/* DWT.jl:340-351  iwpd!(x̂, xw, wt, tree), 1-D: getbasiscoef then iwpt!. */
int FN(wxo_iwpd1d_tree)(T *xh, const T *Xw, int64_t n, int k, const uint8_t *tree, int64_t ntree,
                        const double *qmf, int F)
{
    if (!wxo_isvalidtree1d(n, tree, ntree)) return -1;
    T *w = (T *)malloc(sizeof(T) * n);
    int rc = FN(wxo_getbasiscoef1d)(w, Xw, n, k, tree, ntree);
    if (rc == 0) rc = FN(wxo_iwpt1d_tree)(xh, w, n, tree, ntree, qmf, F);
    free(w);
    return rc;
}

/* copy a (rows x cols) block between column-major views */
static void FN(blkcpy)(T *dst, int64_t ld, const T *src, int64_t ls, int64_t rows, int64_t cols)
{
    for (int64_t c = 0; c < cols; c++)
        for (int64_t r = 0; r < rows; r++) dst[r + c * ld] = src[r + c * ls];
}

/* DWT.jl:500-548  wpt!(y, x, wt, tree), 2-D quad tree, in place with ping-pong y / yt. */
int FN(wxo_wpt2d_tree)(T *y, const T *x, int64_t m, int64_t n, const uint8_t *tree, int64_t ntree,
                       const double *qmf, int F)
{
    if (!wxo_isvalidtree2d(m, n, tree, ntree)) return -1;
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    int64_t mn = m * n;
    T *yt = (T *)malloc(sizeof(T) * mn), *temp = (T *)malloc(sizeof(T) * mn);
    for (int64_t i = 0; i < mn; i++) { y[i] = x[i]; yt[i] = x[i]; }
    for (int64_t i = 1; i <= ntree; i++) {
        if (!tree[i - 1]) continue;
        int64_t r0, r1, c0, c1, rr[4][2], cc[4][2];
        wxo_getrowrange(m, i, &r0, &r1); wxo_getcolrange(n, i, &c0, &c1);
        for (int c = 0; c < 4; c++) {
            wxo_getrowrange(m, 4 * i - 2 + c, &rr[c][0], &rr[c][1]);
            wxo_getcolrange(n, 4 * i - 2 + c, &cc[c][0], &cc[c][1]);
        }
        int64_t nr = rr[0][1] - rr[0][0] + 1, nc = cc[0][1] - cc[0][0] + 1;
        FN(dwt_step2)(&M2(y, m, rr[0][0], cc[0][0]), m, &M2(y, m, rr[1][0], cc[1][0]), m,
                      &M2(y, m, rr[2][0], cc[2][0]), m, &M2(y, m, rr[3][0], cc[3][0]), m,
                      &M2(yt, m, r0, c0), m, &M2(temp, m, r0, c0), m, nr, nc, h, g, F);
        if (4 * i < ntree)
            FN(blkcpy)(&M2(yt, m, r0, c0), m, &M2(y, m, r0, c0), m, r1 - r0 + 1, c1 - c0 + 1);
    }
    free(yt); free(temp); free(g); free(h);
    return 0;
}

/* DWT.jl:662-710  iwpt!(x̂, xw, wt, tree), 2-D */
int FN(wxo_iwpt2d_tree)(T *xh, const T *xw, int64_t m, int64_t n, const uint8_t *tree, int64_t ntree,
                        const double *qmf, int F)
{
    if (!wxo_isvalidtree2d(m, n, tree, ntree)) return -1;
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    int64_t mn = m * n;
    T *xwt = (T *)malloc(sizeof(T) * mn), *temp = (T *)malloc(sizeof(T) * mn);
    for (int64_t i = 0; i < mn; i++) { xh[i] = xw[i]; xwt[i] = xw[i]; }
    for (int64_t i = ntree; i >= 1; i--) {
        if (!tree[i - 1]) continue;
        int64_t r0, r1, c0, c1, rr[4][2], cc[4][2];
        wxo_getrowrange(m, i, &r0, &r1); wxo_getcolrange(n, i, &c0, &c1);
        for (int c = 0; c < 4; c++) {
            wxo_getrowrange(m, 4 * i - 2 + c, &rr[c][0], &rr[c][1]);
            wxo_getcolrange(n, 4 * i - 2 + c, &cc[c][0], &cc[c][1]);
        }
        int64_t nr = rr[0][1] - rr[0][0] + 1, nc = cc[0][1] - cc[0][0] + 1;
        FN(idwt_step2)(&M2(xh, m, r0, c0), m, &M2(xwt, m, rr[0][0], cc[0][0]), m,
                       &M2(xwt, m, rr[1][0], cc[1][0]), m, &M2(xwt, m, rr[2][0], cc[2][0]), m,
                       &M2(xwt, m, rr[3][0], cc[3][0]), m, &M2(temp, m, r0, c0), m, nr, nc, h, g, F);
        if (i > 1)
            FN(blkcpy)(&M2(xwt, m, r0, c0), m, &M2(xh, m, r0, c0), m, r1 - r0 + 1, c1 - c0 + 1);
    }
    free(xwt); free(temp); free(g); free(h);
    return 0;
}

/* DWT.jl:354-401  iwpd!(x̂, xw, wt, tree), 2-D; xw is (m, n, k) */
int FN(wxo_iwpd2d_tree)(T *xh, const T *xw, int64_t m, int64_t n, int k, const uint8_t *tree,
                        int64_t ntree, const double *qmf, int F)
{
    if (!wxo_isvalidtree2d(m, n, tree, ntree)) return -1;
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    int64_t mn = m * n;
    T *xwt = (T *)malloc(sizeof(T) * mn * k), *temp = (T *)malloc(sizeof(T) * mn);
    for (int64_t i = 0; i < mn * k; i++) xwt[i] = xw[i];
    int rc = 0;
    for (int64_t i = ntree; i >= 1; i--) {
        if (!tree[i - 1]) continue;
        int d = wxo_getdepth_quad(i);
        if (d + 2 > k) { rc = -3; break; }                           /* BoundsError in Julia */
        int64_t r0, r1, c0, c1, rr[4][2], cc[4][2];
        wxo_getrowrange(m, i, &r0, &r1); wxo_getcolrange(n, i, &c0, &c1);
        for (int c = 0; c < 4; c++) {
            wxo_getrowrange(m, 4 * i - 2 + c, &rr[c][0], &rr[c][1]);
            wxo_getcolrange(n, 4 * i - 2 + c, &cc[c][0], &cc[c][1]);
        }
        int64_t nr = rr[0][1] - rr[0][0] + 1, nc = cc[0][1] - cc[0][0] + 1;
        T *v = d == 0 ? xh : &M2(xwt + (int64_t)d * mn, m, r0, c0);
        const T *ch = xwt + (int64_t)(d + 1) * mn;
        FN(idwt_step2)(v, m, &M2(ch, m, rr[0][0], cc[0][0]), m, &M2(ch, m, rr[1][0], cc[1][0]), m,
                       &M2(ch, m, rr[2][0], cc[2][0]), m, &M2(ch, m, rr[3][0], cc[3][0]), m,
                       &M2(temp, m, r0, c0), m, nr, nc, h, g, F);
    }
    /* tree[1]==false: the reference returns x̂ untouched (undef); the oracle copies slice 1 */
    if (rc == 0 && !(ntree >= 1 && tree[0])) for (int64_t i = 0; i < mn; i++) xh[i] = xw[i];
    free(xwt); free(temp); free(g); free(h);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* swt/swt_one_level.jl:99-127  sdwt_step!(w1, w2, v, d, h, g)                */
/* ------------------------------------------------------------------------- */
static void FN(sdwt_step_s)(T *w1, int64_t s1, T *w2, int64_t s2, const T *v, int64_t sv, int64_t n,
                            int d, const double *h, const double *g, int filtlen)
{
    int64_t st = (int64_t)1 << d;
    for (int64_t i = 1; i <= n; i++) {
        int64_t k1 = wxo_mod1(i - st, n);
        int64_t k2 = i;
        V1(w1, s1, i) = (T)(g[filtlen - 1] * (double)V1(v, sv, k1));
        V1(w2, s2, i) = (T)(h[0] * (double)V1(v, sv, k2));
        for (int j = 2; j <= filtlen; j++) {
            k1 = k1 + st; if (k1 > n) k1 = wxo_mod1(k1, n);
            k2 = k2 - st; if (k2 <= 0) k2 = wxo_mod1(k2, n);
            V1(w1, s1, i) = (T)((double)V1(w1, s1, i) + g[filtlen - j] * (double)V1(v, sv, k1));
            V1(w2, s2, i) = (T)((double)V1(w2, s2, i) + h[j - 1] * (double)V1(v, sv, k2));
        }
    }
}

/* swt/swt_one_level.jl:279-318  isdwt_step!(v,w1,w2,d,sv,sw,h,g; add2out) shift-based.
 * returns -1 on the two @assert failures (:288-289) */
static int FN(isdwt_step_shift_s)(T *v, int64_t svs, const T *w1, int64_t s1, const T *w2, int64_t s2,
                                  int64_t n, int d, int64_t sv, int64_t sw, const double *h,
                                  const double *g, int filtlen, int add2out)
{
    if (!(0 <= sv && sv < ((int64_t)1 << d))) return -1;
    if (!(sv <= sw && sw < ((int64_t)1 << (d + 1)))) return -1;
    int64_t ip = sv + 1, sp = (int64_t)1 << d, ic = sw + 1, sc = (int64_t)1 << (d + 1);
    int64_t t = 0;
    for (int64_t m = ip; m <= n; m += sp) {
        t++;
        int i0 = (int)wxo_mod1(t, 2);
        int i1 = filtlen - i0 + 1;
        int i2 = (int)wxo_mod1(t + 1, 2);
        /* m-1<<d == m-(1<<d);  m+sp-1<<d == m+sp-(1<<d) */
        int64_t j = (sw == sv) ? wxo_mod1(m - ((int64_t)1 << d), n) : wxo_mod1(m + sp - ((int64_t)1 << d), n);
        int64_t k1 = ((t - 1) >> 1) * sc + ic;
        int64_t k2 = ((t - 1) >> 1) * sc + ic;
        double first = g[i1 - 1] * (double)V1(w1, s1, k1) + h[i2 - 1] * (double)V1(w2, s2, k2);
        V1(v, svs, j) = add2out ? (T)((double)V1(v, svs, j) + g[i1 - 1] * (double)V1(w1, s1, k1) +
                                      h[i2 - 1] * (double)V1(w2, s2, k2))
                                : (T)first;
        for (int i = i0 + 2; i <= filtlen; i += 2) {
            i1 = filtlen - i + 1;
            i2 = i + (i & 1) - ((i & 1) == 0);
            k1 = k1 - sc; if (k1 <= 0) k1 = wxo_mod1(k1, n);
            k2 = k2 + sc; if (k2 > n) k2 = wxo_mod1(k2, n);
            V1(v, svs, j) = (T)((double)V1(v, svs, j) +
                                (g[i1 - 1] * (double)V1(w1, s1, k1) + h[i2 - 1] * (double)V1(w2, s2, k2)));
        }
    }
    return 0;
}

/* swt/swt_one_level.jl:257-277  isdwt_step!(v,w1,w2,d,h,g) average-based */
static void FN(isdwt_step_avg_s)(T *v, int64_t svs, const T *w1, int64_t s1, const T *w2, int64_t s2,
                                 int64_t n, int d, const double *h, const double *g, int filtlen)
{
    int64_t nd = (int64_t)1 << d;
    for (int64_t sv = 0; sv <= nd - 1; sv++) {
        int64_t sw1 = sv;
        int64_t sw2 = sv + ((int64_t)1 << d);                         /* sv + 1<<d */
        FN(isdwt_step_shift_s)(v, svs, w1, s1, w2, s2, n, d, sv, sw1, h, g, filtlen, 0);
        FN(isdwt_step_shift_s)(v, svs, w1, s1, w2, s2, n, d, sv, sw2, h, g, filtlen, 1);
    }
    for (int64_t i = 1; i <= n; i++) V1(v, svs, i) = (T)(V1(v, svs, i) / 2);
}

void FN(wxo_sdwt_step)(T *w1, T *w2, const T *v, int64_t n, int d, const double *h, const double *g, int F)
{ FN(sdwt_step_s)(w1, 1, w2, 1, v, 1, n, d, h, g, F); }
int FN(wxo_isdwt_step_shift)(T *v, const T *w1, const T *w2, int64_t n, int d, int64_t sv, int64_t sw,
                             const double *h, const double *g, int F)
{ return FN(isdwt_step_shift_s)(v, 1, w1, 1, w2, 1, n, d, sv, sw, h, g, F, 0); }
void FN(wxo_isdwt_step_avg)(T *v, const T *w1, const T *w2, int64_t n, int d, const double *h,
                            const double *g, int F)
{ FN(isdwt_step_avg_s)(v, 1, w1, 1, w2, 1, n, d, h, g, F); }

/* swt/swt_one_level.jl:334-370  sdwt_step! 2-D; all arrays (n x m) dense, temp (n,m,2) */
static void FN(sdwt_step2)(T *w1, T *w2, T *w3, T *w4, const T *v, int64_t n, int64_t m, int d,
                           const double *h, const double *g, int F, T *temp)
{
    T *t1 = temp, *t2 = temp + n * m;
    for (int64_t j = 1; j <= m; j++)
        FN(sdwt_step_s)(&M2(t1, n, 1, j), 1, &M2(t2, n, 1, j), 1, &M2(v, n, 1, j), 1, n, d, h, g, F);
    for (int64_t i = 1; i <= n; i++) {
        FN(sdwt_step_s)(&M2(w1, n, i, 1), n, &M2(w2, n, i, 1), n, &M2(t1, n, i, 1), n, m, d, h, g, F);
        FN(sdwt_step_s)(&M2(w3, n, i, 1), n, &M2(w4, n, i, 1), n, &M2(t2, n, i, 1), n, m, d, h, g, F);
    }
}
/* swt/swt_one_level.jl:395-431 (average) and :433-469 (shift) isdwt_step! 2-D */
static int FN(isdwt_step2)(T *v, const T *w1, const T *w2, const T *w3, const T *w4, int64_t n, int64_t m,
                           int d, int use_shift, int64_t sv, int64_t sw, const double *h, const double *g,
                           int F, T *temp)
{
    T *t1 = temp, *t2 = temp + n * m;
    int rc = 0;
    for (int64_t i = 1; i <= n; i++) {
        if (use_shift) {
            rc |= FN(isdwt_step_shift_s)(&M2(t1, n, i, 1), n, &M2(w1, n, i, 1), n, &M2(w2, n, i, 1), n, m, d, sv, sw, h, g, F, 0);
            rc |= FN(isdwt_step_shift_s)(&M2(t2, n, i, 1), n, &M2(w3, n, i, 1), n, &M2(w4, n, i, 1), n, m, d, sv, sw, h, g, F, 0);
        } else {
            FN(isdwt_step_avg_s)(&M2(t1, n, i, 1), n, &M2(w1, n, i, 1), n, &M2(w2, n, i, 1), n, m, d, h, g, F);
            FN(isdwt_step_avg_s)(&M2(t2, n, i, 1), n, &M2(w3, n, i, 1), n, &M2(w4, n, i, 1), n, m, d, h, g, F);
        }
    }
    for (int64_t j = 1; j <= m; j++) {
        if (use_shift)
            rc |= FN(isdwt_step_shift_s)(&M2(v, n, 1, j), 1, &M2(t1, n, 1, j), 1, &M2(t2, n, 1, j), 1, n, d, sv, sw, h, g, F, 0);
        else
            FN(isdwt_step_avg_s)(&M2(v, n, 1, j), 1, &M2(t1, n, 1, j), 1, &M2(t2, n, 1, j), 1, n, d, h, g, F);
    }
    return rc ? -1 : 0;
}
void FN(wxo_sdwt_step2)(T *w1, T *w2, T *w3, T *w4, const T *v, int64_t n, int64_t m, int d,
                        const double *h, const double *g, int F)
{
    T *temp = (T *)malloc(sizeof(T) * 2 * n * m);
    FN(sdwt_step2)(w1, w2, w3, w4, v, n, m, d, h, g, F, temp);
    free(temp);
}
int FN(wxo_isdwt_step2)(T *v, const T *w1, const T *w2, const T *w3, const T *w4, int64_t n, int64_t m,
                        int d, int use_shift, int64_t sv, int64_t sw, const double *h, const double *g, int F)
{
    T *temp = (T *)calloc(2 * n * m, sizeof(T));
    int rc = FN(isdwt_step2)(v, w1, w2, w3, w4, n, m, d, use_shift, sv, sw, h, g, F, temp);
    free(temp);
    return rc;
}

/* SWT.jl:109-130  sdwt!(xw, x, wt, L) 1-D; xw is (n, L+1) = [s_L d_L ... d_1] */
int FN(wxo_sdwt1d)(T *xw, const T *x, int64_t n, int L, const double *qmf, int F)
{
    if (!(L <= wxo_maxtransformlevels(n)) || !(L >= 1)) return -1;
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    T *v = (T *)malloc(sizeof(T) * n);
    for (int64_t i = 0; i < n; i++) M2(xw, n, i + 1, L + 1) = x[i];     /* xw[:,end] = x */
    for (int d = 0; d <= L - 1; d++) {
        for (int64_t i = 0; i < n; i++) v[i] = M2(xw, n, i + 1, L - d + 1);
        FN(sdwt_step_s)(&M2(xw, n, 1, L - d), 1, &M2(xw, n, 1, L - d + 1), 1, v, 1, n, d, h, g, F);
    }
    free(v); free(g); free(h);
    return 0;
}

/* SWT.jl:259-284 (sm >= 0, shift based) and :313-330 (sm < 0, average based) isdwt! 1-D */
int FN(wxo_isdwt1d)(T *x, const T *xw, int64_t n, int L, int64_t sm, const double *qmf, int F)
{
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    int64_t *sd = NULL;
    int rc = 0;
    if (sm >= 0) {
        /* @assert 0 <= log2(sm) < L  : sm >= 1 and sm < 2^L */
        if (!(sm >= 1 && sm < ((int64_t)1 << L))) { free(g); free(h); return -1; }
        sd = (int64_t *)malloc(sizeof(int64_t) * (L + 1));
        wxo_main2depthshift(sm, L, sd);
    }
    T *w1 = (T *)malloc(sizeof(T) * n);
    for (int64_t i = 0; i < n; i++) x[i] = xw[i];                       /* x[i] = xw[i,1] */
    for (int d = L - 1; d >= 0; d--) {
        for (int64_t i = 0; i < n; i++) w1[i] = x[i];                   /* w1 = copy(x) */
        const T *w2 = &M2(xw, n, 1, L - d + 1);
        if (sm >= 0) rc |= FN(isdwt_step_shift_s)(x, 1, w1, 1, w2, 1, n, d, sd[d], sd[d + 1], h, g, F, 0);
        else FN(isdwt_step_avg_s)(x, 1, w1, 1, w2, 1, n, d, h, g, F);
    }
    free(w1); free(sd); free(g); free(h);
    return rc ? -1 : 0;
}

/* SWT.jl:439-472  swpt!(xw, x, wt, L) 1-D; xw is (n, 2^L) */
int FN(wxo_swpt1d)(T *xw, const T *x, int64_t n, int L, const double *qmf, int F)
{
    if (!(L <= wxo_maxtransformlevels(n)) || !(L >= 1)) return -1;
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    T *v = (T *)malloc(sizeof(T) * n);
    for (int64_t i = 0; i < n; i++) xw[i] = x[i];
    for (int d = 0; d <= L - 1; d++) {
        int64_t nn = (int64_t)1 << d;
        for (int64_t b = 0; b <= nn - 1; b++) {
            int64_t np = ((int64_t)1 << L) / nn, nc = np / 2;
            int64_t j1 = (2 * b) * nc + 1, j2 = (2 * b + 1) * nc + 1;
            for (int64_t i = 0; i < n; i++) v[i] = M2(xw, n, i + 1, j1);  /* v = xw[:,j1] (copy) */
            FN(sdwt_step_s)(&M2(xw, n, 1, j1), 1, &M2(xw, n, 1, j2), 1, v, 1, n, d, h, g, F);
        }
    }
    free(v); free(g); free(h);
    return 0;
}

/* SWT.jl:613-646 (shift, sm>=0) and :685-712 (average, sm<0)  iswpt! 1-D; xw is (n, m=2^L) */
int FN(wxo_iswpt1d)(T *x, const T *xw, int64_t n, int64_t m, int64_t sm, const double *qmf, int F)
{
    if (!wxo_isdyadic(m)) return -2;
    int L = wxo_ndyadicscales(m);
    if (!(L <= wxo_maxtransformlevels(n))) return -2;
    int64_t *sd = NULL;
    if (sm >= 0) {
        if (!(sm < ((int64_t)1 << L))) return -1;                       /* main2depthshift assert */
        sd = (int64_t *)malloc(sizeof(int64_t) * (L + 1));
        wxo_main2depthshift(sm, L, sd);
    }
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    T *temp = (T *)malloc(sizeof(T) * n * m), *w1 = (T *)malloc(sizeof(T) * n);
    for (int64_t i = 0; i < n * m; i++) temp[i] = xw[i];
    int rc = 0;
    if (L == 0) for (int64_t i = 0; i < n; i++) x[i] = xw[i];
    for (int d = L - 1; d >= 0; d--) {
        int64_t nn = (int64_t)1 << d;
        for (int64_t b = 0; b <= nn - 1; b++) {
            int64_t np = ((int64_t)1 << L) / nn, nc = np / 2;
            int64_t j1 = (2 * b) * nc + 1, j2 = (2 * b + 1) * nc + 1;
            T *v = d == 0 ? x : &M2(temp, n, 1, j1);
            for (int64_t i = 0; i < n; i++) w1[i] = M2(temp, n, i + 1, j1);
            const T *w2 = &M2(temp, n, 1, j2);
            if (sm >= 0) rc |= FN(isdwt_step_shift_s)(v, 1, w1, 1, w2, 1, n, d, sd[d], sd[d + 1], h, g, F, 0);
            else FN(isdwt_step_avg_s)(v, 1, w1, 1, w2, 1, n, d, h, g, F);
        }
    }
    free(temp); free(w1); free(sd); free(g); free(h);
    return rc ? -1 : 0;
}

/* SWT.jl:840-868  swpd!(xw, x, wt, L) 1-D; xw is (n, 2^(L+1)-1), heap order */
int FN(wxo_swpd1d)(T *xw, const T *x, int64_t n, int L, const double *qmf, int F)
{
    if (!(L <= wxo_maxtransformlevels(n)) || !(L >= 1)) return -1;
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    int64_t n0 = ((int64_t)1 << (L + 1)) - 1;
    int64_t n1 = n0 - ((int64_t)1 << L);
    for (int64_t i = 0; i < n; i++) xw[i] = x[i];
    for (int64_t i = 1; i <= n1; i++) {
        int d = wxo_getdepth_binary(i);
        int64_t j1 = i << 1, j2 = (i << 1) + 1;
        FN(sdwt_step_s)(&M2(xw, n, 1, j1), 1, &M2(xw, n, 1, j2), 1, &M2(xw, n, 1, i), 1, n, d, h, g, F);
    }
    free(g); free(h);
    return 0;
}

/* SWT.jl:1063-1093 (shift, sm>=0) and :1137-1160 (average, sm<0) iswpd!(x, xw, wt, tree[, sm]) 1-D;
 * xw is (n, m) heap ordered. */
int FN(wxo_iswpd1d)(T *x, const T *xw, int64_t n, int64_t m, const uint8_t *tree, int64_t ntree,
                    int64_t sm, const double *qmf, int F)
{
    if (!wxo_isvalidtree1d(n, tree, ntree)) return -1;
    int L = wxo_getdepth_binary(m);
    int64_t *sd = NULL;
    if (sm >= 0) {
        if (!(sm < ((int64_t)1 << L))) return -1;
        sd = (int64_t *)malloc(sizeof(int64_t) * (L + 1));
        wxo_main2depthshift(sm, L, sd);
    }
    double *g = (double *)malloc(sizeof(double) * F), *h = (double *)malloc(sizeof(double) * F);
    wxo_makereverseqmfpair(qmf, F, g, h);
    T *tmp = (T *)malloc(sizeof(T) * n * m);
    for (int64_t i = 0; i < n * m; i++) tmp[i] = xw[i];
    int rc = 0;
    for (int64_t i = ntree; i >= 1; i--) {
        if (!tree[i - 1]) continue;
        int d = wxo_getdepth_binary(i);
        int64_t j1 = i << 1, j2 = (i << 1) + 1;
        if (j2 > m) { rc = -3; break; }
        T *v = i == 1 ? x : &M2(tmp, n, 1, i);
        if (sm >= 0) rc |= FN(isdwt_step_shift_s)(v, 1, &M2(tmp, n, 1, j1), 1, &M2(tmp, n, 1, j2), 1, n, d, sd[d], sd[d + 1], h, g, F, 0);
        else FN(isdwt_step_avg_s)(v, 1, &M2(tmp, n, 1, j1), 1, &M2(tmp, n, 1, j2), 1, n, d, h, g, F);
    }
    if (!(ntree >= 1 && tree[0])) for (int64_t i = 0; i < n; i++) x[i] = xw[i];
    free(tmp); free(sd); free(g); free(h);
    return rc < 0 ? rc : 0;
}

/* ------------------------------------------------------------------------- */
/* acwt/acwt_one_level.jl:101-128  acdwt_step!(w1, w2, v, d, h, g)            */
/* (element type of filters == element type of data: Float64 only, App. D)   */
/* ------------------------------------------------------------------------- */
static void FN(acdwt_step_s)(T *w1, int64_t s1, T *w2, int64_t s2, const T *v, int64_t sv, int64_t N,
                             int d, const double *h, const double *g, int L)
{
    int64_t st = (int64_t)1 << d;
    for (int64_t i0 = 1; i0 <= N; i0++) {
        int64_t t = i0 + st; if (t > N) t = wxo_mod1(t, N);
        int64_t i = wxo_mod1(i0 + (int64_t)(L / 2 + 1) * st, N);
        V1(w1, s1, i) = (T)(g[0] * (double)V1(v, sv, t));
        V1(w2, s2, i) = (T)(h[0] * (double)V1(v, sv, t));
        for (int n = 2; n <= L; n++) {
            t = t + st; if (t > N) t = wxo_mod1(t, N);
            V1(w1, s1, i) = (T)((double)V1(w1, s1, i) + g[n - 1] * (double)V1(v, sv, t));
            V1(w2, s2, i) = (T)((double)V1(w2, s2, i) + h[n - 1] * (double)V1(v, sv, t));
        }
    }
}
/* acwt/acwt_one_level.jl:217-224  iacdwt_step!(v, w1, w2) */
static void FN(iacdwt_step_s)(T *v, int64_t sv, const T *w1, int64_t s1, const T *w2, int64_t s2, int64_t n)
{
    const double sqrt2 = sqrt(2.0);
    for (int64_t i = 1; i <= n; i++) V1(v, sv, i) = (T)(((double)V1(w1, s1, i) + (double)V1(w2, s2, i)) / sqrt2);
}
void FN(wxo_acdwt_step)(T *w1, T *w2, const T *v, int64_t n, int d, const double *h, const double *g, int L)
{ FN(acdwt_step_s)(w1, 1, w2, 1, v, 1, n, d, h, g, L); }
void FN(wxo_iacdwt_step)(T *v, const T *w1, const T *w2, int64_t n)
{ FN(iacdwt_step_s)(v, 1, w1, 1, w2, 1, n); }

/* acwt/acwt_one_level.jl:240-276  acdwt_step! 2-D */
void FN(wxo_acdwt_step2)(T *w1, T *w2, T *w3, T *w4, const T *v, int64_t n, int64_t m, int d,
                         const double *h, const double *g, int L)
{
    T *temp = (T *)malloc(sizeof(T) * 2 * n * m);
    T *t1 = temp, *t2 = temp + n * m;
    for (int64_t j = 1; j <= m; j++)
        FN(acdwt_step_s)(&M2(t1, n, 1, j), 1, &M2(t2, n, 1, j), 1, &M2(v, n, 1, j), 1, n, d, h, g, L);
    for (int64_t i = 1; i <= n; i++) {
        FN(acdwt_step_s)(&M2(w1, n, i, 1), n, &M2(w2, n, i, 1), n, &M2(t1, n, i, 1), n, m, d, h, g, L);
        FN(acdwt_step_s)(&M2(w3, n, i, 1), n, &M2(w4, n, i, 1), n, &M2(t2, n, i, 1), n, m, d, h, g, L);
    }
    free(temp);
}
/* acwt/acwt_one_level.jl:288-322  iacdwt_step! 2-D */
void FN(wxo_iacdwt_step2)(T *v, const T *w1, const T *w2, const T *w3, const T *w4, int64_t n, int64_t m)
{
    T *temp = (T *)malloc(sizeof(T) * 2 * n * m);
    T *t1 = temp, *t2 = temp + n * m;
    for (int64_t i = 1; i <= n; i++) {
        FN(iacdwt_step_s)(&M2(t1, n, i, 1), n, &M2(w1, n, i, 1), n, &M2(w2, n, i, 1), n, m);
        FN(iacdwt_step_s)(&M2(t2, n, i, 1), n, &M2(w3, n, i, 1), n, &M2(w4, n, i, 1), n, m);
    }
    for (int64_t j = 1; j <= m; j++)
        FN(iacdwt_step_s)(&M2(v, n, 1, j), 1, &M2(t1, n, 1, j), 1, &M2(t2, n, 1, j), 1, n);
    free(temp);
}

/* ACWT.jl:109-129  acdwt!(xw, x, wt, L) 1-D, xw (n, L+1); step called as (w1,w2,v,d,Qmf,Pmf) */
int FN(wxo_acdwt1d)(T *xw, const T *x, int64_t n, int L, const double *qmf, int F)
{
    if (!(L <= wxo_maxtransformlevels(n)) || !(L >= 1)) return -1;
    int AL = 2 * F - 1;
    double *P = (double *)malloc(sizeof(double) * AL), *Q = (double *)malloc(sizeof(double) * AL);
    wxo_make_acreverseqmfpair(qmf, F, P, Q);
    T *v = (T *)malloc(sizeof(T) * n);
    for (int64_t i = 0; i < n; i++) M2(xw, n, i + 1, L + 1) = x[i];
    for (int d = 0; d <= L - 1; d++) {
        for (int64_t i = 0; i < n; i++) v[i] = M2(xw, n, i + 1, L - d + 1);
        FN(acdwt_step_s)(&M2(xw, n, 1, L - d), 1, &M2(xw, n, 1, L - d + 1), 1, v, 1, n, d, Q, P, AL);
    }
    free(v); free(P); free(Q);
    return 0;
}
/* ACWT.jl:287-304  iacdwt!(x, xw) 1-D */
void FN(wxo_iacdwt1d)(T *x, const T *xw, int64_t n, int L)
{
    T *w1 = (T *)malloc(sizeof(T) * n);
    for (int64_t i = 0; i < n; i++) x[i] = xw[i];
    for (int d = L - 1; d >= 0; d--) {
        for (int64_t i = 0; i < n; i++) w1[i] = x[i];
        FN(iacdwt_step_s)(x, 1, w1, 1, &M2(xw, n, 1, L - d + 1), 1, n);
    }
    free(w1);
}
/* ACWT.jl:427-460  acwpt!(xw, x, wt, L) 1-D, xw (n, 2^L) */
int FN(wxo_acwpt1d)(T *xw, const T *x, int64_t n, int L, const double *qmf, int F)
{
    if (!(L <= wxo_maxtransformlevels(n)) || !(L >= 1)) return -1;
    int AL = 2 * F - 1;
    double *P = (double *)malloc(sizeof(double) * AL), *Q = (double *)malloc(sizeof(double) * AL);
    wxo_make_acreverseqmfpair(qmf, F, P, Q);
    T *v = (T *)malloc(sizeof(T) * n);
    for (int64_t i = 0; i < n; i++) xw[i] = x[i];
    for (int d = 0; d <= L - 1; d++) {
        int64_t nn = (int64_t)1 << d;
        for (int64_t b = 0; b <= nn - 1; b++) {
            int64_t np = ((int64_t)1 << L) / nn, nc = np / 2;
            int64_t j1 = (2 * b) * nc + 1, j2 = (2 * b + 1) * nc + 1;
            for (int64_t i = 0; i < n; i++) v[i] = M2(xw, n, i + 1, j1);
            FN(acdwt_step_s)(&M2(xw, n, 1, j1), 1, &M2(xw, n, 1, j2), 1, v, 1, n, d, Q, P, AL);
        }
    }
    free(v); free(P); free(Q);
    return 0;
}
/* ACWT.jl:581-610  iacwpt!(x, xw) 1-D, xw (n, m=2^L) */
int FN(wxo_iacwpt1d)(T *x, const T *xw, int64_t n, int64_t m)
{
    if (!wxo_isdyadic(m)) return -2;
    int L = wxo_ndyadicscales(m);
    if (!(L <= wxo_maxtransformlevels(n))) return -2;
    T *temp = (T *)malloc(sizeof(T) * n * m), *w1 = (T *)malloc(sizeof(T) * n);
    for (int64_t i = 0; i < n * m; i++) temp[i] = xw[i];
    if (L == 0) for (int64_t i = 0; i < n; i++) x[i] = xw[i];
    for (int d = L - 1; d >= 0; d--) {
        int64_t nn = (int64_t)1 << d;
        for (int64_t b = 0; b <= nn - 1; b++) {
            int64_t np = ((int64_t)1 << L) / nn, nc = np / 2;
            int64_t j1 = (2 * b) * nc + 1, j2 = (2 * b + 1) * nc + 1;
            T *v = d == 0 ? x : &M2(temp, n, 1, j1);
            for (int64_t i = 0; i < n; i++) w1[i] = M2(temp, n, i + 1, j1);
            FN(iacdwt_step_s)(v, 1, w1, 1, &M2(temp, n, 1, j2), 1, n);
        }
    }
    free(temp); free(w1);
    return 0;
}
/* ACWT.jl:733-759  acwpd!(xw, x, wt, L) 1-D, xw (n, 2^(L+1)-1) heap order */
int FN(wxo_acwpd1d)(T *xw, const T *x, int64_t n, int L, const double *qmf, int F)
{
    if (!(L <= wxo_maxtransformlevels(n)) || !(L >= 1)) return -1;
    int AL = 2 * F - 1;
    double *P = (double *)malloc(sizeof(double) * AL), *Q = (double *)malloc(sizeof(double) * AL);
    wxo_make_acreverseqmfpair(qmf, F, P, Q);
    int64_t n0 = ((int64_t)1 << (L + 1)) - 1, n1 = n0 - ((int64_t)1 << L);
    for (int64_t i = 0; i < n; i++) xw[i] = x[i];
    for (int64_t i = 1; i <= n1; i++) {
        int d = wxo_getdepth_binary(i);
        FN(acdwt_step_s)(&M2(xw, n, 1, i << 1), 1, &M2(xw, n, 1, (i << 1) + 1), 1, &M2(xw, n, 1, i), 1,
                         n, d, Q, P, AL);
    }
    free(P); free(Q);
    return 0;
}
/* ACWT.jl:944-968  iacwpd!(x, xw, tree) 1-D */
int FN(wxo_iacwpd1d)(T *x, const T *xw, int64_t n, int64_t m, const uint8_t *tree, int64_t ntree)
{
    if (!wxo_isvalidtree1d(n, tree, ntree)) return -1;
    T *tmp = (T *)malloc(sizeof(T) * n * m);
    for (int64_t i = 0; i < n * m; i++) tmp[i] = xw[i];
    int rc = 0;
    for (int64_t i = ntree; i >= 1; i--) {
        if (!tree[i - 1]) continue;
        int64_t j1 = i << 1, j2 = (i << 1) + 1;
        if (j2 > m) { rc = -3; break; }
        T *v = i == 1 ? x : &M2(tmp, n, 1, i);
        FN(iacdwt_step_s)(v, 1, &M2(tmp, n, 1, j1), 1, &M2(tmp, n, 1, j2), 1, n);
    }
    if (!(ntree >= 1 && tree[0])) for (int64_t i = 0; i < n; i++) x[i] = xw[i];
    free(tmp);
    return rc;
}

/* bestbasis_costs.jl:127-132  coefcost(x, ::LoglpCost) = p*sum(log.(abs.(x)));
 * coefcost(x, ::NormCost) = norm(x,p)^p.  Plain left-to-right sums (Julia's own sum is a
 * pairwise/SIMD reassociation with no fixed order; agreement ~1e-13 relative). */
static T FN(wxo_coefcost_jbb)(const T *x, int64_t n, int cost_kind, double p)
{
    if (cost_kind == 0) {
        T s = 0;
        for (int64_t i = 0; i < n; i++) s = (T)(s + (T)log((double)(T)fabs((double)x[i])));
        return (T)(p * (double)s);
    }
    double s = 0;
    for (int64_t i = 0; i < n; i++) s += pow(fabs((double)x[i]), p);
    return (T)s;
}

/* ------------------------------------------------------------------------- */
/* 2-D redundant drivers.  All images are (n, m) dense column-major, slices are */
/* n*m apart.  AC variants use acdwt_step!/iacdwt_step! 2-D.                    */
/* ------------------------------------------------------------------------- */
static void FN(acdwt_step2)(T *w1, T *w2, T *w3, T *w4, const T *v, int64_t n, int64_t m, int d,
                            const double *h, const double *g, int AL, T *temp)
{
    T *t1 = temp, *t2 = temp + n * m;
    for (int64_t j = 1; j <= m; j++)
        FN(acdwt_step_s)(&M2(t1, n, 1, j), 1, &M2(t2, n, 1, j), 1, &M2(v, n, 1, j), 1, n, d, h, g, AL);
    for (int64_t i = 1; i <= n; i++) {
        FN(acdwt_step_s)(&M2(w1, n, i, 1), n, &M2(w2, n, i, 1), n, &M2(t1, n, i, 1), n, m, d, h, g, AL);
        FN(acdwt_step_s)(&M2(w3, n, i, 1), n, &M2(w4, n, i, 1), n, &M2(t2, n, i, 1), n, m, d, h, g, AL);
    }
}
static void FN(iacdwt_step2)(T *v, const T *w1, const T *w2, const T *w3, const T *w4, int64_t n, int64_t m, T *temp)
{
    T *t1 = temp, *t2 = temp + n * m;
    for (int64_t i = 1; i <= n; i++) {
        FN(iacdwt_step_s)(&M2(t1, n, i, 1), n, &M2(w1, n, i, 1), n, &M2(w2, n, i, 1), n, m);
        FN(iacdwt_step_s)(&M2(t2, n, i, 1), n, &M2(w3, n, i, 1), n, &M2(w4, n, i, 1), n, m);
    }
    for (int64_t j = 1; j <= m; j++)
        FN(iacdwt_step_s)(&M2(v, n, 1, j), 1, &M2(t1, n, 1, j), 1, &M2(t2, n, 1, j), 1, n);
}

/* one forward step on slices: ac selects acdwt_step! (filters P,Q) instead of sdwt_step! (g,h) */
static void FN(red_fwd2)(T *w1, T *w2, T *w3, T *w4, const T *v, int64_t n, int64_t m, int d, int ac,
                         const double *f1, const double *f2, int flen, T *temp)
{
    if (ac) FN(acdwt_step2)(w1, w2, w3, w4, v, n, m, d, f1, f2, flen, temp);     /* (h=Q, g=P) */
    else FN(sdwt_step2)(w1, w2, w3, w4, v, n, m, d, f1, f2, flen, temp);         /* (h, g) */
}

/* SWT.jl:132-158 sdwt! 2-D / ACWT.jl:131-157 acdwt! 2-D; xw is (n, m, 3L+1) */
int FN(wxo_red_dwt2d)(T *xw, const T *x, int64_t n, int64_t m, int L, int ac, const double *qmf, int F)
{
    int Lmax = wxo_maxtransformlevels(n < m ? n : m);
    if (!(L <= Lmax) || !(L >= 1)) return -2;
    int flen = ac ? 2 * F - 1 : F;
    double *f1 = (double *)malloc(sizeof(double) * flen), *f2 = (double *)malloc(sizeof(double) * flen);
    if (ac) { double *P = f2, *Q = f1; wxo_make_acreverseqmfpair(qmf, F, P, Q); }   /* g=P -> f2, h=Q -> f1 */
    else { double *g = f2, *h = f1; wxo_makereverseqmfpair(qmf, F, g, h); }
    int64_t nm = n * m;
    T *temp = (T *)malloc(sizeof(T) * 2 * nm), *v = (T *)malloc(sizeof(T) * nm);
    for (int64_t i = 0; i < nm; i++) xw[(int64_t)(3 * L) * nm + i] = x[i];          /* xw[:,:,end] = x */
    for (int d = 0; d <= L - 1; d++) {
        for (int64_t i = 0; i < nm; i++) v[i] = xw[(int64_t)(3 * (L - d)) * nm + i]; /* slice 3(L-d)+1 */
        FN(red_fwd2)(xw + (int64_t)(3 * (L - d) - 3) * nm, xw + (int64_t)(3 * (L - d) - 2) * nm,
                     xw + (int64_t)(3 * (L - d) - 1) * nm, xw + (int64_t)(3 * (L - d)) * nm, v, n, m, d, ac,
                     f1, f2, flen, temp);
    }
    free(temp); free(v); free(f1); free(f2);
    return 0;
}

/* SWT.jl:286-311 (sm >= 0), :332-358 (sm < 0) isdwt! 2-D; ACWT.jl:306-329 iacdwt! 2-D (ac) */
int FN(wxo_ired_dwt2d)(T *x, const T *xw, int64_t n, int64_t m, int k, int ac, int64_t sm, const double *qmf, int F)
{
    int L = (k - 1) / 3;
    int64_t nm = n * m;
    int64_t *sd = NULL;
    double *g = NULL, *h = NULL;
    if (!ac) {
        if (sm >= 0) {
            /* @assert 0 <= log2(sm) <= L (:293) then main2depthshift asserts sm < 2^L */
            if (!(sm >= 1 && sm <= ((int64_t)1 << L))) return -1;
            if (!(sm < ((int64_t)1 << L))) return -1;
            sd = (int64_t *)malloc(sizeof(int64_t) * (L + 1));
            wxo_main2depthshift(sm, L, sd);
        }
        g = (double *)malloc(sizeof(double) * F); h = (double *)malloc(sizeof(double) * F);
        wxo_makereverseqmfpair(qmf, F, g, h);
    }
    T *temp = (T *)calloc(2 * nm, sizeof(T)), *w1 = (T *)malloc(sizeof(T) * nm);
    int rc = 0;
    for (int64_t i = 0; i < nm; i++) x[i] = xw[i];
    for (int d = L - 1; d >= 0; d--) {
        for (int64_t i = 0; i < nm; i++) w1[i] = x[i];
        const T *w2 = xw + (int64_t)(3 * (L - d) - 2) * nm, *w3 = xw + (int64_t)(3 * (L - d) - 1) * nm,
                *w4 = xw + (int64_t)(3 * (L - d)) * nm;
        if (ac) FN(iacdwt_step2)(x, w1, w2, w3, w4, n, m, temp);
        else rc |= FN(isdwt_step2)(x, w1, w2, w3, w4, n, m, d, sm >= 0, sm >= 0 ? sd[d] : 0, sm >= 0 ? sd[d + 1] : 0, h, g, F, temp);
    }
    free(temp); free(w1); free(sd); free(g); free(h);
    return rc ? -1 : 0;
}

/* SWT.jl:474-513 swpt! 2-D / ACWT.jl:462-501 acwpt! 2-D; xw is (n, m, 4^L) */
int FN(wxo_red_wpt2d)(T *xw, const T *x, int64_t n, int64_t m, int L, int ac, const double *qmf, int F)
{
    int Lmax = wxo_maxtransformlevels(n < m ? n : m);
    if (!(L <= Lmax) || !(L >= 1)) return -2;
    int flen = ac ? 2 * F - 1 : F;
    double *f1 = (double *)malloc(sizeof(double) * flen), *f2 = (double *)malloc(sizeof(double) * flen);
    if (ac) wxo_make_acreverseqmfpair(qmf, F, f2, f1); else wxo_makereverseqmfpair(qmf, F, f2, f1);
    int64_t nm = n * m;
    T *temp = (T *)malloc(sizeof(T) * 2 * nm), *v = (T *)malloc(sizeof(T) * nm);
    for (int64_t i = 0; i < nm; i++) xw[i] = x[i];
    int64_t tot = (int64_t)1 << (2 * L);
    for (int d = 0; d <= L - 1; d++) {
        int64_t nn = (int64_t)1 << (2 * d);
        for (int64_t b = 0; b <= nn - 1; b++) {
            int64_t np = tot / nn, nc = np / 4;
            int64_t j1 = (4 * b) * nc, j2 = (4 * b + 1) * nc, j3 = (4 * b + 2) * nc, j4 = (4 * b + 3) * nc;   /* 0-based */
            for (int64_t i = 0; i < nm; i++) v[i] = xw[j1 * nm + i];
            FN(red_fwd2)(xw + j1 * nm, xw + j2 * nm, xw + j3 * nm, xw + j4 * nm, v, n, m, d, ac, f1, f2, flen, temp);
        }
    }
    free(temp); free(v); free(f1); free(f2);
    return 0;
}

/* SWT.jl:648-683 (shift), :714-758 (average) iswpt! 2-D; ACWT.jl:612-648 iacwpt! 2-D; xw (n, m, k = 4^L) */
int FN(wxo_ired_wpt2d)(T *x, const T *xw, int64_t n, int64_t m, int64_t k, int ac, int64_t sm, const double *qmf, int F)
{
    int L = 0;
    while (((int64_t)1 << (2 * (L + 1))) <= k) L++;
    if (((int64_t)1 << (2 * L)) != k) return -2;                          /* not a power of 4 */
    if (!(L <= wxo_maxtransformlevels(n < m ? n : m))) return -1;
    int64_t nm = n * m;
    int64_t *sd = NULL;
    double *g = NULL, *h = NULL;
    if (!ac) {
        if (sm >= 0) {
            if (!(sm < ((int64_t)1 << L))) return -1;
            sd = (int64_t *)malloc(sizeof(int64_t) * (L + 1));
            wxo_main2depthshift(sm, L, sd);
        }
        g = (double *)malloc(sizeof(double) * F); h = (double *)malloc(sizeof(double) * F);
        wxo_makereverseqmfpair(qmf, F, g, h);
    }
    T *xwt = (T *)malloc(sizeof(T) * nm * k), *temp = (T *)calloc(2 * nm, sizeof(T)), *w1 = (T *)malloc(sizeof(T) * nm);
    for (int64_t i = 0; i < nm * k; i++) xwt[i] = xw[i];
    int rc = 0;
    if (L == 0) for (int64_t i = 0; i < nm; i++) x[i] = xw[i];
    for (int d = L - 1; d >= 0; d--) {
        int64_t nn = (int64_t)1 << (2 * d);
        for (int64_t b = 0; b <= nn - 1; b++) {
            int64_t np = k / nn, nc = np / 4;
            int64_t j1 = (4 * b) * nc, j2 = (4 * b + 1) * nc, j3 = (4 * b + 2) * nc, j4 = (4 * b + 3) * nc;
            T *v = d == 0 ? x : xwt + j1 * nm;
            for (int64_t i = 0; i < nm; i++) w1[i] = xwt[j1 * nm + i];
            if (ac) FN(iacdwt_step2)(v, w1, xwt + j2 * nm, xwt + j3 * nm, xwt + j4 * nm, n, m, temp);
            else rc |= FN(isdwt_step2)(v, w1, xwt + j2 * nm, xwt + j3 * nm, xwt + j4 * nm, n, m, d, sm >= 0,
                                       sm >= 0 ? sd[d] : 0, sm >= 0 ? sd[d + 1] : 0, h, g, F, temp);
        }
    }
    free(xwt); free(temp); free(w1); free(sd); free(g); free(h);
    return rc ? -1 : 0;
}

/* SWT.jl:870-902 swpd! 2-D / ACWT.jl:761-793 acwpd! 2-D; xw is (n, m, (4^(L+1)-1)/3), quad heap order */
int FN(wxo_red_wpd2d)(T *xw, const T *x, int64_t n, int64_t m, int L, int ac, const double *qmf, int F)
{
    int flen = ac ? 2 * F - 1 : F;
    double *f1 = (double *)malloc(sizeof(double) * flen), *f2 = (double *)malloc(sizeof(double) * flen);
    if (ac) wxo_make_acreverseqmfpair(qmf, F, f2, f1); else wxo_makereverseqmfpair(qmf, F, f2, f1);
    int64_t nm = n * m;
    int64_t k = ((((int64_t)1 << (2 * (L + 1))) - 1) / 3);
    int64_t n1 = k - ((int64_t)1 << (2 * L));
    T *temp = (T *)malloc(sizeof(T) * 2 * nm);
    for (int64_t i = 0; i < nm; i++) xw[i] = x[i];
    for (int64_t i = 1; i <= n1; i++) {
        int d = wxo_getdepth_quad(i);
        FN(red_fwd2)(xw + (4 * i - 2 - 1) * nm, xw + (4 * i - 1 - 1) * nm, xw + (4 * i - 1) * nm, xw + (4 * i + 1 - 1) * nm,
                     xw + (i - 1) * nm, n, m, d, ac, f1, f2, flen, temp);
    }
    free(temp); free(f1); free(f2);
    return 0;
}

/* SWT.jl:1095-1129 (shift), :1162-1199 (average) iswpd! 2-D; ACWT.jl:970-1000 iacwpd! 2-D */
int FN(wxo_ired_wpd2d)(T *x, const T *xw, int64_t n, int64_t m, int64_t k, const uint8_t *tree, int64_t ntree,
                       int ac, int64_t sm, const double *qmf, int F)
{
    if (!wxo_isvalidtree2d(n, m, tree, ntree)) return -1;
    int64_t nm = n * m;
    int L = wxo_getdepth_quad(k);
    int64_t *sd = NULL;
    double *g = NULL, *h = NULL;
    if (!ac) {
        if (sm >= 0) {
            if (!(sm < ((int64_t)1 << L))) return -1;
            sd = (int64_t *)malloc(sizeof(int64_t) * (L + 1));
            wxo_main2depthshift(sm, L, sd);
        }
        g = (double *)malloc(sizeof(double) * F); h = (double *)malloc(sizeof(double) * F);
        wxo_makereverseqmfpair(qmf, F, g, h);
    }
    T *xwt = (T *)malloc(sizeof(T) * nm * k), *temp = (T *)calloc(2 * nm, sizeof(T));
    for (int64_t i = 0; i < nm * k; i++) xwt[i] = xw[i];
    int rc = 0;
    for (int64_t i = ntree; i >= 1; i--) {
        if (!tree[i - 1]) continue;
        int d = wxo_getdepth_quad(i);
        if (4 * i + 1 > k) { rc = -3; break; }
        T *v = i == 1 ? x : xwt + (i - 1) * nm;
        const T *c1 = xwt + (4 * i - 2 - 1) * nm, *c2 = xwt + (4 * i - 1 - 1) * nm, *c3 = xwt + (4 * i - 1) * nm,
                *c4 = xwt + (4 * i + 1 - 1) * nm;
        if (ac) FN(iacdwt_step2)(v, c1, c2, c3, c4, n, m, temp);
        else rc |= FN(isdwt_step2)(v, c1, c2, c3, c4, n, m, d, sm >= 0, sm >= 0 ? sd[d] : 0, sm >= 0 ? sd[d + 1] : 0,
                                   h, g, F, temp);
    }
    if (!(ntree >= 1 && tree[0])) for (int64_t i = 0; i < nm; i++) x[i] = xw[i];
    free(xwt); free(temp); free(sd); free(g); free(h);
    return rc < 0 ? rc : (rc ? -1 : 0);
}

/* Utils.jl:101-134  getbasiscoef(Xw, tree), N==3 branch (2-D signals); Xw is (n, m, k) */
int FN(wxo_getbasiscoef2d)(T *xw, const T *Xw, int64_t n, int64_t m, int k, const uint8_t *tree, int64_t ntree)
{
    int L = wxo_maxtransformlevels(n < m ? n : m);
    if (!wxo_isvalidtree2d(n, m, tree, ntree)) return -1;
    if (!(k - 1 <= L)) return -1;
    int64_t nleaf = 4 * ntree + 1;
    uint8_t *leaf = (uint8_t *)calloc(nleaf, 1);
    if (wxo_getleaf_quad(leaf, tree, ntree) != 0) { free(leaf); return -1; }
    int64_t leaf_len = ((((int64_t)1 << (2 * (L + 1))) - 1) / 3);               /* gettreelength(2^(L+1), 2^(L+1)) */
    if (leaf_len != nleaf) { free(leaf); return -1; }
    for (int64_t i = 1; i <= nleaf; i++) {
        if (!leaf[i - 1]) continue;
        int d = wxo_getdepth_quad(i);
        if (!(d < k)) { free(leaf); return -2; }
        int64_t r0, r1, c0, c1;
        wxo_getrowrange(n, i, &r0, &r1); wxo_getcolrange(m, i, &c0, &c1);
        for (int64_t c = c0; c <= c1; c++)
            for (int64_t r = r0; r <= r1; r++) M2(xw, n, r, c) = M2(Xw + (int64_t)d * n * m, n, r, c);
    }
    free(leaf);
    return 0;
}

/* bestbasis/bestbasis_tree.jl:182-207  tree_costs(X::Array{T,4}, ::JBB); X is (n, m, L, N) */
int FN(wxo_tree_costs_jbb2d)(T *costs, const T *X, int64_t n, int64_t m, int64_t L, int64_t N, int redundant,
                             int cost_kind, double p)
{
    int64_t nl = n * m * L;
    T *EX = (T *)calloc(nl, sizeof(T)), *EX2 = (T *)calloc(nl, sizeof(T)), *sig = (T *)malloc(sizeof(T) * nl);
    for (int64_t s = 0; s < N; s++)
        for (int64_t e = 0; e < nl; e++) {
            T xv = X[s * nl + e];
            EX[e] = (T)(EX[e] + xv);
            EX2[e] = (T)(EX2[e] + (T)(xv * xv));
        }
    int bad = 0;
    for (int64_t e = 0; e < nl; e++) {
        T ex = (T)(EX[e] / (T)N), ex2 = (T)(EX2[e] / (T)N);
        sig[e] = (T)sqrt((double)(T)(ex2 - (T)(ex * ex)));
        if (!(sig[e] >= 0)) bad = 1;
    }
    if (bad) { free(EX); free(EX2); free(sig); return -1; }
    if (redundant) {
        for (int64_t i = 1; i <= L; i++) {
            int d = wxo_getdepth_quad(i);
            costs[i - 1] = (T)(FN(wxo_coefcost_jbb)(sig + (i - 1) * n * m, n * m, cost_kind, p) / (T)((int64_t)1 << (2 * d)));
        }
    } else {
        int64_t nc = ((((int64_t)1 << (2 * L)) - 1) / 3);                       /* gettreelength(1<<L, 1<<L) */
        T *blk = (T *)malloc(sizeof(T) * n * m);
        for (int64_t i = 1; i <= nc; i++) {
            int d = wxo_getdepth_quad(i);
            int64_t r0, r1, c0, c1;
            wxo_getrowrange(n, i, &r0, &r1); wxo_getcolrange(m, i, &c0, &c1);
            int64_t cnt = 0;
            for (int64_t c = c0; c <= c1; c++)
                for (int64_t r = r0; r <= r1; r++) blk[cnt++] = M2(sig + (int64_t)d * n * m, n, r, c);
            costs[i - 1] = FN(wxo_coefcost_jbb)(blk, cnt, cost_kind, p);
        }
        free(blk);
    }
    free(EX); free(EX2); free(sig);
    return 0;
}

/* BestBasis.jl:85-110  bestbasis_treeselection(costs, n, m, type) (quad tree) */
int FN(wxo_bestbasis_treeselection2d)(uint8_t *tree, T *costs, int64_t k, int64_t n, int64_t m, int type_max)
{
    if (!(k <= wxo_gettreelength2d(2 * n, 2 * m))) return -1;
    int L = wxo_getdepth_quad(k);
    int64_t ntree = wxo_gettreelength2d(n, m);
    if (wxo_maketree2d(tree, n, m, L, 0) != 0) return -1;
    for (int64_t i = ntree; i >= 1; i--) {
        if (tree[i - 1]) {
            T pc = costs[i - 1];
            T cc = (T)((T)((T)(costs[4 * i - 2 - 1] + costs[4 * i - 1 - 1]) + costs[4 * i - 1]) + costs[4 * i + 1 - 1]);
            if (!type_max && cc < pc) costs[i - 1] = cc;
            else if (type_max && cc > pc) costs[i - 1] = cc;
            else wxo_delete_subtree(tree, ntree, i, 1);
        }
    }
    return wxo_isvalidtree2d(n, m, tree, ntree) ? 0 : -1;
}

/* ------------------------------------------------------------------------- */
/* bestbasis/bestbasis_tree.jl:150-180  tree_costs(X::Array{T,3}, ::JBB)      */
/* X is (n, L, N).  cost_kind 0 = LoglpCost(p), 1 = NormCost(p)               */
/* (bestbasis_costs.jl:127-132).  costs must hold L (redundant) or 2^L-1      */
/* entries.  Returns -1 if the @assert all(sigma .>= 0) fails (NaN sigma).    */
/* ------------------------------------------------------------------------- */
/* the part of tree_costs after the two sums over the signal axis (bestbasis_tree.jl:155-179): EX / EX2 hold
 * sum(X, dims=3) and sum(X.^2, dims=3) and are overwritten.  Split out so that a test can accumulate the sums signal by
 * signal (same order, same roundings) without holding the (n, L, N) table. */
int FN(wxo_tree_costs_jbb_sums)(T *costs, T *EX, T *EX2, int64_t n, int64_t L, int64_t N, int redundant,
                                int cost_kind, double p)
{
    int64_t nl = n * L;
    T *sig = (T *)malloc(sizeof(T) * nl);
    int bad = 0;
    for (int64_t e = 0; e < nl; e++) {
        T ex = (T)(EX[e] / (T)N), ex2 = (T)(EX2[e] / (T)N);
        T var = (T)(ex2 - (T)(ex * ex));
        sig[e] = (T)sqrt((double)var);           /* VarX .^ 0.5 (DomainError if var<0 in Julia) */
        if (!(sig[e] >= 0)) bad = 1;
    }
    if (bad) { free(sig); return -1; }
    if (redundant) {
        for (int64_t i = 1; i <= L; i++) {
            int j = wxo_getdepth_binary(i);
            costs[i - 1] = (T)(FN(wxo_coefcost_jbb)(sig + (i - 1) * n, n, cost_kind, p) / (T)((int64_t)1 << j));
        }
    } else {
        int64_t i = 1;
        for (int64_t lvl = 0; lvl <= L - 1; lvl++) {
            int64_t n0 = n >> lvl;
            for (int64_t node = 0; node <= ((int64_t)1 << lvl) - 1; node++) {
                costs[i - 1] = FN(wxo_coefcost_jbb)(sig + lvl * n + node * n0, n0, cost_kind, p);
                i++;
            }
        }
    }
    free(sig);
    return 0;
}

int FN(wxo_tree_costs_jbb)(T *costs, const T *X, int64_t n, int64_t L, int64_t N, int redundant,
                           int cost_kind, double p)
{
    int64_t nl = n * L;
    T *EX = (T *)calloc(nl, sizeof(T)), *EX2 = (T *)calloc(nl, sizeof(T));
    /* sum(X, dims=3): sequential over the signal axis for each (i,j) */
    for (int64_t s = 0; s < N; s++)
        for (int64_t e = 0; e < nl; e++) {
            T xv = X[s * nl + e];
            EX[e] = (T)(EX[e] + xv);
            EX2[e] = (T)(EX2[e] + (T)(xv * xv));
        }
    int rc = FN(wxo_tree_costs_jbb_sums)(costs, EX, EX2, n, L, N, redundant, cost_kind, p);
    free(EX); free(EX2);
    return rc;
}

/* BestBasis.jl:59-83  bestbasis_treeselection(costs, n, type) (costs is mutated) */
int FN(wxo_bestbasis_treeselection)(uint8_t *tree, T *costs, int64_t k, int64_t n, int type_max)
{
    int64_t tl = ((int64_t)1 << wxo_maxtransformlevels(2 * n)) - 1;   /* gettreelength(2n) */
    if (!(k <= tl)) return -1;
    int L = wxo_getdepth_binary(k);
    int64_t ntree = n - 1;
    if (wxo_maketree1d(tree, n, L, 0) != 0) return -1;
    for (int64_t i = ntree; i >= 1; i--) {
        if (tree[i - 1]) {
            T pc = costs[i - 1];
            T cc = (T)(costs[(i << 1) - 1] + costs[(i << 1) + 1 - 1]);
            if (!type_max && cc < pc) costs[i - 1] = cc;
            else if (type_max && cc > pc) costs[i - 1] = cc;
            else wxo_delete_subtree(tree, ntree, i, 0);
        }
    }
    return wxo_isvalidtree1d(n, tree, ntree) ? 0 : -1;
}

/* ------------------------------------------------------------------------- */
/* Standard (per-signal) best basis, BB: bestbasis/bestbasis_costs.jl:94-121  */
/* coefcost(x, ::ShannonEntropyCost | ::LogEnergyEntropyCost, nrm) and        */
/* bestbasis/bestbasis_tree.jl:209-258 tree_costs(X, ::BB).                   */
/* cost_kind 0 = ShannonEntropyCost, 1 = LogEnergyEntropyCost.                */
/* norm(x) is restated as sqrt(sum x^2) accumulated in T (Julia dispatches    */
/* to BLAS nrm2 / generic_norm2: rounding-level differences, parity unpinned).*/
/* ------------------------------------------------------------------------- */
static T FN(wxo_norm2)(const T *x, int64_t n)
{
    T acc = 0;
    for (int64_t i = 0; i < n; i++) acc = (T)(acc + (T)(x[i] * x[i]));
    return (T)sqrt((double)acc);
}
static T FN(wxo_coefcost_bb)(const T *x, int64_t n, int cost_kind, T nrm)
{
    T sum = 0;
    if (nrm == sum) return sum;                                 /* bestbasis_costs.jl:115 */
    for (int64_t i = 0; i < n; i++) {
        T r = (T)(x[i] / nrm);
        T s = (T)(r * r);
        T c;
        if (s == 0) c = (T)(-0.0);
        else if (cost_kind == 0) c = (T)(-(T)(s * (T)log((double)s)));
        else c = (T)(-(T)log((double)s));
        sum = (T)(sum + c);
    }
    return sum;
}

/* tree_costs(X::Array{T,2}, ::BB) bestbasis_tree.jl:209-233; X is (n, L) */
int FN(wxo_tree_costs_bb)(T *costs, const T *X, int64_t n, int64_t L, int redundant, int cost_kind)
{
    T nrm = FN(wxo_norm2)(X, n);
    if (redundant) {
        for (int64_t i = 1; i <= L; i++) {
            int j = wxo_getdepth_binary(i);
            costs[i - 1] = (T)(FN(wxo_coefcost_bb)(X + (i - 1) * n, n, cost_kind, nrm) / (T)((int64_t)1 << j));
        }
    } else {
        int64_t i = 1;
        for (int64_t lvl = 0; lvl <= L - 1; lvl++) {
            int64_t n0 = n >> lvl;
            for (int64_t node = 0; node <= ((int64_t)1 << lvl) - 1; node++) {
                costs[i - 1] = FN(wxo_coefcost_bb)(X + lvl * n + node * n0, n0, cost_kind, nrm);
                i++;
            }
        }
    }
    return 0;
}

/* tree_costs(X::Array{T,3}, ::BB) bestbasis_tree.jl:235-258; X is (n, m, L).  The non-redundant branch
 * calls coefcost without nrm (:253), so every block is normalised by its own norm. */
int FN(wxo_tree_costs_bb2d)(T *costs, const T *X, int64_t n, int64_t m, int64_t L, int redundant, int cost_kind)
{
    if (redundant) {
        T nrm = FN(wxo_norm2)(X, n * m);
        for (int64_t i = 1; i <= L; i++) {
            int d = wxo_getdepth_quad(i);
            costs[i - 1] = (T)(FN(wxo_coefcost_bb)(X + (i - 1) * n * m, n * m, cost_kind, nrm) / (T)((int64_t)1 << (2 * d)));
        }
    } else {
        int64_t nc = ((((int64_t)1 << (2 * L)) - 1) / 3);
        T *blk = (T *)malloc(sizeof(T) * n * m);
        for (int64_t i = 1; i <= nc; i++) {
            int d = wxo_getdepth_quad(i);
            int64_t r0, r1, c0, c1;
            wxo_getrowrange(n, i, &r0, &r1); wxo_getcolrange(m, i, &c0, &c1);
            int64_t cnt = 0;
            for (int64_t c = c0; c <= c1; c++)
                for (int64_t r = r0; r <= r1; r++) blk[cnt++] = M2(X + (int64_t)d * n * m, n, r, c);
            costs[i - 1] = FN(wxo_coefcost_bb)(blk, cnt, cost_kind, FN(wxo_norm2)(blk, cnt));
        }
        free(blk);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Denoising core: noise estimate and thresholding.         */
/* Denoising.jl:214-232 noisest(x, redundant, tree) = mad!(dr)/0.6745 with    */
/* Wavelets.jl Threshold.mad!  (m = median!(y); y .= abs.(y .- m);            */
/* median!(y)) and Statistics.median! (even length: middle(a,b) = a/2 + b/2); */
/* Wavelets.jl Threshold.threshold!(x, TH, t) for HardTH / SoftTH /           */
/* SemiSoftTH / SteinTH.  Wavelets.jl is not vendored: these four loops and   */
/* mad! are restated from its published source -- parity unpinned.            */
/* ------------------------------------------------------------------------- */
static int FN(wxo_cmp)(const void *a, const void *b)
{
    T x = *(const T *)a, y = *(const T *)b;
    return (x > y) - (x < y);
}
static T FN(wxo_median_inplace)(T *y, int64_t n)
{
    qsort(y, (size_t)n, sizeof(T), FN(wxo_cmp));
    if (n & 1) return y[n / 2];
    return (T)((T)(y[n / 2 - 1] / 2) + (T)(y[n / 2] / 2));
}
/* mad!(y)/0.6745 of `cnt` values taken with stride 1 from p */
T FN(wxo_noisest_range)(const T *p, int64_t cnt)
{
    T *y = (T *)malloc(sizeof(T) * (size_t)(cnt > 0 ? cnt : 1));
    for (int64_t i = 0; i < cnt; i++) y[i] = p[i];
    T m = FN(wxo_median_inplace)(y, cnt);
    for (int64_t i = 0; i < cnt; i++) y[i] = (T)fabs((double)(T)(y[i] - m));
    T r = FN(wxo_median_inplace)(y, cnt);
    free(y);
    return (T)(r / (T)0.6745);
}
/* threshold!(x, TH, t): th_kind 0 Hard, 1 Soft, 2 SemiSoft, 3 Stein */
void FN(wxo_threshold)(T *x, int64_t cnt, int th_kind, T t)
{
    for (int64_t i = 0; i < cnt; i++) {
        T v = x[i];
        if (th_kind == 0) { if ((T)fabs((double)v) <= t) x[i] = 0; }
        else if (th_kind == 1) {
            T sh = (T)((T)fabs((double)v) - t);
            x[i] = sh < 0 ? (T)0 : (T)((v > 0 ? (T)1 : (v < 0 ? (T)-1 : v)) * sh);
        } else if (th_kind == 2) {
            /* semisoft, upper knee at 2t: 0 below t, sign(x) (2|x| - 2t) up to 2t, x above */
            T av = (T)fabs((double)v);
            if (!(av > (T)((T)2 * t))) {
                T tmp = (T)((T)((T)2 * av) - (T)((T)2 * t));
                x[i] = tmp < 0 ? (T)0 : (T)((v > 0 ? (T)1 : (v < 0 ? (T)-1 : v)) * tmp);
            }
        } else {
            T sh = (T)((T)1 - (T)((T)(t * t) / (T)(v * v)));
            x[i] = sh < 0 ? (T)0 : (T)(v * sh);
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Shift-invariant WPD: siwt/siwt_one_level.jl:71-98       */
/* sidwt_step!(w1, w2, v, h, g, s) and :154-185 isidwt_step!(v, w1, w2, h,   */
/* g, s); node cost = coefcost(Value, ShannonEntropyCost(), signalNorm)      */
/* (siwt_utls.jl:118-126 via bestbasis_costs.jl:104-124).                    */
/* ------------------------------------------------------------------------- */
void FN(wxo_sidwt_step)(T *w1, T *w2, const T *v, int64_t n, const double *h, const double *g, int filtlen, int s)
{
    int64_t n1 = n / 2;
    for (int64_t i = 1; i <= n1; i++) {
        int64_t k1 = wxo_mod1(2 * i - 1 - s, n);
        int64_t k2 = 2 * i - s;
        V1(w1, 1, i) = (T)(g[filtlen - 1] * (double)V1(v, 1, k1));
        V1(w2, 1, i) = (T)(h[0] * (double)V1(v, 1, k2));
        for (int j = 2; j <= filtlen; j++) {
            k1 = k1 + 1; if (k1 > n) k1 = wxo_mod1(k1, n);
            k2 = k2 - 1; if (k2 <= 0) k2 = wxo_mod1(k2, n);
            V1(w1, 1, i) = (T)((double)V1(w1, 1, i) + g[filtlen - j] * (double)V1(v, 1, k1));
            V1(w2, 1, i) = (T)((double)V1(w2, 1, i) + h[j - 1] * (double)V1(v, 1, k2));
        }
    }
}
void FN(wxo_isidwt_step)(T *v, const T *w1, const T *w2, int64_t n, const double *h, const double *g, int filtlen,
                         int s)
{
    int64_t n1 = n / 2;
    for (int64_t i = 1; i <= n; i++) {
        int64_t l = wxo_mod1(i - s, n);
        int j0 = (int)wxo_mod1(i, 2);
        int j1 = filtlen - j0 + 1;
        int j2 = (int)wxo_mod1(i + 1, 2);
        int64_t k1 = (i + 1) >> 1;
        int64_t k2 = (i + 1) >> 1;
        V1(v, 1, l) = (T)(g[j1 - 1] * (double)V1(w1, 1, k1) + h[j2 - 1] * (double)V1(w2, 1, k2));
        for (int j = j0 + 2; j <= filtlen; j += 2) {
            j1 = filtlen - j + 1;
            j2 = j + (j & 1) - ((j & 1) ? 0 : 1);
            k1 = k1 - 1; if (k1 <= 0) k1 = wxo_mod1(k1, n1);
            k2 = k2 + 1; if (k2 > n1) k2 = wxo_mod1(k2, n1);
            V1(v, 1, l) = (T)((double)V1(v, 1, l) + (g[j1 - 1] * (double)V1(w1, 1, k1) + h[j2 - 1] * (double)V1(w2, 1, k2)));
        }
    }
}
/* coefcost(x, ShannonEntropyCost(), nrm); nrm < 0 -> norm(x) */
T FN(wxo_siwt_nodecost)(const T *x, int64_t n, T nrm)
{
    if (nrm < 0) nrm = FN(wxo_norm2)(x, n);
    return FN(wxo_coefcost_bb)(x, n, 0, nrm);
}
T FN(wxo_siwt_norm)(const T *x, int64_t n) { return FN(wxo_norm2)(x, n); }

#undef T
#undef FN
#undef V1
#undef M2
#undef WXO_CAT
#undef WXO_CAT_
