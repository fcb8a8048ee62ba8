// wx_api_swt2d.hip -- C ABI for the 2-D redundant families (SWT, ACWT), 2-D JBB costs / tree selection and
// 2-D getbasiscoef.
#include "../../include/waveletsext_hip.h"
#include "wx_host.h"
#include "wx_kernels.h"
#include <math.h>
#include <string.h>

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

static int wx_need_device4()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    }
    return WX_OK;
}

enum { L2_DWT = 0, L2_WPT = 1, L2_WPD = 2 };

static int64_t ncols2d(int layout, int L)
{
    if (layout == L2_DWT) return 3 * (int64_t)L + 1;
    if (layout == L2_WPT) return (int64_t)1 << (2 * L);
    return ((((int64_t)1 << (2 * (L + 1))) - 1) / 3);
}

static void pack_ac(const WxFilt &f, WxAcFilt *ac)
{
    memset(ac, 0, sizeof *ac);
    ac->F = f.F;
    ac->c1 = 1.0 / sqrt(2.0);
    const double c2 = ac->c1 / 2;
    for (int k = 1; k <= f.F - 1; ++k) {
        double r = 0.0;
        for (int i = 1; i <= f.F - k; ++i) r += f.q[i - 1] * f.q[i + k - 1];
        ac->b[k - 1] = c2 * (2 * r);
    }
}

template <typename T>
static int api_red2d_fwd(const T *x, T *xw, int64_t m, int64_t n, int L, int layout, bool ac, int64_t batch,
                         const double *qmf, int F, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(m >= 1 && n >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    // SWT.jl:137-139, 478-480; ACWT.jl:135-137, 466-468 (`@assert c || throw(ArgumentError)`)
    WX_REQUIRE(L <= wx_maxtransformlevels(m < n ? m : n), WX_EARG, "Too many transform levels");
    WX_REQUIRE(L >= 1, WX_EARG, "L must be >= 1");
    WX_REQUIRE(L <= 12, WX_EUNSUPPORTED, "more than 12 redundant 2-D levels are not supported");
    if ((rc = wx_need_device4())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const int64_t mn = m * n, nc = ncols2d(layout, L);
    const T *dx = (const T *)io.in(x, sizeof(T) * mn * batch);
    T *dxw = (T *)io.out(xw, sizeof(T) * mn * nc * batch);
    if (batch && (!dx || !dxw)) return io.finish(WX_EHIP);
    const int64_t nodes = layout == L2_DWT ? 1 : ((int64_t)1 << (2 * (L - 1)));
    T *tmp = batch ? (T *)scr.alloc(sizeof(T) * 2 * nodes * batch * mn) : nullptr;
    if (batch && !tmp) return io.finish(WX_EHIP);
    WxAcFilt acf;
    if (ac) pack_ac(filt, &acf);
    rc = wx_dev_red2d_fwd<T>(dx, dxw, m, n, L, layout, batch, filt, ac ? &acf : nullptr, tmp, st);
    return io.finish(rc);
}

template <typename T>
static int api_red2d_inv(const T *xw, T *x, int64_t m, int64_t n, int64_t ncols, int L, int layout, bool ac,
                         const uint8_t *tree, int64_t ntree, int64_t sm, int64_t batch, const double *qmf, int F,
                         void *stream)
{
    WxFilt filt;
    memset(&filt, 0, sizeof filt);
    int rc;
    if (!ac && (rc = wx_pack_filter(qmf, F, &filt))) return rc;
    WX_REQUIRE(m >= 1 && n >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    const int Lmax = wx_maxtransformlevels(m < n ? m : n);
    int Leff = L;
    if (layout == L2_DWT) {
        // SWT.jl:293 `@assert 0 <= log2(sm) <= L`, then main2depthshift (Utils.jl:298) `@assert sm < 1<<L`
        if (!ac && sm >= 0) WX_REQUIRE(sm >= 1 && sm < ((int64_t)1 << L), WX_EASSERT, "isdwt! 2-D: bad shift sm");
    } else if (layout == L2_WPT) {
        WX_REQUIRE(L <= Lmax, WX_EASSERT, "@assert log(4,k) <= maxtransformlevels(x) (SWT.jl:656, ACWT.jl:620)");
        if (!ac && sm >= 0) WX_REQUIRE(sm < ((int64_t)1 << L), WX_EASSERT, "main2depthshift: @assert sm < 1<<L");
    } else {
        const int Lshift = wx_getdepth_quad(ncols);
        if (tree) {
            WX_REQUIRE(wx_isvalidtree2d(m, n, tree, ntree), WX_EASSERT, "@assert isvalidtree(x, tree) (SWT.jl:1101, ACWT.jl:973)");
            Leff = wx_tree_depth2d(tree, ntree);
        } else {
            WX_REQUIRE(L <= Lmax, WX_EARG, "Too many transform levels.");
            WX_REQUIRE(L >= 1, WX_EARG, "L must be >= 1");
        }
        if (!ac && sm >= 0) WX_REQUIRE(sm < ((int64_t)1 << Lshift), WX_EASSERT, "main2depthshift: @assert sm < 1<<L");
        WX_REQUIRE(ncols2d(L2_WPD, Leff) <= ncols, WX_EBOUNDS, "tree reaches below the last slice of xw");
    }
    WX_REQUIRE(Leff <= 12, WX_EUNSUPPORTED, "more than 12 redundant 2-D levels are not supported");
    if ((rc = wx_need_device4())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    const uint8_t *dtree = nullptr;
    if (layout == L2_WPD && tree) {
        int64_t nfull = 0, pw = 1;
        for (int i = 0; i < Leff; ++i) { nfull += pw; pw *= 4; }
        bool full = true;
        for (int64_t i = 1; i <= nfull && full; ++i) full = tree[i - 1] != 0;
        if (!full) { dtree = (const uint8_t *)scr.upload(tree, (size_t)ntree); if (!dtree) return WX_EHIP; }
    }
    WxIO io(st);
    const int64_t mn = m * n;
    const T *dxw = (const T *)io.in(xw, sizeof(T) * mn * ncols * batch);
    T *dx = (T *)io.out(x, sizeof(T) * mn * batch);
    if (batch && (!dxw || !dx)) return io.finish(WX_EHIP);
    int64_t need[2] = {0, 0}, jobs_max = 1;
    for (int d = 0; d <= Leff - 1; ++d) {
        const int64_t cols = layout == L2_DWT ? 1 : ((int64_t)1 << (2 * d));
        if (d >= 1 && cols > need[d & 1]) need[d & 1] = cols;
        if (cols > jobs_max) jobs_max = cols;
    }
    T *s0 = nullptr, *s1 = nullptr, *tmp = nullptr;
    if (batch && need[0]) { s0 = (T *)scr.alloc(sizeof(T) * mn * need[0] * batch); if (!s0) return io.finish(WX_EHIP); }
    if (batch && need[1]) { s1 = (T *)scr.alloc(sizeof(T) * mn * need[1] * batch); if (!s1) return io.finish(WX_EHIP); }
    if (batch && !ac && Leff > 0) { tmp = (T *)scr.alloc(sizeof(T) * 2 * jobs_max * batch * mn); if (!tmp) return io.finish(WX_EHIP); }
    rc = wx_dev_red2d_inv<T>(dxw, dx, m, n, Leff, layout, ncols, batch, sm, ac, dtree, ntree, filt, s0, s1, tmp, st);
    return io.finish(rc);
}

template <typename T>
static int api_jbb_costs2d(const T *sum, const T *sumsq, int64_t Ntot, int64_t m, int64_t n, int64_t k, int redundant,
                           int cost_kind, double p, T *costs, void *stream)
{
    WX_REQUIRE(m >= 1 && n >= 1 && k >= 1 && Ntot >= 1, WX_EARG, "bad dimensions");
    WX_REQUIRE(cost_kind == 0 || cost_kind == 1, WX_EARG, "cost_kind must be 0 (LoglpCost) or 1 (NormCost)");
    if (!redundant) WX_REQUIRE(k - 1 <= wx_maxtransformlevels(m < n ? m : n) && k <= 14, WX_EASSERT, "more packet levels than the image admits");
    int rc;
    if ((rc = wx_need_device4())) return rc;
    hipStream_t st = wx_stream(stream);
    WxIO io(st);
    const int64_t ncost = redundant ? k : ((((int64_t)1 << (2 * k)) - 1) / 3);
    const T *ds = (const T *)io.in(sum, sizeof(T) * m * n * k);
    const T *dq = (const T *)io.in(sumsq, sizeof(T) * m * n * k);
    T *dc = (T *)io.out(costs, sizeof(T) * ncost);
    if (!ds || !dq || !dc) return io.finish(WX_EHIP);
    rc = wx_dev_jbb_costs2d<T>(ds, dq, Ntot, m, n, k, redundant, cost_kind, p, dc, st);
    return io.finish(rc);
}

// BestBasis.jl:85-110 bestbasis_treeselection(costs, n, m, type) + delete_subtree!(:quad) (host only)
static void del_subtree4(uint8_t *bt, int64_t len, int64_t i)
{
    bt[i - 1] = 0;
    for (int c = 0; c < 4; ++c) {
        const int64_t ch = 4 * i - 2 + c;
        if (ch <= len && bt[ch - 1]) del_subtree4(bt, len, ch);
    }
}
template <typename T>
static int api_treeselect2d(T *costs, int64_t k, int64_t m, int64_t n, int type_max, uint8_t *tree)
{
    WX_REQUIRE(costs && tree, WX_EARG, "NULL argument");
    WX_REQUIRE(type_max == 0 || type_max == 1, WX_EARG, "Unsupported type (BestBasis.jl:91)");
    WX_REQUIRE(k <= wx_gettreelength2d(2 * m, 2 * n), WX_EASSERT, "@assert k <= gettreelength(2*n,2*m) (BestBasis.jl:90)");
    const int L = wx_getdepth_quad(k);
    WX_REQUIRE(L <= wx_maxtransformlevels(m < n ? m : n), WX_EASSERT, "maketree(n, m, L, :full)");
    WX_REQUIRE(L == 0 || k >= ((((int64_t)1 << (2 * (L + 1))) - 1) / 3), WX_EBOUNDS,
               "costs do not cover the children of depth L-1");
    const int64_t ntree = wx_gettreelength2d(m, n);
    memset(tree, 0, (size_t)ntree);
    int64_t nfull = 0, pw = 1;
    for (int i = 0; i < L; ++i) { nfull += pw; pw *= 4; }
    for (int64_t i = 1; i <= nfull; ++i) tree[i - 1] = 1;
    for (int64_t i = ntree; i >= 1; --i) {
        if (!tree[i - 1]) continue;
        const T pc = costs[i - 1];
        const T cc = (T)((T)((T)(costs[4 * i - 3] + costs[4 * i - 2]) + costs[4 * i - 1]) + costs[4 * i]);
        if (!type_max && cc < pc) costs[i - 1] = cc;
        else if (type_max && cc > pc) costs[i - 1] = cc;
        else del_subtree4(tree, ntree, i);
    }
    WX_REQUIRE(wx_isvalidtree2d(m, n, tree, ntree), WX_EASSERT, "@assert isvalidtree(zeros(n,m), tree)");
    return WX_OK;
}

static void colmap2d_rec(const uint8_t *tree, int64_t ntree, int64_t node, int d, int j, int k, int Leff, std::vector<int> &col)
{
    const int nblk = 1 << Leff;
    if (node <= ntree && tree[node - 1]) {
        for (int c = 0; c < 4; ++c) colmap2d_rec(tree, ntree, 4 * node - 2 + c, d + 1, 2 * j + (c >> 1), 2 * k + (c & 1), Leff, col);
    } else {
        const int w = 1 << (Leff - d);
        for (int r = j * w; r < (j + 1) * w; ++r)
            for (int c = k * w; c < (k + 1) * w; ++c) col[(size_t)r * nblk + c] = d;
    }
}

template <typename T>
static int api_getbasiscoef2d(const T *Xw, T *out, int64_t m, int64_t n, int k, const uint8_t *tree, int64_t ntree,
                              int64_t batch, void *stream)
{
    WX_REQUIRE(m >= 1 && n >= 1 && batch >= 0 && k >= 1, WX_EARG, "getbasiscoef: bad dimensions");
    WX_REQUIRE(wx_isvalidtree2d(m, n, tree, ntree), WX_EASSERT, "@assert isvalidtree(x, tree) (Utils.jl:109)");
    const int Lmax = wx_maxtransformlevels(m < n ? m : n);
    WX_REQUIRE(k - 1 <= Lmax, WX_EASSERT, "@assert k-1 <= L (Utils.jl:110)");
    WX_REQUIRE(4 * ntree + 1 == ((((int64_t)1 << (2 * (Lmax + 1))) - 1) / 3), WX_EASSERT, "@assert leaf_len == length(leaf) (Utils.jl:113)");
    const int Leff = wx_tree_depth2d(tree, ntree);
    WX_REQUIRE(Leff < k, WX_EARG, "Not enough decomposition levels in Xw (Utils.jl:120)");
    int rc;
    if ((rc = wx_need_device4())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    std::vector<int> col((size_t)1 << (2 * Leff), 0);
    colmap2d_rec(tree, ntree, 1, 0, 0, 0, Leff, col);
    const int *dcol = (const int *)scr.upload(col.data(), col.size() * sizeof(int));
    if (!dcol) return WX_EHIP;
    WxIO io(st);
    const T *dX = (const T *)io.in(Xw, sizeof(T) * m * n * k * batch);
    T *dout = (T *)io.out(out, sizeof(T) * m * n * batch);
    if (batch && (!dX || !dout)) return io.finish(WX_EHIP);
    rc = wx_dev_gather_leaves2d<T>(dX, dout, m, n, k, batch, dcol, 1 << Leff, st);
    return io.finish(rc);
}

extern "C" {

#define WX_FWD2(name, layout, ac)                                                                                         \
    int name##_f64(const double *x, double *xw, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F,     \
                   void *stream)                                                                                          \
    { return api_red2d_fwd<double>(x, xw, m, n, L, layout, ac, batch, qmf, F, stream); }
#define WX_FWD2_32(name, layout)                                                                                          \
    int name##_f32(const float *x, float *xw, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F,       \
                   void *stream)                                                                                          \
    { return api_red2d_fwd<float>(x, xw, m, n, L, layout, false, batch, qmf, F, stream); }
WX_FWD2(wx_sdwt2d, L2_DWT, false) WX_FWD2_32(wx_sdwt2d, L2_DWT)
WX_FWD2(wx_swpt2d, L2_WPT, false) WX_FWD2_32(wx_swpt2d, L2_WPT)
WX_FWD2(wx_swpd2d, L2_WPD, false) WX_FWD2_32(wx_swpd2d, L2_WPD)
WX_FWD2(wx_acdwt2d, L2_DWT, true)
WX_FWD2(wx_acwpt2d, L2_WPT, true)
WX_FWD2(wx_acwpd2d, L2_WPD, true)

int wx_isdwt2d_f64(const double *xw, double *x, int64_t m, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{ return api_red2d_inv<double>(xw, x, m, n, 3 * (int64_t)L + 1, L, L2_DWT, false, nullptr, 0, sm, batch, qmf, F, stream); }
int wx_isdwt2d_f32(const float *xw, float *x, int64_t m, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{ return api_red2d_inv<float>(xw, x, m, n, 3 * (int64_t)L + 1, L, L2_DWT, false, nullptr, 0, sm, batch, qmf, F, stream); }
int wx_iswpt2d_f64(const double *xw, double *x, int64_t m, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{ return api_red2d_inv<double>(xw, x, m, n, (int64_t)1 << (2 * L), L, L2_WPT, false, nullptr, 0, sm, batch, qmf, F, stream); }
int wx_iswpt2d_f32(const float *xw, float *x, int64_t m, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{ return api_red2d_inv<float>(xw, x, m, n, (int64_t)1 << (2 * L), L, L2_WPT, false, nullptr, 0, sm, batch, qmf, F, stream); }
int wx_iswpd2d_f64(const double *xw, double *x, int64_t m, int64_t n, int64_t ncols, int L, const uint8_t *tree, int64_t ntree,
                   int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{ return api_red2d_inv<double>(xw, x, m, n, ncols, L, L2_WPD, false, tree, ntree, sm, batch, qmf, F, stream); }
int wx_iswpd2d_f32(const float *xw, float *x, int64_t m, int64_t n, int64_t ncols, int L, const uint8_t *tree, int64_t ntree,
                   int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{ return api_red2d_inv<float>(xw, x, m, n, ncols, L, L2_WPD, false, tree, ntree, sm, batch, qmf, F, stream); }

int wx_iacdwt2d_f64(const double *xw, double *x, int64_t m, int64_t n, int L, int64_t batch, void *stream)
{ return api_red2d_inv<double>(xw, x, m, n, 3 * (int64_t)L + 1, L, L2_DWT, true, nullptr, 0, -1, batch, nullptr, 0, stream); }
int wx_iacwpt2d_f64(const double *xw, double *x, int64_t m, int64_t n, int L, int64_t batch, void *stream)
{ return api_red2d_inv<double>(xw, x, m, n, (int64_t)1 << (2 * L), L, L2_WPT, true, nullptr, 0, -1, batch, nullptr, 0, stream); }
int wx_iacwpd2d_f64(const double *xw, double *x, int64_t m, int64_t n, int64_t ncols, int L, const uint8_t *tree, int64_t ntree,
                    int64_t batch, void *stream)
{ return api_red2d_inv<double>(xw, x, m, n, ncols, L, L2_WPD, true, tree, ntree, -1, batch, nullptr, 0, stream); }

int wx_jbb_costs2d_f64(const double *sum, const double *sumsq, int64_t Ntot, int64_t m, int64_t n, int64_t k, int redundant,
                       int cost_kind, double p, double *costs, void *stream)
{ return api_jbb_costs2d<double>(sum, sumsq, Ntot, m, n, k, redundant, cost_kind, p, costs, stream); }
int wx_jbb_costs2d_f32(const float *sum, const float *sumsq, int64_t Ntot, int64_t m, int64_t n, int64_t k, int redundant,
                       int cost_kind, double p, float *costs, void *stream)
{ return api_jbb_costs2d<float>(sum, sumsq, Ntot, m, n, k, redundant, cost_kind, p, costs, stream); }
int wx_treeselect2d_f64(double *costs, int64_t k, int64_t m, int64_t n, int type_max, uint8_t *tree)
{ return api_treeselect2d<double>(costs, k, m, n, type_max, tree); }
int wx_treeselect2d_f32(float *costs, int64_t k, int64_t m, int64_t n, int type_max, uint8_t *tree)
{ return api_treeselect2d<float>(costs, k, m, n, type_max, tree); }
int wx_getbasiscoef2d_f64(const double *Xw, double *out, int64_t m, int64_t n, int k, const uint8_t *tree, int64_t ntree,
                          int64_t batch, void *stream)
{ return api_getbasiscoef2d<double>(Xw, out, m, n, k, tree, ntree, batch, stream); }
int wx_getbasiscoef2d_f32(const float *Xw, float *out, int64_t m, int64_t n, int k, const uint8_t *tree, int64_t ntree,
                          int64_t batch, void *stream)
{ return api_getbasiscoef2d<float>(Xw, out, m, n, k, tree, ntree, batch, stream); }

}  // extern "C"
