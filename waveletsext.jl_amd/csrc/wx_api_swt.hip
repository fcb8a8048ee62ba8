// wx_api_swt.hip -- C ABI for the redundant families (SWT, ACWT) and the JBB best-basis
// reduction.  Argument checks mirror the reference's @assert / throw(ArgumentError) sites.
#include "../../include/waveletsext_hip.h"
#include "wx_host.h"
#include "wx_kernels.h"
#include <math.h>
#include <string.h>

int wx_force_generic();

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

static int wx_need_device2()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    }
    return WX_OK;
}

enum { LAYOUT_DWT = 0, LAYOUT_WPT = 1, LAYOUT_WPD = 2 };

static int64_t wx_ncols(int layout, int L)
{
    return layout == LAYOUT_DWT ? L + 1 : (layout == LAYOUT_WPT ? ((int64_t)1 << L) : ((int64_t)1 << (L + 1)) - 1);
}

// acwt_utils.jl:7-48: a_k = 2 sum_i q[i] q[i+k]; b = c2 * a with c1 = 1/sqrt(2), c2 = c1/2
static void wx_pack_acfilter(const WxFilt &f, WxAcFilt *ac)
{
    memset(ac, 0, sizeof *ac);
    ac->F = f.F;
    ac->c1 = 1.0 / sqrt(2.0);
    const double c2 = ac->c1 / 2;
    for (int k = 1; k <= f.F - 1; ++k) {
        double r = 0.0;
        for (int i = 1; i <= f.F - k; ++i) r += f.q[i - 1] * f.q[i + k - 1];
        r *= 2;
        ac->b[k - 1] = c2 * r;
    }
}

// ---- forward: sdwt / swpt / swpd / acdwt / acwpt / acwpd -----------------------------------
template <typename T>
static int api_redundant_fwd(const T *x, T *xw, int64_t n, int L, int layout, bool ac, int64_t batch,
                             const double *qmf, int F, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(n >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    // SWT.jl:113-115, 443-447, 844-846; ACWT.jl:113-114, 431-433, 737-739 (`@assert c || throw(ArgumentError)`)
    WX_REQUIRE(L <= wx_maxtransformlevels(n), WX_EARG, "Too many transform levels (length(x) < 2^L)");
    WX_REQUIRE(L >= 1, WX_EARG, "L must be >= 1");
    WX_REQUIRE(L <= 24, WX_EUNSUPPORTED, "more than 24 redundant levels are not supported");
    if ((rc = wx_need_device2())) return rc;
    hipStream_t st = wx_stream(stream);
    WxIO io(st);
    const int64_t ncols = wx_ncols(layout, L);
    const T *dx = (const T *)io.in(x, sizeof(T) * n * batch);
    T *dxw = (T *)io.out(xw, sizeof(T) * n * ncols * batch);
    if ((batch && n) && (!dx || !dxw)) return io.finish(WX_EHIP);
    WxAcFilt acf;
    if (ac) wx_pack_acfilter(filt, &acf);
    rc = wx_dev_swt_fwd<T>(dx, dxw, n, L, layout, batch, filt, ac ? &acf : nullptr, st);
    return io.finish(rc);
}

// ---- SWT inverse ----------------------------------------------------------------------------
template <typename T>
static int api_swt_inv(const T *xw, T *x, int64_t n, int64_t ncols, int L, int layout, const uint8_t *tree,
                       int64_t ntree, int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(n >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    int Leff = L;          // levels to run
    int Lshift = L;        // L used by main2depthshift
    if (layout == LAYOUT_DWT) {
        if (sm >= 0) WX_REQUIRE(sm >= 1 && sm < ((int64_t)1 << L), WX_EASSERT, "@assert 0 <= log2(sm) < L (SWT.jl:266)");
    } else if (layout == LAYOUT_WPT) {
        // SWT.jl:619-626
        WX_REQUIRE(L <= wx_maxtransformlevels(n), WX_EARG,
                   "Number of nodes in `xw` is more than possible number of nodes at any depth for signal of length `n`");
        if (sm >= 0) WX_REQUIRE(sm < ((int64_t)1 << L), WX_EASSERT, "main2depthshift: @assert sm < 1<<L (Utils.jl:298)");
    } else {
        Lshift = wx_getdepth_binary(ncols);                        // SWT.jl:1074 L = getdepth(m,:binary)
        if (tree) {
            WX_REQUIRE(wx_isvalidtree1d(n, tree, ntree), WX_EASSERT, "@assert isvalidtree(xw[:,1], tree) (SWT.jl:1069)");
            Leff = wx_tree_depth1d(tree, ntree);
        } else {
            // SWT.jl:1040-1047 (`@assert c || throw(ArgumentError)`), then maketree(n, L, :full)
            WX_REQUIRE(L <= wx_maxtransformlevels(n), WX_EARG, "Too many transform levels.");
            WX_REQUIRE(L >= 1, WX_EARG, "L must be >= 1");
            WX_REQUIRE(wx_isdyadic(n), WX_EASSERT, "maketree: isdyadic(n)");
        }
        if (sm >= 0) WX_REQUIRE(sm < ((int64_t)1 << Lshift), WX_EASSERT, "main2depthshift: @assert sm < 1<<L (Utils.jl:298)");
        WX_REQUIRE(((int64_t)1 << (Leff + 1)) - 1 <= ncols, WX_EBOUNDS, "tree reaches below the last column of xw");
    }
    WX_REQUIRE(Leff <= 24, WX_EUNSUPPORTED, "more than 24 redundant levels are not supported");
    if ((rc = wx_need_device2())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    const uint8_t *dtree = nullptr;
    if (layout == LAYOUT_WPD && tree) {
        bool full = true;
        for (int64_t i = 1; i <= ((int64_t)1 << Leff) - 1 && full; ++i) full = tree[i - 1] != 0;
        if (!full) {
            dtree = (const uint8_t *)scr.upload(tree, (size_t)ntree);
            if (!dtree) return WX_EHIP;
        }
    }
    WxIO io(st);
    const T *dxw = (const T *)io.in(xw, sizeof(T) * n * ncols * batch);
    T *dx = (T *)io.out(x, sizeof(T) * n * batch);
    if ((batch && n) && (!dxw || !dx)) return io.finish(WX_EHIP);
    // a full tree only reads its leaves, the last 2^Leff columns of the heap-ordered table: that is the iswpt
    // layout at a column offset (signal stride stays ncols), so it takes the fused iswpt passes
    if (layout == LAYOUT_WPD && !dtree && Leff >= 1 && (sm < 0 || Lshift == Leff)) {
        layout = LAYOUT_WPT;
        dxw += (((int64_t)1 << Leff) - 1) * n;
    }
    // level buffers follow the inverse schedule (fused iswpt passes skip every other depth)
    WxSwtInvPlan plan;
    wx_swt_inv_plan(layout, Leff, F, sm, n, sizeof(T), dtree != nullptr, &plan,
                    layout == LAYOUT_WPT && sm < 0 && !dtree && wx_haar_swpt6_ok(n, Leff, filt, sizeof(T)));
    T *s0 = nullptr, *s1 = nullptr;
    if (batch && plan.need_cols[0]) { s0 = (T *)scr.alloc(sizeof(T) * n * plan.need_cols[0] * batch); if (!s0) return io.finish(WX_EHIP); }
    if (batch && plan.need_cols[1]) { s1 = (T *)scr.alloc(sizeof(T) * n * plan.need_cols[1] * batch); if (!s1) return io.finish(WX_EHIP); }
    // with a sparse tree the shifts still follow the table depth (sd has Lshift+1 entries)
    rc = wx_dev_swt_inv<T>(dxw, dx, n, Leff, layout, (int)ncols, batch, sm, dtree, ntree, filt, plan, s0, s1, st);
    return io.finish(rc);
}

// ---- ACWT inverses --------------------------------------------------------------------------
template <typename T>
static int api_iac(const T *xw, T *x, int64_t n, int64_t ncols, int L, int layout, const uint8_t *tree,
                   int64_t ntree, int64_t batch, void *stream)
{
    WX_REQUIRE(n >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    int rc;
    int Leff = L;
    if (layout == LAYOUT_WPT) {
        WX_REQUIRE(L <= wx_maxtransformlevels(n), WX_EARG,
                   "Number of nodes in `xw` is more than possible number of nodes at any depth for signal of length `n`");
        WX_REQUIRE(L <= 30, WX_EUNSUPPORTED, "too many levels");
    } else if (layout == LAYOUT_WPD) {
        if (tree) {
            WX_REQUIRE(wx_isvalidtree1d(n, tree, ntree), WX_EASSERT, "@assert isvalidtree(x, tree) (ACWT.jl:948)");
            Leff = wx_tree_depth1d(tree, ntree);
        } else {
            WX_REQUIRE(L <= wx_maxtransformlevels(n), WX_EARG, "Too many transform levels.");   // ACWT.jl:924-926
            WX_REQUIRE(L >= 1, WX_EARG, "L must be >= 1.");
            WX_REQUIRE(wx_isdyadic(n), WX_EASSERT, "maketree: isdyadic(n)");
        }
        WX_REQUIRE(((int64_t)1 << (Leff + 1)) - 1 <= ncols, WX_EBOUNDS, "tree reaches below the last column of xw");
        WX_REQUIRE(Leff <= 30, WX_EUNSUPPORTED, "too many levels");
    }
    if ((rc = wx_need_device2())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    const uint8_t *dtree = nullptr;
    if (layout == LAYOUT_WPD && tree) {
        dtree = (const uint8_t *)scr.upload(tree, (size_t)ntree);
        if (!dtree) return WX_EHIP;
    }
    WxIO io(st);
    const T *dxw = (const T *)io.in(xw, sizeof(T) * n * ncols * batch);
    T *dx = (T *)io.out(x, sizeof(T) * n * batch);
    if ((batch && n) && (!dxw || !dx)) return io.finish(WX_EHIP);
    if (layout == LAYOUT_DWT) rc = wx_dev_iacdwt<T>(dxw, dx, n, L, batch, st);
    else if (layout == LAYOUT_WPT) rc = wx_dev_iacwpt<T>(dxw, dx, n, L, batch, st);
    else rc = wx_dev_iacwpd<T>(dxw, dx, n, (int)ncols, batch, dtree, ntree, Leff, st);
    return io.finish(rc);
}

// ---- JBB --------------------------------------------------------------------------------------
template <typename T>
static int api_jbb_moments(const T *X, T *sum, T *sumsq, int64_t nk, int64_t batch, int accumulate, void *stream)
{
    WX_REQUIRE(nk >= 0 && batch >= 0, WX_EARG, "bad dimensions");
    int rc;
    if ((rc = wx_need_device2())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dX = (const T *)io.in(X, sizeof(T) * nk * batch);
    // sum / sumsq are read (accumulate) and written
    T *dsum, *dsq;
    if (accumulate) {
        dsum = (T *)io.in(sum, sizeof(T) * nk);
        dsq = (T *)io.in(sumsq, sizeof(T) * nk);
        // staged host buffers must also be copied back
        for (auto &it : io.items) if (it.user == sum || it.user == sumsq) it.copy_out = true;
    } else {
        dsum = (T *)io.out(sum, sizeof(T) * nk);
        dsq = (T *)io.out(sumsq, sizeof(T) * nk);
    }
    if (nk && (!dsum || !dsq || (batch && !dX))) return io.finish(WX_EHIP);
    // few elements but many signals: split the signal axis so the chip is filled
    int nchunks = 1;
    if (nk < ((int64_t)1 << 20) && batch >= 64) {
        int64_t want = (((int64_t)1 << 21) + nk - 1) / nk;
        if (want > batch / 16) want = batch / 16;
        if (want > 1024) want = 1024;
        nchunks = (int)(want < 1 ? 1 : want);
    }
    T *scratch = nullptr;
    if (nchunks > 1) {
        scratch = (T *)scr.alloc(sizeof(T) * 2 * nchunks * nk);
        if (!scratch) return io.finish(WX_EHIP);
    }
    rc = wx_dev_jbb_moments<T>(dX, dsum, dsq, nk, batch, accumulate, scratch, nchunks, st);
    return io.finish(rc);
}

template <typename T>
static int api_jbb_costs(const T *sum, const T *sumsq, int64_t Ntot, int64_t n, int64_t k, int redundant,
                         int cost_kind, double p, T *costs, void *stream)
{
    WX_REQUIRE(n >= 1 && k >= 1 && Ntot >= 1, WX_EARG, "bad dimensions");
    WX_REQUIRE(cost_kind == 0 || cost_kind == 1, WX_EARG, "cost_kind must be 0 (LoglpCost) or 1 (NormCost)");
    if (!redundant) WX_REQUIRE(k - 1 <= wx_maxtransformlevels(n) && k <= 30, WX_EASSERT, "more packet levels than the signal admits");
    int rc;
    if ((rc = wx_need_device2())) return rc;
    hipStream_t st = wx_stream(stream);
    WxIO io(st);
    const int64_t ncost = redundant ? k : (((int64_t)1 << k) - 1);
    const T *ds = (const T *)io.in(sum, sizeof(T) * n * k);
    const T *dq = (const T *)io.in(sumsq, sizeof(T) * n * k);
    T *dc = (T *)io.out(costs, sizeof(T) * ncost);
    if (!ds || !dq || !dc) return io.finish(WX_EHIP);
    rc = wx_dev_jbb_costs<T>(ds, dq, Ntot, n, k, redundant, cost_kind, p, dc, st);
    return io.finish(rc);
}

// BestBasis.jl:59-83 bestbasis_treeselection + :128-140 delete_subtree!  (host only, tiny)
static void wx_delete_subtree(uint8_t *bt, int64_t len, int64_t i)
{
    bt[i - 1] = 0;
    for (int c = 0; c < 2; ++c) {
        const int64_t ch = 2 * i + c;
        if (ch <= len && bt[ch - 1]) wx_delete_subtree(bt, len, ch);
    }
}
template <typename T>
static int api_treeselect(T *costs, int64_t k, int64_t n, int type_max, uint8_t *tree, double *min_rel_gap = nullptr)
{
    WX_REQUIRE(costs && tree, WX_EARG, "NULL argument");
    WX_REQUIRE(type_max == 0 || type_max == 1, WX_EARG, "Unsupported type (BestBasis.jl:64)");
    WX_REQUIRE(n >= 1 && k >= 1, WX_EARG, "bad dimensions");
    const int64_t tl = ((int64_t)1 << wx_maxtransformlevels(2 * n)) - 1;
    WX_REQUIRE(k <= tl, WX_EASSERT, "@assert k <= gettreelength(2*n) (BestBasis.jl:63)");
    const int L = wx_getdepth_binary(k);
    WX_REQUIRE(wx_isdyadic(n) && L <= wx_maxtransformlevels(n), WX_EASSERT, "maketree(n, L, :full)");
    // the loop below reads the children of every node of depth < L: costs[2^(L+1)-2] (the reference would throw a
    // BoundsError for a shorter vector)
    WX_REQUIRE(L == 0 || k >= ((int64_t)1 << (L + 1)) - 1, WX_EBOUNDS, "costs do not cover the children of depth L-1");
    const int64_t ntree = n - 1;
    memset(tree, 0, (size_t)ntree);
    for (int64_t i = 1; i <= ((int64_t)1 << L) - 1; ++i) tree[i - 1] = 1;
    // the margin of the closest decision actually taken: min |cc - pc| / |pc| ("report the minimum relative cost gap
    // next to the tree"): a device / reference difference in summation order can only flip a split whose margin is of the
    // order of the costs' rounding error
    double gap = INFINITY;
    for (int64_t i = ntree; i >= 1; --i) {
        if (!tree[i - 1]) continue;
        const T pc = costs[i - 1];
        const T cc = (T)(costs[2 * i - 1] + costs[2 * i]);
        if (min_rel_gap) {
            const double diff = fabs((double)cc - (double)pc), den = fabs((double)pc);
            const double g = diff == 0.0 ? 0.0 : (den > 0.0 ? diff / den : INFINITY);
            if (g < gap) gap = g;                                    // NaN / inf - inf margins never lower it
        }
        if (!type_max && cc < pc) costs[i - 1] = cc;
        else if (type_max && cc > pc) costs[i - 1] = cc;
        else wx_delete_subtree(tree, ntree, i);
    }
    if (min_rel_gap) *min_rel_gap = gap;
    WX_REQUIRE(wx_isvalidtree1d(n, tree, ntree), WX_EASSERT, "@assert isvalidtree(zeros(n), tree)");
    return WX_OK;
}

// acwpd + JBB moments without materialising the whole (n, 2^(L+1)-1, batch) table: signals are
// processed in chunks through a bounded scratch table; moments accumulate in signal order.
static int api_acwpd_jbb_moments(const double *x, double *sum, double *sumsq, int64_t n, int L, int64_t batch,
                                 const double *qmf, int F, int accumulate, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(n >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    WX_REQUIRE(L <= wx_maxtransformlevels(n), WX_EARG, "Too many transform levels (length(x) < 2^L)");
    WX_REQUIRE(L >= 1 && L <= 24, WX_EARG, "L must be >= 1");
    if ((rc = wx_need_device2())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const int64_t ncols = ((int64_t)1 << (L + 1)) - 1;
    const int64_t nk = n * ncols;
    const double *dx = (const double *)io.in(x, sizeof(double) * n * batch);
    double *dsum, *dsq;
    if (accumulate) {
        dsum = (double *)io.in(sum, sizeof(double) * nk);
        dsq = (double *)io.in(sumsq, sizeof(double) * nk);
        for (auto &it : io.items) if (it.user == sum || it.user == sumsq) it.copy_out = true;
    } else {
        dsum = (double *)io.out(sum, sizeof(double) * nk);
        dsq = (double *)io.out(sumsq, sizeof(double) * nk);
    }
    if (!dsum || !dsq || (batch && !dx)) return io.finish(WX_EHIP);
    WxAcFilt acf;
    wx_pack_acfilter(filt, &acf);
    if (batch == 0 && !accumulate) {
        WX_HIP_CHECK(hipMemsetAsync(dsum, 0, sizeof(double) * nk, st));
        WX_HIP_CHECK(hipMemsetAsync(dsq, 0, sizeof(double) * nk, st));
    }
    const int D0 = wx_force_generic() ? -1 : wx_acwpd_fused_depth(n, L, F);
    if (D0 >= 0 && D0 < L) {
        // fused: only the shallow top table (depth <= D0) goes through HBM
        const int64_t ncols_top = ((int64_t)1 << (D0 + 1)) - 1;
        const int64_t nk_top = n * ncols_top;
        int64_t chunk = ((int64_t)8 << 30) / (int64_t)(sizeof(double) * nk_top);
        static const int64_t chunk_env = wx_getenv("WX_ACWPD_CHUNK") ? atoll(wx_getenv("WX_ACWPD_CHUNK")) : 0;   // signals per pass (experiment: table in the Infinity Cache)
        if (chunk_env > 0 && chunk_env < chunk) chunk = chunk_env;
        if (chunk < 1) chunk = 1;
        if (chunk > batch) chunk = batch;
        double *tab = batch ? (double *)scr.alloc(sizeof(double) * nk_top * chunk) : nullptr;
        if (batch && !tab) return io.finish(WX_EHIP);
        // First moments by LINEARITY (round 5): sum_b acwpd(x_b) = acwpd(sum_b x_b), so the sums of all 2^(L+1)-1 columns are ONE
        // extra signal through the plain acwpd kernels (64 MiB for n = 2048) instead of an add per coefficient and signal inside the
        // moment kernels, which are bound by FP64 issue.  Rounding differs from the reference's sequential sum(X, dims=3)
        // (bestbasis_tree.jl:153) by the association only; identical signals in a power-of-two batch still give sigma = 0 exactly
        // (scaling by 2^k commutes with every rounding of the transform).
        static const bool lin_off = wx_getenv("WX_ACWPD_LINSUM") && atoi(wx_getenv("WX_ACWPD_LINSUM")) == 0;
        const bool linear = !lin_off && batch > 0 && D0 > 0 && wx_acwpd_top_moments_ok(n, D0) && wx_acwpd_mfma_ok(n, L, D0);
        if (linear) {
            const int groups = 64;
            double *part = (double *)scr.alloc(sizeof(double) * n * (groups + 1));
            double *tsum = accumulate ? (double *)scr.alloc(sizeof(double) * nk) : dsum;
            if (!part || !tsum) return io.finish(WX_EHIP);
            double *ssig = part + (int64_t)groups * n;
            rc = wx_dev_sum_signals(dx, n, batch, ssig, part, groups, st);
            if (rc == WX_OK) rc = wx_dev_swt_fwd<double>(ssig, tsum, n, L, LAYOUT_WPD, 1, filt, &acf, st);
            if (rc == WX_OK && accumulate) rc = wx_dev_add_to(dsum, tsum, nk, st);
            if (rc != WX_OK) return io.finish(rc);
        }
        double *sum_dump = nullptr;
        for (int64_t b0 = 0; b0 < batch && rc == WX_OK; b0 += chunk) {
            const int64_t bc = (batch - b0 < chunk) ? batch - b0 : chunk;
            const int acc = (accumulate || b0 > 0) ? 1 : 0;
            // the top table and its moments in the same passes (only what the subtree kernel and the next pass read is written);
            // otherwise three plain passes and the moment kernel over the whole top table
            int fusedm = D0 > 0 ? wx_dev_acwpd_top_moments(dx + b0 * n, tab, n, D0, bc, acf, dsum, dsq, acc, st, !linear) : 0;
            if (fusedm < 0) rc = fusedm;
            if (fusedm == 0) {
                // three plain passes + the moment kernel.  With the first moments already made from the sum signal (linear) the sums of
                // this pass go to a scratch column that nobody reads (a decline here used to be a hard error)
                double *sums = dsum;
                if (linear) {
                    if (!sum_dump) sum_dump = (double *)scr.alloc(sizeof(double) * nk_top);
                    if (!sum_dump) return io.finish(WX_EHIP);
                    sums = sum_dump;
                }
                if (D0 > 0) rc = wx_dev_swt_fwd<double>(dx + b0 * n, tab, n, D0, LAYOUT_WPD, bc, filt, &acf, st);
                else WX_HIP_CHECK(hipMemcpyAsync(tab, dx + b0 * n, sizeof(double) * n * bc, hipMemcpyDeviceToDevice, st));
                if (rc == WX_OK) rc = wx_dev_jbb_moments<double>(tab, sums, dsq, nk_top, bc, acc, nullptr, 1, st);
            }
            if (rc == WX_OK)
                rc = wx_acwpd_mfma_ok(n, L, D0) ? wx_dev_acwpd_subtree_mfma(tab, dsum, dsq, n, L, D0, bc, acf, acc, st, !linear)
                                                : wx_dev_acwpd_subtree_moments(tab, dsum, dsq, n, L, D0, bc, acf, acc, st);
        }
        return io.finish(rc);
    }
    // fallback: materialise the table in chunks of <= 8 GiB
    int64_t chunk = ((int64_t)8 << 30) / (int64_t)(sizeof(double) * nk);
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    double *tab = batch ? (double *)scr.alloc(sizeof(double) * nk * chunk) : nullptr;
    if (batch && !tab) return io.finish(WX_EHIP);
    for (int64_t b0 = 0; b0 < batch && rc == WX_OK; b0 += chunk) {
        const int64_t bc = (batch - b0 < chunk) ? batch - b0 : chunk;
        rc = wx_dev_swt_fwd<double>(dx + b0 * n, tab, n, L, LAYOUT_WPD, bc, filt, &acf, st);
        if (rc == WX_OK)
            rc = wx_dev_jbb_moments<double>(tab, dsum, dsq, nk, bc, (accumulate || b0 > 0) ? 1 : 0, nullptr, 1, st);
    }
    return io.finish(rc);
}

extern "C" {

#define WX_FWD(name, layout, ac)                                                                                   \
    int name##_f64(const double *x, double *xw, int64_t n, int L, int64_t batch, const double *qmf, int F,         \
                   void *stream)                                                                                   \
    { return api_redundant_fwd<double>(x, xw, n, L, layout, ac, batch, qmf, F, stream); }
#define WX_FWD32(name, layout)                                                                                     \
    int name##_f32(const float *x, float *xw, int64_t n, int L, int64_t batch, const double *qmf, int F,           \
                   void *stream)                                                                                   \
    { return api_redundant_fwd<float>(x, xw, n, L, layout, false, batch, qmf, F, stream); }

WX_FWD(wx_sdwt1d, LAYOUT_DWT, false) WX_FWD32(wx_sdwt1d, LAYOUT_DWT)
WX_FWD(wx_swpt1d, LAYOUT_WPT, false) WX_FWD32(wx_swpt1d, LAYOUT_WPT)
WX_FWD(wx_swpd1d, LAYOUT_WPD, false) WX_FWD32(wx_swpd1d, LAYOUT_WPD)
WX_FWD(wx_acdwt1d, LAYOUT_DWT, true)
WX_FWD(wx_acwpt1d, LAYOUT_WPT, true)
WX_FWD(wx_acwpd1d, LAYOUT_WPD, true)

int wx_isdwt1d_f64(const double *xw, double *x, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{ return api_swt_inv<double>(xw, x, n, L + 1, L, LAYOUT_DWT, nullptr, 0, sm, batch, qmf, F, stream); }
int wx_isdwt1d_f32(const float *xw, float *x, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{ return api_swt_inv<float>(xw, x, n, L + 1, L, LAYOUT_DWT, nullptr, 0, sm, batch, qmf, F, stream); }
int wx_iswpt1d_f64(const double *xw, double *x, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{ return api_swt_inv<double>(xw, x, n, (int64_t)1 << L, L, LAYOUT_WPT, nullptr, 0, sm, batch, qmf, F, stream); }
int wx_iswpt1d_f32(const float *xw, float *x, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{ return api_swt_inv<float>(xw, x, n, (int64_t)1 << L, L, LAYOUT_WPT, nullptr, 0, sm, batch, qmf, F, stream); }
int wx_iswpd1d_f64(const double *xw, double *x, int64_t n, int64_t ncols, int L, const uint8_t *tree, int64_t ntree,
                   int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{ return api_swt_inv<double>(xw, x, n, ncols, L, LAYOUT_WPD, tree, ntree, sm, batch, qmf, F, stream); }
int wx_iswpd1d_f32(const float *xw, float *x, int64_t n, int64_t ncols, int L, const uint8_t *tree, int64_t ntree,
                   int64_t sm, int64_t batch, const double *qmf, int F, void *stream)
{ return api_swt_inv<float>(xw, x, n, ncols, L, LAYOUT_WPD, tree, ntree, sm, batch, qmf, F, stream); }

int wx_iacdwt1d_f64(const double *xw, double *x, int64_t n, int L, int64_t batch, void *stream)
{ return api_iac<double>(xw, x, n, L + 1, L, LAYOUT_DWT, nullptr, 0, batch, stream); }
int wx_iacwpt1d_f64(const double *xw, double *x, int64_t n, int L, int64_t batch, void *stream)
{ return api_iac<double>(xw, x, n, (int64_t)1 << L, L, LAYOUT_WPT, nullptr, 0, batch, stream); }
int wx_iacwpd1d_f64(const double *xw, double *x, int64_t n, int64_t ncols, int L, const uint8_t *tree, int64_t ntree,
                    int64_t batch, void *stream)
{ return api_iac<double>(xw, x, n, ncols, L, LAYOUT_WPD, tree, ntree, batch, stream); }

int wx_jbb_moments_f64(const double *X, double *sum, double *sumsq, int64_t nk, int64_t batch, int accumulate, void *stream)
{ return api_jbb_moments<double>(X, sum, sumsq, nk, batch, accumulate, stream); }
int wx_jbb_moments_f32(const float *X, float *sum, float *sumsq, int64_t nk, int64_t batch, int accumulate, void *stream)
{ return api_jbb_moments<float>(X, sum, sumsq, nk, batch, accumulate, stream); }
int wx_jbb_costs_f64(const double *sum, const double *sumsq, int64_t Ntot, int64_t n, int64_t k, int redundant,
                     int cost_kind, double p, double *costs, void *stream)
{ return api_jbb_costs<double>(sum, sumsq, Ntot, n, k, redundant, cost_kind, p, costs, stream); }
int wx_jbb_costs_f32(const float *sum, const float *sumsq, int64_t Ntot, int64_t n, int64_t k, int redundant,
                     int cost_kind, double p, float *costs, void *stream)
{ return api_jbb_costs<float>(sum, sumsq, Ntot, n, k, redundant, cost_kind, p, costs, stream); }
int wx_treeselect_f64(double *costs, int64_t k, int64_t n, int type_max, uint8_t *tree)
{ return api_treeselect<double>(costs, k, n, type_max, tree); }
int wx_treeselect_f32(float *costs, int64_t k, int64_t n, int type_max, uint8_t *tree)
{ return api_treeselect<float>(costs, k, n, type_max, tree); }
int wx_treeselect_gap_f64(double *costs, int64_t k, int64_t n, int type_max, uint8_t *tree, double *min_rel_gap)
{ WX_REQUIRE(min_rel_gap, WX_EARG, "NULL argument"); return api_treeselect<double>(costs, k, n, type_max, tree, min_rel_gap); }
int wx_treeselect_gap_f32(float *costs, int64_t k, int64_t n, int type_max, uint8_t *tree, double *min_rel_gap)
{ WX_REQUIRE(min_rel_gap, WX_EARG, "NULL argument"); return api_treeselect<float>(costs, k, n, type_max, tree, min_rel_gap); }
int wx_acwpd_jbb_moments_f64(const double *x, double *sum, double *sumsq, int64_t n, int L, int64_t batch,
                             const double *qmf, int F, int accumulate, void *stream)
{ return api_acwpd_jbb_moments(x, sum, sumsq, n, L, batch, qmf, F, accumulate, stream); }

}  // extern "C"
