// wx_api.hip -- C ABI (include/waveletsext_hip.h): library entry points and the 1-D decimated
// wavelet-packet family.  Argument checks mirror the reference's @assert / throw sites.
#include "../../include/waveletsext_hip.h"
#include "wx_host.h"
#include "wx_debug.h"
#include "wx_kernels.h"
#include "wx_lanetree.h"
#include "wx_lattice_dn_api.h"
#include <atomic>
#include <cstring>

const char *wx_err_cstr();
static std::atomic<int> g_force_generic{0};
int wx_force_generic() { return g_force_generic.load() == 1; }
int wx_skip_register_kernels() { return g_force_generic.load() == 2; }

extern "C" {

int wx_version(void) { return 100; /* 0.1.0 */ }
const char *wx_last_error(void) { return wx_err_cstr(); }
int wx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
void wx_debug_set_dispatch(int on) { g_force_generic.store(on == 2 ? 2 : (on ? 1 : 0)); }   // wx_debug.h, not the public ABI

}  // extern "C"

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

static int wx_need_device()
{
    if (wx_device_count() < 1) return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    return WX_OK;
}

// ---- wpd ----------------------------------------------------------------------------------
template <typename T>
static int api_wpd1d(const T *x, T *y, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(n >= 1 && batch >= 0, WX_EARG, "wpd: bad dimensions");
    WX_REQUIRE(0 <= L && L <= wx_maxtransformlevels(n), WX_EASSERT, "wpd!: 0 <= L <= maxtransformlevels(x) (DWT.jl:137)");
    WX_REQUIRE(n < ((int64_t)1 << 30), WX_EUNSUPPORTED, "wpd: signal length >= 2^30 not supported");
    if ((rc = wx_need_device())) return rc;
    hipStream_t st = wx_stream(stream);
    WxIO io(st);
    const T *dx = (const T *)io.in(x, sizeof(T) * n * batch);
    T *dy = (T *)io.out(y, sizeof(T) * n * (L + 1) * batch);
    if ((batch && n) && (!dx || !dy)) return io.finish(WX_EHIP);
    rc = wx_dev_wpd1d<T>(dx, dy, n, L, batch, filt, st, wx_force_generic());
    return io.finish(rc);
}

// resolves (L | tree) into the device-side description shared by wpt / iwpt / iwpd
struct WxTree1d {
    int Leff = 0;
    const uint8_t *dstatus = nullptr;   // device copy of the tree (nullptr = full tree)
    int64_t nstatus = 0;
    bool full = true;
};

static int wx_resolve_tree1d(int64_t n, int L, const uint8_t *tree, int64_t ntree, WxScratch &scr, WxTree1d *out,
                             const char *who)
{
    (void)who;
    WX_REQUIRE(wx_isdyadic(n), WX_EASSERT, "maketree/isvalidtree: signal length must be dyadic (Wavelets.jl)");
    if (!tree) {
        WX_REQUIRE(0 <= L && L <= wx_maxtransformlevels(n), WX_EASSERT, "maketree: 0 <= L <= maxtransformlevels(n)");
        out->Leff = L;
        out->full = true;
        return WX_OK;
    }
    WX_REQUIRE(wx_isvalidtree1d(n, tree, ntree), WX_EASSERT, "@assert isvalidtree(x, tree)");
    out->Leff = wx_tree_depth1d(tree, ntree);
    // a full tree of depth Leff needs no status bytes
    bool full = true;
    for (int64_t i = 1; i <= ((int64_t)1 << out->Leff) - 1 && full; ++i) full = tree[i - 1] != 0;
    out->full = full;
    if (!full) {
        out->dstatus = (const uint8_t *)scr.upload(tree, (size_t)ntree);
        if (!out->dstatus) return WX_EHIP;
        out->nstatus = ntree;
    }
    return WX_OK;
}

// tree = the pyramid of depth Ld and its deep levels can run lane-locally (wx_dwttail.hip): returns the number of those
// levels and the pyramid cut above them in `cut`; 0 otherwise.  The inverse feeds the 64 rebuilt samples to the loads of the
// fused tree-driven kernel: that kernel must apply, and the remaining pyramid must not be a full tree (other kernels).
template <typename T>
static int wx_pyramid_tail(int64_t n, int F, bool inverse, const uint8_t *tree, int64_t ntree, std::vector<uint8_t> &cut, const WxFilt &filt,
                           bool allow_long = false)
{
    if (!tree || wx_force_generic()) return 0;
    const int Ld = wx_tree_depth1d(tree, ntree);
    bool pyramid = Ld >= 1;
    for (int64_t i = 1; i <= ntree && pyramid; ++i) pyramid = (tree[i - 1] != 0) == ((i & (i - 1)) == 0 && i < ((int64_t)1 << Ld));
    if (!pyramid) return 0;
    const int tail = wx_dwt_tail_levels(n, Ld, F, sizeof(T));
    if (!tail || (inverse && !((wx_fused1d_ok<T>(n, F) || (allow_long && wx_dwt_long_ok<T>(n, filt))) && Ld - tail >= 2))) return 0;
    cut.assign(tree, tree + ntree);
    for (int64_t i = (int64_t)1 << (Ld - tail); i <= ntree; ++i) cut[i - 1] = 0;
    return tail;
}

// ---- wpt / iwpt ---------------------------------------------------------------------------
template <typename T, bool INVERSE>
static int api_wpt1d(const T *x, T *y, int64_t n, int L, const uint8_t *tree, int64_t ntree, int64_t batch,
                     const double *qmf, int F, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(n >= 1 && batch >= 0, WX_EARG, "wpt: bad dimensions");
    WX_REQUIRE(n < ((int64_t)1 << 30), WX_EUNSUPPORTED, "wpt: signal length >= 2^30 not supported");
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxTree1d tr;
    // argument errors are reported before any device is needed
    if (tree) WX_REQUIRE(wx_isdyadic(n) && wx_isvalidtree1d(n, tree, ntree), WX_EASSERT, "@assert isvalidtree(x, tree)");
    else WX_REQUIRE(wx_isdyadic(n) && 0 <= L && L <= wx_maxtransformlevels(n), WX_EASSERT,
                    "maketree: isdyadic(n) and 0 <= L <= maxtransformlevels(n)");
    if ((rc = wx_need_device())) return rc;
    // the pyramid (dwt / dwtall): the levels from 64 samples down run in the registers of a lane (wx_dwttail.hip)
    std::vector<uint8_t> ttree;
    // (measured, 65536 x 4096 Float64 db4, depth-12 pyramid: the whole pyramid through the tree-driven lattice kernels 1.22 /
    // 1.51 ms, with the tail 1.15 / 1.00 ms -- the tail stays; the levels above it take the lattice in the forward direction)
    // short signals take the small-signal kernel whole (wx_smalltree.hip), pyramids included: no tail cut for them
    bool is_pyr = false;
    if (tree) {
        const int Ld = wx_tree_depth1d(tree, ntree);
        is_pyr = Ld >= 1;
        for (int64_t i = 1; i <= ntree && is_pyr; ++i) is_pyr = (tree[i - 1] != 0) == ((i & (i - 1)) == 0 && i < ((int64_t)1 << Ld));
    }
    // 64 .. 512 samples along a tree: the masked lattice kernels take the whole tree (wx_lattice_tree_s.h), pyramids included -- no tail cut
    const bool lat_short = tree && !wx_force_generic() && !wx_skip_register_kernels() && n >= 64 && n <= 512 && batch >= 4096 / n && x != y &&
                           (sizeof(T) == 8 ? wx_lattice_tree_applicable_f64(n, filt) : wx_lattice_tree_applicable_f32(n, filt));
    const bool small_pyr = is_pyr && !wx_force_generic() && (lat_short || wx_small_tree_wanted<T>(n, F, true, true));
    const int tail = small_pyr ? 0 : wx_pyramid_tail<T>(n, F, INVERSE, tree, ntree, ttree, filt, true);
    if (tail) tree = ttree.data();
    if ((rc = wx_resolve_tree1d(n, L, tree, ntree, scr, &tr, "wpt"))) return rc;
    const bool small = !wx_force_generic() && !tail && wx_small_tree_wanted<T>(n, F, !tr.full, is_pyr);
    // the pyramid of a long signal (dwt / idwt of 16384 .. 65536 samples): tiled top levels + the lattice (wx_dev_dwt_long)
    bool longp = false;
    if (tree && tr.dstatus && tr.Leff >= 1 && !wx_force_generic() && wx_dwt_long_ok<T>(n, filt)) {
        longp = true;
        for (int64_t i = 1; i <= ntree && longp; ++i) longp = (tree[i - 1] != 0) == ((i & (i - 1)) == 0 && i < ((int64_t)1 << tr.Leff));
    }
    WxIO io(st);
    const T *dx = (const T *)io.in(x, sizeof(T) * n * batch);
    T *dy = (T *)io.out(y, sizeof(T) * n * batch);
    if ((batch && n) && (!dx || !dy)) return io.finish(WX_EHIP);
    const int force = wx_force_generic();
    // Float32 FULL trees of 128 / 256 samples: the masked tree kernels in Float32 arithmetic on pairs of signals (wx_lattice_tree_s.h) beat the
    // interleaved full-tree kernels (256 samples: depth 1 0.53 -> 0.41 ms per GiB, depth 8 0.55 -> 0.46): taken as a tree of ones
    static const bool f32tree_off = wx_getenv("WX_TREES32_FULL") && atoi(wx_getenv("WX_TREES32_FULL")) == 0;
    const bool f32_full_as_tree = sizeof(T) == 4 && !f32tree_off && tr.full && (n == 256 || (n == 128 && tr.Leff >= 2)) && F <= 8 && tr.Leff >= 1;   // (128 samples, depth 1: 0.38 against 0.41)
    if constexpr (sizeof(T) == 4) {
        // Float32 full trees of 64 .. 2048 samples: the interleaved lattice kernels where they apply (wx_lattice_sg32.h)
        if (small && tr.full && !f32_full_as_tree && tr.Leff >= 1 && batch && dx != dy && !wx_skip_register_kernels()) {
            const int r = wx_lattice_f32(INVERSE, (const float *)dx, (float *)dy, n, tr.Leff, batch, n, filt, st);
            if (r) return io.finish(r < 0 ? r : WX_OK);
        }
    }
    // short signals along a tree (pyramids, best bases, ...): 8 .. 64 signals in the registers of a wavefront, every level under the tree's
    // masks (wx_lattice_tree_s.h); longer filters and odd geometries go on to the kernels below
    // (a FULL tree with a filter of 10 ... 16 taps, which the interleaved full-tree kernels are not built for, goes the same way as the tree
    // that happens to be full: 64 samples, db8, depth 1: 1.16 -> 0.4 ms per GiB)
    const bool full_long = tr.full && (F > 8 || f32_full_as_tree) && n >= 64 && n <= 512 && batch >= 4096 / n && !wx_force_generic() &&
                           (sizeof(T) == 8 ? wx_lattice_tree_applicable_f64(n, filt) : wx_lattice_tree_applicable_f32(n, filt));
    if ((small || lat_short || full_long) && !tail && (!tr.full || full_long) && tr.Leff >= 1 && batch && dx != dy && !wx_skip_register_kernels()) {
        const uint8_t *ds = tr.dstatus;
        int64_t ns = tr.nstatus;
        std::vector<uint8_t> ones;
        if (tr.full) {
            ones.assign(((size_t)1 << tr.Leff) - 1, (uint8_t)1);
            ds = (const uint8_t *)scr.upload(ones.data(), ones.size());
            ns = (int64_t)ones.size();
        }
        int r = 0;
        if (ds) {
            if constexpr (sizeof(T) == 8)
                r = wx_lattice_tree_f64(INVERSE, (const double *)dx, (double *)dy, n, tr.Leff, batch, n, 0, filt, ds, ns, st, nullptr, 0);
            else
                r = wx_lattice_tree_f32(INVERSE, (const float *)dx, (float *)dy, n, tr.Leff, batch, n, filt, ds, ns, st, nullptr, 0);
        }
        if (r) return io.finish(r < 0 ? r : WX_OK);
    }
    if (small && tr.Leff >= 1 && batch && dx != dy) {
        static const bool lane_off = wx_getenv("WX_LANETREE") && atoi(wx_getenv("WX_LANETREE")) == 0;
        if (!lane_off && (n <= 64 || (n <= 128 && sizeof(T) == 4))) {
            // one lane per signal: the tree as a bit mask (at most 255 nodes of depth < Leff)
            WxLaneTree lt;
            memset(&lt, 0, sizeof lt);
            const int64_t nn = ((int64_t)1 << tr.Leff) - 1;
            for (int64_t h = 0; h < nn; ++h)
                if (tr.full || (h < ntree && tree[h])) lt.bits[h >> 5] |= 1u << (h & 31);
            if constexpr (sizeof(T) == 8)
                return io.finish(wx_lane_tree_f64(INVERSE, (const double *)dx, (double *)dy, n, tr.Leff, batch, lt, filt, st));
            else
                return io.finish(wx_lane_tree_f32(INVERSE, (const float *)dx, (float *)dy, n, tr.Leff, batch, lt, filt, st));
        }
        return io.finish(wx_dev_small_tree<T>(INVERSE, dx, dy, n, tr.Leff, batch, filt, tr.dstatus, tr.nstatus, st));
    }
    T *s1 = nullptr;
    const bool fused = !force && wx_fused1d_ok<T>(n, F);
    // long Float64 signals with a full tree take one pass per top level and then the lattice kernels: one scratch array
    const bool long_lattice = !force && !tr.dstatus && n > 4096 && n <= 65536 && tr.Leff > 1;
    // any other tree on a long Float64 signal: tiled top levels node by node + one lattice launch per 4096-sample subtree
    const bool longt = !longp && tree && tr.dstatus && tr.Leff >= 1 && !force && wx_wpt_long_tree_ok<T>(n, filt) && dx != dy;
    if (((!fused && tr.Leff > 1) || long_lattice || longp || longt) && batch) {
        s1 = (T *)scr.alloc(sizeof(T) * n * batch);
        if (!s1) return io.finish(WX_EHIP);
    }
    if (longt && batch) return io.finish(wx_dev_wpt_long_tree<T>(dx, dy, n, tr.Leff, batch, filt, tree, ntree, s1, INVERSE, st));
    if (longp && batch && dx != dy) {
        if (INVERSE) {
            WxThreshArg thr{nullptr, 0, 0, 0, 1.0};
            if (tail) {
                T *head = (T *)scr.alloc(sizeof(T) * 64 * batch);
                if (!head) return io.finish(WX_EHIP);
                if ((rc = wx_idwt_tail<T>(dx, head, n, tail, batch, filt, thr, st))) return io.finish(rc);
                thr.head = head;
            }
            rc = wx_dev_idwt_long<T>(dx, dy, n, tr.Leff, batch, filt, tr.dstatus, tr.nstatus, thr, s1, st);
        } else {
            rc = wx_dev_dwt_long<T>(dx, dy, n, tr.Leff, batch, filt, tr.dstatus, tr.nstatus, s1, st);
            if (rc == WX_OK && tail) rc = wx_dwt_tail<T>(dy, n, tail, batch, filt, st);
        }
        return io.finish(rc);
    }
    if (INVERSE && tail && batch) {
        T *head = (T *)scr.alloc(sizeof(T) * 64 * batch);
        if (!head) return io.finish(WX_EHIP);
        WxThreshArg thr{nullptr, 0, 0, 0, 1.0};
        rc = wx_idwt_tail<T>(dx, head, n, tail, batch, filt, thr, st);
        thr.head = head;
        if (rc == WX_OK) rc = wx_dev_iwpt1d_thresh<T>(dx, dy, n, tr.Leff, batch, filt, tr.dstatus, tr.nstatus, thr, st);
    } else if (INVERSE)
        rc = wx_dev_iwpt1d<T>(dx, dy, n, tr.Leff, batch, filt, tr.dstatus, tr.nstatus, nullptr, 0, n, s1, nullptr, st, force);
    else
        rc = wx_dev_wpt1d<T>(dx, dy, n, tr.Leff, batch, filt, tr.dstatus, tr.nstatus, s1, st, force);
    if (rc == WX_OK && tail && !INVERSE) rc = wx_dwt_tail<T>(dy, n, tail, batch, filt, st);
    return io.finish(rc);
}

// ---- iwpt with the threshold of denoise() in its load stage ------------------------------------------------
template <typename T>
static int api_iwpt1d_thresh(const T *x, T *y, int64_t n, int L, const uint8_t *tree, int64_t ntree, int64_t batch, const double *qmf,
                             int F, int th_kind, const T *t, int64_t nt, int64_t row_lo, double scale, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(n >= 1 && batch >= 0, WX_EARG, "iwpt: bad dimensions");
    WX_REQUIRE(n < ((int64_t)1 << 30), WX_EUNSUPPORTED, "iwpt: signal length >= 2^30 not supported");
    WX_REQUIRE(th_kind >= 0 && th_kind <= 3, WX_EARG, "th_kind: 0 HardTH, 1 SoftTH, 2 SemiSoftTH, 3 SteinTH");
    WX_REQUIRE(t != nullptr && (nt == 1 || nt == batch), WX_EARG, "one threshold, or one per signal");
    WX_REQUIRE(0 <= row_lo && row_lo <= n, WX_EBOUNDS, "row range outside the array");
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxTree1d tr;
    if (tree) WX_REQUIRE(wx_isdyadic(n) && wx_isvalidtree1d(n, tree, ntree), WX_EASSERT, "@assert isvalidtree(x, tree)");
    else WX_REQUIRE(wx_isdyadic(n) && 0 <= L && L <= wx_maxtransformlevels(n), WX_EASSERT,
                    "maketree: isdyadic(n) and 0 <= L <= maxtransformlevels(n)");
    if ((rc = wx_need_device())) return rc;
    std::vector<uint8_t> ttree;
    // 64 ... 512 samples: the masked lattice kernels take the whole tree (wx_lattice_tree_s.h), without a threshold in their loads -- a
    // thresholded copy first, no lane-local tail (the fused kernel with the threshold took 4.1 ms per GiB of 64-sample signals against 0.8)
    const bool shortlat = tree && !wx_force_generic() && !wx_skip_register_kernels() && n >= 64 && n <= 512 && batch >= 4096 / n &&
                          (const void *)x != (const void *)y &&
                          (sizeof(T) == 8 ? wx_lattice_tree_applicable_f64(n, filt) : wx_lattice_tree_applicable_f32(n, filt));
    const int tail = shortlat ? 0 : wx_pyramid_tail<T>(n, F, true, tree, ntree, ttree, filt);   // denoise(:dwt): the pyramid's deep levels lane-locally
    if (tail) tree = ttree.data();
    if ((rc = wx_resolve_tree1d(n, L, tree, ntree, scr, &tr, "iwpt"))) return rc;
    WxIO io(st);
    const T *dx = (const T *)io.in(x, sizeof(T) * n * batch);
    T *dy = (T *)io.out(y, sizeof(T) * n * batch);
    const T *dt = (const T *)io.in(t, sizeof(T) * nt);
    if ((batch && n) && (!dx || !dy || !dt)) return io.finish(WX_EHIP);
    if (batch == 0) return io.finish(WX_OK);
    const int per = nt == batch && batch > 1 ? 1 : 0;
    if (tr.Leff >= 1 && !wx_force_generic() && !shortlat && wx_iwpt1d_thresh_fusable<T>(n, F, tr.dstatus)) {
        WxThreshArg thr{dt, th_kind, (int)row_lo, per, scale};
        if (tail) {
            T *head = (T *)scr.alloc(sizeof(T) * 64 * batch);
            if (!head) return io.finish(WX_EHIP);
            if ((rc = wx_idwt_tail<T>(dx, head, n, tail, batch, filt, thr, st))) return io.finish(rc);
            thr.head = head;
        }
        return io.finish(wx_dev_iwpt1d_thresh<T>(dx, dy, n, tr.Leff, batch, filt, tr.dstatus, tr.nstatus, thr, st));
    }
    if (tail) return io.finish(wx_set_error(WX_EHIP, "iwpt: inconsistent pyramid plan"));
    // other kernels (full trees on the lattice, long signals, L = 0): threshold into a scratch copy, then the inverse
    T *xt = (T *)scr.alloc(sizeof(T) * n * batch);
    if (!xt) return io.finish(WX_EHIP);
    if ((rc = wx_dev_threshold_copy<T>(dx, xt, n, batch, th_kind, dt, per, row_lo, scale, st))) return io.finish(rc);
    const int force = wx_force_generic();
    T *s1 = nullptr;
    if (!(!force && wx_fused1d_ok<T>(n, F)) && tr.Leff > 1) {
        s1 = (T *)scr.alloc(sizeof(T) * n * batch);
        if (!s1) return io.finish(WX_EHIP);
    }
    // the pyramid of a long signal (denoise(:dwt) of 16384 ... 65536 samples): the tiled top levels + the lattice (wx_dev_idwt_long), as idwt takes
    // it -- level by level it was 14 launches and 7.9 ms per GiB
    if (tree && tr.dstatus && tr.Leff >= 1 && !force && s1 && wx_dwt_long_ok<T>(n, filt)) {
        bool longp = true;
        for (int64_t i = 1; i <= ntree && longp; ++i) longp = (tree[i - 1] != 0) == ((i & (i - 1)) == 0 && i < ((int64_t)1 << tr.Leff));
        if (longp) {
            const WxThreshArg none{nullptr, 0, 0, 0, 1.0};
            return io.finish(wx_dev_idwt_long<T>(xt, dy, n, tr.Leff, batch, filt, tr.dstatus, tr.nstatus, none, s1, st));
        }
    }
    rc = wx_dev_iwpt1d<T>(xt, dy, n, tr.Leff, batch, filt, tr.dstatus, tr.nstatus, nullptr, 0, n, s1, nullptr, st, force);
    return io.finish(rc);
}

// ---- denoiseall(x, :sig, wt; L, dnt, estnoise = noisest, smooth) Denoising.jl:651-712 ------------------------------------------
// One pass over the signals where the lattice kernel applies (wx_lattice_dn.h: Float64, 64 ... 4096 samples, up to 8 taps, Hard / Soft /
// SemiSoft), else the three steps the reference takes -- dwtall, noisest per signal, threshold on the loads of idwtall -- with the
// coefficients and the estimates in stream-ordered scratch.  sigma (optional, device or host): the noise estimates.
extern "C" int wx_noisest_f64(const double *X, int64_t n, int64_t k, int64_t batch, int64_t row_lo, int64_t col, double *sigma, void *stream);
extern "C" int wx_noisest_f32(const float *X, int64_t n, int64_t k, int64_t batch, int64_t row_lo, int64_t col, float *sigma, void *stream);
// COEFS: x holds the coefficients dwtall(., wt, L) (inputtype :dwt) instead of the signals
template <typename T, bool COEFS>
static int api_denoiseall_sig(const T *x, T *y, int64_t n, int L, int64_t batch, const double *qmf, int F, int th_kind, double tscale,
                              int undersmooth, T *sigma, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(n >= 1 && batch >= 0, WX_EARG, "denoiseall: bad dimensions");
    WX_REQUIRE(n < ((int64_t)1 << 30), WX_EUNSUPPORTED, "denoiseall: signal length >= 2^30 not supported");
    WX_REQUIRE(th_kind >= 0 && th_kind <= 3, WX_EARG, "th_kind: 0 HardTH, 1 SoftTH, 2 SemiSoftTH, 3 SteinTH");
    WX_REQUIRE(wx_isdyadic(n) && n >= 2, WX_EASSERT, "@assert isdyadic(size(y,1))");                    // noisest, Denoising.jl:218
    WX_REQUIRE(0 <= L && L <= wx_maxtransformlevels(n), WX_EASSERT, "maketree: isdyadic(n) and 0 <= L <= maxtransformlevels(n)");
    if ((rc = wx_need_device())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dx = (const T *)io.in(x, sizeof(T) * n * batch);
    T *dy = (T *)io.out(y, sizeof(T) * n * batch);
    T *dsig = sigma ? (T *)io.out(sigma, sizeof(T) * batch) : nullptr;
    if (batch && (!dx || !dy || (sigma && !dsig))) return io.finish(WX_EHIP);
    if (batch == 0) return io.finish(WX_OK);
    if constexpr (sizeof(T) == 8) {
        static const bool off = wx_getenv("WX_DENOISE_ONEPASS") && atoi(wx_getenv("WX_DENOISE_ONEPASS")) == 0;
        if (!off && !wx_force_generic() && !wx_skip_register_kernels() && th_kind != 3 && L >= 1 && wx_lattice_applicable_f64(filt)) {
            int r = 0;
            switch (n) {
#define WX_DN_CASE(k) case 4096 >> k: r = wx_lattice_denoise##k##_f64(dx, dy, n, L, batch, filt, th_kind, tscale, undersmooth, dsig, COEFS ? 1 : 0, st); break;
                WX_DN_CASE(0) WX_DN_CASE(1) WX_DN_CASE(2) WX_DN_CASE(3) WX_DN_CASE(4) WX_DN_CASE(5) WX_DN_CASE(6)
#undef WX_DN_CASE
            default: break;
            }
            if (r) return io.finish(r < 0 ? r : WX_OK);
        }
    }
    // the separate steps: every one of them takes device pointers as they are
    T *w = COEFS ? const_cast<T *>(dx) : (T *)scr.alloc(sizeof(T) * n * batch);
    if (!w) return io.finish(WX_EHIP);
    if (!dsig) {
        dsig = (T *)scr.alloc(sizeof(T) * batch);
        if (!dsig) return io.finish(WX_EHIP);
    }
    std::vector<uint8_t> tree((size_t)(n > 1 ? n - 1 : 1), 0);
    for (int d = 0; d < L; ++d) tree[((size_t)1 << d) - 1] = 1;
    const int64_t ntree = n - 1;
    rc = COEFS ? WX_OK : api_wpt1d<T, false>(dx, w, n, 0, tree.data(), ntree, batch, qmf, F, stream);
    if (rc == WX_OK) {
        if constexpr (sizeof(T) == 8) rc = wx_noisest_f64(w, n, 1, batch, n / 2, 0, dsig, stream);
        else rc = wx_noisest_f32(w, n, 1, batch, n / 2, 0, dsig, stream);
    }
    if (rc == WX_OK)
        rc = api_iwpt1d_thresh<T>(w, dy, n, 0, tree.data(), ntree, batch, qmf, F, th_kind, dsig, batch, undersmooth ? (n >> L) : 0, tscale, stream);
    return io.finish(rc);
}

// ---- iwpd ---------------------------------------------------------------------------------
template <typename T>
static int api_iwpd1d(const T *xw, T *xh, int64_t n, int k, int L, const uint8_t *tree, int64_t ntree, int64_t batch,
                      const double *qmf, int F, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(n >= 1 && batch >= 0 && k >= 1, WX_EARG, "iwpd: bad dimensions");
    WX_REQUIRE(n < ((int64_t)1 << 30), WX_EUNSUPPORTED, "iwpd: signal length >= 2^30 not supported");
    if (tree) WX_REQUIRE(wx_isdyadic(n) && wx_isvalidtree1d(n, tree, ntree), WX_EASSERT, "@assert isvalidtree(x̂, tree) (DWT.jl:346)");
    else WX_REQUIRE(wx_isdyadic(n) && 0 <= L && L <= wx_maxtransformlevels(n), WX_EASSERT,
                    "maketree: isdyadic(n) and 0 <= L <= maxtransformlevels(n)");
    WX_REQUIRE(k - 1 <= wx_maxtransformlevels(n), WX_EASSERT, "getbasiscoef: @assert k-1 <= L (Utils.jl:110)");
    const int Leff = tree ? wx_tree_depth1d(tree, ntree) : L;
    WX_REQUIRE(Leff < k, WX_EARG, "getbasiscoef: Not enough decomposition levels in Xw (Utils.jl:120)");
    if ((rc = wx_need_device())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxTree1d tr;
    if ((rc = wx_resolve_tree1d(n, L, tree, ntree, scr, &tr, "iwpd"))) return rc;
    WxIO io(st);
    const T *dxw = (const T *)io.in(xw, sizeof(T) * n * k * batch);
    T *dxh = (T *)io.out(xh, sizeof(T) * n * batch);
    if ((batch && n) && (!dxw || !dxh)) return io.finish(WX_EHIP);
    const int force = wx_force_generic();
    const bool fused = !force && wx_fused1d_ok<T>(n, F);
    const int *dcol = nullptr;
    int log2blk = 0;
    if (!tr.full) {
        std::vector<int> col;
        wx_leaf_colmap1d(tree, ntree, tr.Leff, col);
        for (int64_t v = n >> tr.Leff; v > 1; v >>= 1) ++log2blk;
        dcol = (const int *)scr.upload(col.data(), col.size() * sizeof(int));
        if (!dcol) return io.finish(WX_EHIP);
    }
    T *s1 = nullptr, *s2 = nullptr;
    if (!fused && batch) {
        const bool want_s2 = dcol || (tr.full && tr.Leff > 1);      // gathered leaves, or the densified leaf column
        if (want_s2) s2 = (T *)scr.alloc(sizeof(T) * n * batch);
        if (tr.Leff > 1) s1 = (T *)scr.alloc(sizeof(T) * n * batch);
        if ((want_s2 && !s2) || (tr.Leff > 1 && !s1)) return io.finish(WX_EHIP);
    }
    // full tree: every leaf sits in column Leff -> shift the base pointer, keep the table stride
    const T *base = tr.full ? dxw + (int64_t)tr.Leff * n : dxw;
    rc = wx_dev_iwpt1d<T>(base, dxh, n, tr.Leff, batch, filt, tr.dstatus, tr.nstatus, dcol, log2blk, n * (int64_t)k,
                          s1, s2, st, force);
    return io.finish(rc);
}

template <typename T>
static int api_getbasiscoef1d(const T *Xw, T *out, int64_t n, int k, const uint8_t *tree, int64_t ntree, int64_t batch,
                              void *stream)
{
    WX_REQUIRE(n >= 1 && batch >= 0 && k >= 1, WX_EARG, "getbasiscoef: bad dimensions");
    WX_REQUIRE(wx_isvalidtree1d(n, tree, ntree), WX_EASSERT, "@assert isvalidtree(x, tree) (Utils.jl:109)");
    WX_REQUIRE(k - 1 <= wx_maxtransformlevels(n), WX_EASSERT, "@assert k-1 <= L (Utils.jl:110)");
    WX_REQUIRE(wx_isdyadic(n), WX_EASSERT, "@assert leaf_len == length(leaf) (Utils.jl:113)");
    const int Leff = wx_tree_depth1d(tree, ntree);
    WX_REQUIRE(Leff < k, WX_EARG, "Not enough decomposition levels in Xw (Utils.jl:120)");
    int rc;
    if ((rc = wx_need_device())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    std::vector<int> col;
    wx_leaf_colmap1d(tree, ntree, Leff, col);
    const int *dcol = (const int *)scr.upload(col.data(), col.size() * sizeof(int));
    if (!dcol) return WX_EHIP;
    WxIO io(st);
    const T *dX = (const T *)io.in(Xw, sizeof(T) * n * k * batch);
    T *dout = (T *)io.out(out, sizeof(T) * n * batch);
    if ((batch && n) && (!dX || !dout)) return io.finish(WX_EHIP);
    rc = wx_dev_getbasiscoef1d<T>(dX, dout, n, k, batch, dcol, (int)(n >> Leff), st);
    return io.finish(rc);
}

extern "C" {

int wx_wpd1d_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream)
{ return api_wpd1d<double>(x, y, n, L, batch, qmf, F, stream); }
int wx_wpd1d_f32(const float *x, float *y, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream)
{ return api_wpd1d<float>(x, y, n, L, batch, qmf, F, stream); }

int wx_wpt1d_f64(const double *x, double *y, int64_t n, int L, const uint8_t *tree, int64_t ntree, int64_t batch,
                 const double *qmf, int F, void *stream)
{ return api_wpt1d<double, false>(x, y, n, L, tree, ntree, batch, qmf, F, stream); }
int wx_wpt1d_f32(const float *x, float *y, int64_t n, int L, const uint8_t *tree, int64_t ntree, int64_t batch,
                 const double *qmf, int F, void *stream)
{ return api_wpt1d<float, false>(x, y, n, L, tree, ntree, batch, qmf, F, stream); }

int wx_iwpt1d_f64(const double *xw, double *xhat, int64_t n, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream)
{ return api_wpt1d<double, true>(xw, xhat, n, L, tree, ntree, batch, qmf, F, stream); }
int wx_iwpt1d_f32(const float *xw, float *xhat, int64_t n, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream)
{ return api_wpt1d<float, true>(xw, xhat, n, L, tree, ntree, batch, qmf, F, stream); }

int wx_iwpt1d_thresh_f64(const double *xw, double *xhat, int64_t n, int L, const uint8_t *tree, int64_t ntree, int64_t batch,
                         const double *qmf, int F, int th_kind, const double *t, int64_t nt, int64_t row_lo, double scale,
                         void *stream)
{ return api_iwpt1d_thresh<double>(xw, xhat, n, L, tree, ntree, batch, qmf, F, th_kind, t, nt, row_lo, scale, stream); }
int wx_iwpt1d_thresh_f32(const float *xw, float *xhat, int64_t n, int L, const uint8_t *tree, int64_t ntree, int64_t batch,
                         const double *qmf, int F, int th_kind, const float *t, int64_t nt, int64_t row_lo, double scale,
                         void *stream)
{ return api_iwpt1d_thresh<float>(xw, xhat, n, L, tree, ntree, batch, qmf, F, th_kind, t, nt, row_lo, scale, stream); }

int wx_denoiseall_sig_f64(const double *x, double *xhat, int64_t n, int L, int64_t batch, const double *qmf, int F, int th_kind, double t,
                          int undersmooth, double *sigma, void *stream)
{ return api_denoiseall_sig<double, false>(x, xhat, n, L, batch, qmf, F, th_kind, t, undersmooth, sigma, stream); }
int wx_denoiseall_sig_f32(const float *x, float *xhat, int64_t n, int L, int64_t batch, const double *qmf, int F, int th_kind, double t,
                          int undersmooth, float *sigma, void *stream)
{ return api_denoiseall_sig<float, false>(x, xhat, n, L, batch, qmf, F, th_kind, t, undersmooth, sigma, stream); }
int wx_denoiseall_dwt_f64(const double *xw, double *xhat, int64_t n, int L, int64_t batch, const double *qmf, int F, int th_kind, double t,
                          int undersmooth, double *sigma, void *stream)
{ return api_denoiseall_sig<double, true>(xw, xhat, n, L, batch, qmf, F, th_kind, t, undersmooth, sigma, stream); }
int wx_denoiseall_dwt_f32(const float *xw, float *xhat, int64_t n, int L, int64_t batch, const double *qmf, int F, int th_kind, double t,
                          int undersmooth, float *sigma, void *stream)
{ return api_denoiseall_sig<float, true>(xw, xhat, n, L, batch, qmf, F, th_kind, t, undersmooth, sigma, stream); }

int wx_iwpd1d_f64(const double *xw, double *xhat, int64_t n, int k, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream)
{ return api_iwpd1d<double>(xw, xhat, n, k, L, tree, ntree, batch, qmf, F, stream); }
int wx_iwpd1d_f32(const float *xw, float *xhat, int64_t n, int k, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream)
{ return api_iwpd1d<float>(xw, xhat, n, k, L, tree, ntree, batch, qmf, F, stream); }

int wx_getbasiscoef1d_f64(const double *Xw, double *out, int64_t n, int k, const uint8_t *tree, int64_t ntree,
                          int64_t batch, void *stream)
{ return api_getbasiscoef1d<double>(Xw, out, n, k, tree, ntree, batch, stream); }
int wx_getbasiscoef1d_f32(const float *Xw, float *out, int64_t n, int k, const uint8_t *tree, int64_t ntree,
                          int64_t batch, void *stream)
{ return api_getbasiscoef1d<float>(Xw, out, n, k, tree, ntree, batch, stream); }

}  // extern "C"
