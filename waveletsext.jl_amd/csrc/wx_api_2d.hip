// wx_api_2d.hip -- C ABI for the 2-D decimated wavelet-packet family (quad trees).
#include "../../include/waveletsext_hip.h"
#include "wx_host.h"
#include "wx_kernels.h"

int wx_force_generic();

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

static int wx_need_device3()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    }
    return WX_OK;
}

// leaf depth of every (2^Leff x 2^Leff) block, quad-tree traversal of getbasiscoef (Utils.jl:117-131)
// 2-D wpd of 64 x 64 images in one pass (wx_lattice_2d64w.hip); 0 = not applicable, 1 = launched, < 0 = error
int wx_lattice_2d64_wpd_f64(const double *x, double *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st);
int wx_lattice_2d64_wpd_f32(const float *x, float *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st);
static inline int wx_lattice_2d64_wpd(const double *x, double *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st) { return wx_lattice_2d64_wpd_f64(x, y, L, batch, filt, st); }
static inline int wx_lattice_2d64_wpd(const float *x, float *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st) { return wx_lattice_2d64_wpd_f32(x, y, L, batch, filt, st); }
// 64 x 64 images along any quad tree in one pass (wx_lattice_2d64t.h); 0 = not applicable, 1 = launched, < 0 = error
int wx_lattice_2d64t_fwd_f64(const double *, double *, int, int64_t, const WxFilt &, const uint8_t *, int64_t, hipStream_t);
int wx_lattice_2d64t_fwd_f32(const float *, float *, int, int64_t, const WxFilt &, const uint8_t *, int64_t, hipStream_t);
int wx_lattice_2d64t_inv_f64(const double *, double *, int, int64_t, int64_t, const WxFilt &, const uint8_t *, int64_t, hipStream_t);
int wx_lattice_2d64t_inv_f32(const float *, float *, int, int64_t, int64_t, const WxFilt &, const uint8_t *, int64_t, hipStream_t);
static inline int wx_lattice_2d64t(bool inverse, const double *x, double *y, int L, int64_t batch, int64_t in_img, const WxFilt &filt,
                                   const uint8_t *ds, int64_t ns, hipStream_t st)
{
    if (inverse) return wx_lattice_2d64t_inv_f64(x, y, L, batch, in_img, filt, ds, ns, st);
    return in_img == 4096 ? wx_lattice_2d64t_fwd_f64(x, y, L, batch, filt, ds, ns, st) : 0;
}
static inline int wx_lattice_2d64t(bool inverse, const float *x, float *y, int L, int64_t batch, int64_t in_img, const WxFilt &filt,
                                   const uint8_t *ds, int64_t ns, hipStream_t st)
{
    if (inverse) return wx_lattice_2d64t_inv_f32(x, y, L, batch, in_img, filt, ds, ns, st);
    return in_img == 4096 ? wx_lattice_2d64t_fwd_f32(x, y, L, batch, filt, ds, ns, st) : 0;
}

static void wx_colmap2d_rec(const uint8_t *tree, int64_t ntree, int64_t node, int d, int j, int k, int Leff,
                            std::vector<int> &col)
{
    const int nblk = 1 << Leff;
    if (node <= ntree && tree[node - 1]) {
        for (int c = 0; c < 4; ++c)
            wx_colmap2d_rec(tree, ntree, 4 * node - 2 + c, d + 1, 2 * j + (c >> 1), 2 * k + (c & 1), Leff, col);
    } else {
        const int w = 1 << (Leff - d);
        for (int r = j * w; r < (j + 1) * w; ++r)
            for (int c = k * w; c < (k + 1) * w; ++c) col[(size_t)r * nblk + c] = d;
    }
}

struct WxTree2d {
    int Leff = 0;
    bool full = true;
    const uint8_t *dstatus = nullptr;
    int64_t nstatus = 0;
    const uint8_t *htree = nullptr;           // host copy (the caller's), for the per-depth lists of decomposed nodes
};

static int wx_check_tree2d(int64_t m, int64_t n, int L, const uint8_t *tree, int64_t ntree)
{
    const int L0 = wx_maxtransformlevels(m < n ? m : n);
    // the reference bounds L by maxtransformlevels(min(m, n)) only; a side that is not divisible by 2^depth
    // then trips the size asserts of its 2-D dwt_step! (dwt_one_level.jl:319-354) -- rejected here up front
    const int Lm = wx_maxtransformlevels(m), Ln = wx_maxtransformlevels(n);
    const int Lside = Lm < Ln ? Lm : Ln;
    if (!tree) {
        WX_REQUIRE(0 <= L && L <= L0, WX_EASSERT, "maketree(n, m, L): @assert 0 <= L <= L0 (utils_tree.jl:196)");
        WX_REQUIRE(L <= Lside, WX_EASSERT, "both sides must be divisible by 2^L (dwt_step! size asserts)");
        return WX_OK;
    }
    WX_REQUIRE(wx_isvalidtree2d(m, n, tree, ntree), WX_EASSERT, "@assert isvalidtree(x, tree) (DWT.jl:504,666)");
    WX_REQUIRE(wx_tree_depth2d(tree, ntree) <= Lside, WX_EASSERT,
               "both sides must be divisible by 2^depth(tree) (dwt_step! size asserts)");
    return WX_OK;
}

static int wx_resolve_tree2d(int L, const uint8_t *tree, int64_t ntree, WxScratch &scr, WxTree2d *out)
{
    if (!tree) { out->Leff = L; out->full = true; return WX_OK; }
    out->Leff = wx_tree_depth2d(tree, ntree);
    int64_t nfull = 0, p = 1;
    for (int i = 0; i < out->Leff; ++i) { nfull += p; p *= 4; }
    bool full = true;
    for (int64_t i = 1; i <= nfull && full; ++i) full = tree[i - 1] != 0;
    out->full = full;
    if (!full) {
        out->dstatus = (const uint8_t *)scr.upload(tree, (size_t)ntree);
        if (!out->dstatus) return WX_EHIP;
        out->nstatus = ntree;
        out->htree = tree;
    }
    return WX_OK;
}

template <typename T>
static int api_wpd2d(const T *x, T *y, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(m >= 1 && n >= 1 && batch >= 0, WX_EARG, "wpd: bad dimensions");
    WX_REQUIRE(0 <= L && L <= wx_maxtransformlevels(m < n ? m : n), WX_EASSERT,
               "wpd!: @assert 0 <= L <= maxtransformlevels(x) (DWT.jl:169)");
    WX_REQUIRE(m < ((int64_t)1 << 30) && n < ((int64_t)1 << 30), WX_EUNSUPPORTED, "image side >= 2^30");
    if ((rc = wx_need_device3())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dx = (const T *)io.in(x, sizeof(T) * m * n * batch);
    T *dy = (T *)io.out(y, sizeof(T) * m * n * (L + 1) * batch);
    if (batch && (!dx || !dy)) return io.finish(WX_EHIP);
    // 64 x 64 images: the image read once, every slice written once (wx_lattice_2d64w.hip)
    if (m == 64 && n == 64 && L >= 1 && batch > 0 && !wx_force_generic() && !wx_getenv("WX_NO_2D64W")) {
        const int r = wx_lattice_2d64_wpd(dx, dy, L, batch, filt, st);
        if (r) return io.finish(r < 0 ? r : WX_OK);
    }
    T *tmp = nullptr;
    if (batch && L > 0) { tmp = (T *)scr.alloc(sizeof(T) * m * n * batch); if (!tmp) return io.finish(WX_EHIP); }
    rc = wx_dev_wpd2d<T>(dx, dy, m, n, L, batch, filt, tmp, st);
    return io.finish(rc);
}

template <typename T, bool INVERSE>
static int api_wpt2d(const T *x, T *y, int64_t m, int64_t n, int L, const uint8_t *tree, int64_t ntree,
                     int64_t batch, const double *qmf, int F, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(m >= 1 && n >= 1 && batch >= 0, WX_EARG, "wpt: bad dimensions");
    WX_REQUIRE(m < ((int64_t)1 << 30) && n < ((int64_t)1 << 30), WX_EUNSUPPORTED, "image side >= 2^30");
    if ((rc = wx_check_tree2d(m, n, L, tree, ntree))) return rc;
    if ((rc = wx_need_device3())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxTree2d tr;
    // forward pyramid (dwt / dwtall of images): the levels from the 8 x 8 approximation down run in the registers of a lane
    // (wx_dwttail.hip); the quad tree's approximation chain is 1, 2, 6, 22, ... (first child of i = 4 i - 2)
    int tail = 0;
    std::vector<uint8_t> ttree;
    // 64 x 64 images along a quad tree whose root is split: the whole tree in the registers of one wavefront (wx_lattice_2d64t.h) --
    // pyramids included (with the policy below)
    const bool q64 = tree && !wx_force_generic() && m == 64 && n == 64 && batch > 0 && ntree >= 1 && tree[0] && F >= 2 && F <= 16 && !(F & 1) &&
                     x != y && !wx_getenv("WX_NO_2D64T");
    if (tree && !wx_force_generic()) {
        const int Ld = wx_tree_depth2d(tree, ntree);
        std::vector<uint8_t> chain((size_t)ntree, 0);
        int64_t node = 1;
        for (int d = 0; d < Ld && node <= ntree; ++d, node = 4 * node - 2) chain[(size_t)node - 1] = 1;
        bool pyramid = Ld >= 1;
        for (int64_t i = 0; i < ntree && pyramid; ++i) pyramid = (tree[i] != 0) == (chain[(size_t)i] != 0);
        // small images: the whole pyramid of an image in LDS, one read and one write of the batch (wx_pyr2d.hip)
        // (64 x 64: the one-wavefront kernel pays four exchanges per level -- forward pyramids keep their lane-local tail, so it sees at most
        // three levels; inverse pyramids deeper than four levels are faster here: 0.76 against 0.84 ms per GiB of Float64 images at depth 6)
        if (pyramid && wx_pyr2d_small_ok<T>(m, n, Ld, F) && !(q64 && (!INVERSE || Ld <= 4 || F > 8))) {    // (10+ taps: the LDS kernel takes 3.3 ms per GiB)
            WxIO io(st);
            const T *dx = (const T *)io.in(x, sizeof(T) * m * n * batch);
            T *dy = (T *)io.out(y, sizeof(T) * m * n * batch);
            if (batch && (!dx || !dy)) return io.finish(WX_EHIP);
            return io.finish(wx_dev_pyr2d_small<T>(INVERSE, dx, dy, m, n, Ld, batch, filt, st));
        }
        if (!INVERSE && pyramid && (tail = wx_dwt2d_tail_levels(m, n, Ld, F, sizeof(T)))) {
            ttree.assign(tree, tree + ntree);
            node = 1;
            for (int d = 0; d < Ld && node <= ntree; ++d, node = 4 * node - 2) if (d >= Ld - tail) ttree[(size_t)node - 1] = 0;
            tree = ttree.data();
        }
    }
    if ((rc = wx_resolve_tree2d(L, tree, ntree, scr, &tr))) return rc;
    WxIO io(st);
    const T *dx = (const T *)io.in(x, sizeof(T) * m * n * batch);
    T *dy = (T *)io.out(y, sizeof(T) * m * n * batch);
    if (batch && (!dx || !dy)) return io.finish(WX_EHIP);
    if (q64 && !tr.full && tr.dstatus && tr.Leff >= 1) {
        const int r = wx_lattice_2d64t(INVERSE, dx, dy, tr.Leff, batch, m * n, filt, tr.dstatus, tr.nstatus, st);
        if (r < 0) return io.finish(r);
        if (r == 1) {
            rc = WX_OK;
            if (tail) rc = wx_dwt2d_tail<T>(dy, m, tail, batch, filt, st);
            return io.finish(rc);
        }
    }
    // 64 x 64 images: the whole quad tree in the registers of one wavefront, one pass (wx_lattice_2d64.h)
    if (tr.full && tr.Leff > 0 && !tail && m == 64 && n == 64 && batch && !wx_force_generic() && !wx_getenv("WX_NO_2D64")) {
        const int r = wx_lattice_2d64(INVERSE, dx, dy, tr.Leff, batch, m * n, filt, st);
        if (r) return io.finish(r < 0 ? r : WX_OK);
    }
    T *tmp = nullptr, *pong = nullptr;
    if (batch && tr.Leff > 0) {
        // (the ring of the fused 2-D launch is at most the batch, plus one image for an odd batch of 256 x 256 images)
        int64_t te = m * n * batch;
        if (wx_lattice2d_ok(m, n, tr.Leff, filt, sizeof(T)) && wx_lattice2d_ring_elems(m, batch) > te) te = wx_lattice2d_ring_elems(m, batch);
        tmp = (T *)scr.alloc(sizeof(T) * te);
        if (!tmp) return io.finish(WX_EHIP);
    }
    if (tr.full && tr.Leff > 0 && !wx_force_generic() && (wx_wpt2d_fast_ok<T>(m, n, F) || wx_lattice2d_ok(m, n, tr.Leff, filt, sizeof(T)))) {
        rc = wx_dev_wpt2d_fast<T>(dx, dy, m, n, tr.Leff, batch, filt, tmp, INVERSE, m * n, st);
        if (rc == WX_OK && tail) rc = wx_dwt2d_tail<T>(dy, m, tail, batch, filt, st);
        return io.finish(rc);
    }
    // a full tree the fast path does not take (rows longer than its LDS strips: Float32 1024 x 1024, Float64 from 512 x 512): as a
    // tree that happens to be full through the tree-driven tile kernels -- one launch per level instead of two naive 1-D passes
    // (1024 x 1024 Float32, L = 3: 7.9 / 5.4 ms; 512 x 512 Float64: 9.8 / 7.6 ms per GiB before)
    std::vector<uint8_t> fulltree;
    if (tr.full && tr.Leff > 0 && tr.Leff <= 10 && !tr.dstatus && !wx_force_generic()) {
        fulltree.assign((size_t)((((int64_t)1 << (2 * tr.Leff)) - 1) / 3), (uint8_t)1);
        tr.dstatus = (const uint8_t *)scr.upload(fulltree.data(), fulltree.size());
        if (!tr.dstatus) return io.finish(WX_EHIP);
        tr.nstatus = (int64_t)fulltree.size();
        tr.htree = fulltree.data();
    }
    if (batch && tr.Leff > 1) { pong = (T *)scr.alloc(sizeof(T) * m * n * batch); if (!pong) return io.finish(WX_EHIP); }
    rc = wx_dev_wpt2d<T>(dx, dy, m, n, tr.Leff, batch, filt, tr.dstatus, tr.nstatus, tmp, pong, INVERSE, m * n, st, tr.htree);
    if (rc == WX_OK && tail) rc = wx_dwt2d_tail<T>(dy, m, tail, batch, filt, st);
    return io.finish(rc);
}

template <typename T>
static int api_iwpd2d(const T *xw, T *xh, int64_t m, int64_t n, int k, int L, const uint8_t *tree, int64_t ntree,
                      int64_t batch, const double *qmf, int F, void *stream)
{
    WxFilt filt;
    int rc = wx_pack_filter(qmf, F, &filt);
    if (rc) return rc;
    WX_REQUIRE(m >= 1 && n >= 1 && batch >= 0 && k >= 1, WX_EARG, "iwpd: bad dimensions");
    WX_REQUIRE(m < ((int64_t)1 << 30) && n < ((int64_t)1 << 30), WX_EUNSUPPORTED, "image side >= 2^30");
    if ((rc = wx_check_tree2d(m, n, L, tree, ntree))) return rc;
    const int Leff = tree ? wx_tree_depth2d(tree, ntree) : L;
    WX_REQUIRE(Leff < k, WX_EBOUNDS, "iwpd!: the tree needs slice d+2 beyond size(xw,3) (DWT.jl:383-386)");
    if ((rc = wx_need_device3())) return rc;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxTree2d tr;
    if ((rc = wx_resolve_tree2d(L, tree, ntree, scr, &tr))) return rc;
    WxIO io(st);
    const int64_t mn = m * n;
    const T *dxw = (const T *)io.in(xw, sizeof(T) * mn * k * batch);
    T *dxh = (T *)io.out(xh, sizeof(T) * mn * batch);
    if (batch && (!dxw || !dxh)) return io.finish(WX_EHIP);
    T *tmp = nullptr, *pong = nullptr, *leaves = nullptr;
    if (tr.full && tr.Leff > 0 && m == 64 && n == 64 && batch && !wx_force_generic() && !wx_getenv("WX_NO_2D64")) {
        const int r = wx_lattice_2d64(true, dxw + (int64_t)tr.Leff * mn, dxh, tr.Leff, batch, mn * k, filt, st);
        if (r) return io.finish(r < 0 ? r : WX_OK);
    }
    if (batch && tr.Leff > 0) {
        int64_t te = mn * batch;
        if (wx_lattice2d_ok(m, n, tr.Leff, filt, sizeof(T)) && wx_lattice2d_ring_elems(m, batch) > te) te = wx_lattice2d_ring_elems(m, batch);
        tmp = (T *)scr.alloc(sizeof(T) * te);
        if (!tmp) return io.finish(WX_EHIP);
    }
    if (tr.full && tr.Leff > 0 && !wx_force_generic() && (wx_wpt2d_fast_ok<T>(m, n, F) || wx_lattice2d_ok(m, n, tr.Leff, filt, sizeof(T)))) {
        rc = wx_dev_wpt2d_fast<T>(dxw + (int64_t)tr.Leff * mn, dxh, m, n, tr.Leff, batch, filt, tmp, true, mn * k, st);
        return io.finish(rc);
    }
    if (batch && tr.Leff > 1) { pong = (T *)scr.alloc(sizeof(T) * mn * batch); if (!pong) return io.finish(WX_EHIP); }
    const T *src = dxw + (int64_t)tr.Leff * mn;      // full tree: every leaf is in slice Leff
    int64_t in_img = mn * k;
    if (!tr.full) {
        std::vector<int> col((size_t)1 << (2 * tr.Leff), 0);
        wx_colmap2d_rec(tree, ntree, 1, 0, 0, 0, tr.Leff, col);
        const int *dcol = (const int *)scr.upload(col.data(), col.size() * sizeof(int));
        if (!dcol) return io.finish(WX_EHIP);
        if (batch) { leaves = (T *)scr.alloc(sizeof(T) * mn * batch); if (!leaves) return io.finish(WX_EHIP); }
        rc = wx_dev_gather_leaves2d<T>(dxw, leaves, m, n, k, batch, dcol, 1 << tr.Leff, st);
        if (rc) return io.finish(rc);
        src = leaves;
        in_img = mn;
    }
    rc = wx_dev_wpt2d<T>(src, dxh, m, n, tr.Leff, batch, filt, tr.dstatus, tr.nstatus, tmp, pong, true, in_img, st, tr.htree);
    return io.finish(rc);
}

extern "C" {

int wx_wpd2d_f64(const double *x, double *y, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream)
{ return api_wpd2d<double>(x, y, m, n, L, batch, qmf, F, stream); }
int wx_wpd2d_f32(const float *x, float *y, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream)
{ return api_wpd2d<float>(x, y, m, n, L, batch, qmf, F, stream); }
int wx_wpt2d_f64(const double *x, double *y, int64_t m, int64_t n, int L, const uint8_t *tree, int64_t ntree, int64_t batch,
                 const double *qmf, int F, void *stream)
{ return api_wpt2d<double, false>(x, y, m, n, L, tree, ntree, batch, qmf, F, stream); }
int wx_wpt2d_f32(const float *x, float *y, int64_t m, int64_t n, int L, const uint8_t *tree, int64_t ntree, int64_t batch,
                 const double *qmf, int F, void *stream)
{ return api_wpt2d<float, false>(x, y, m, n, L, tree, ntree, batch, qmf, F, stream); }
int wx_iwpt2d_f64(const double *xw, double *xhat, int64_t m, int64_t n, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream)
{ return api_wpt2d<double, true>(xw, xhat, m, n, L, tree, ntree, batch, qmf, F, stream); }
int wx_iwpt2d_f32(const float *xw, float *xhat, int64_t m, int64_t n, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream)
{ return api_wpt2d<float, true>(xw, xhat, m, n, L, tree, ntree, batch, qmf, F, stream); }
int wx_iwpd2d_f64(const double *xw, double *xhat, int64_t m, int64_t n, int k, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream)
{ return api_iwpd2d<double>(xw, xhat, m, n, k, L, tree, ntree, batch, qmf, F, stream); }
int wx_iwpd2d_f32(const float *xw, float *xhat, int64_t m, int64_t n, int k, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream)
{ return api_iwpd2d<float>(xw, xhat, m, n, k, L, tree, ntree, batch, qmf, F, stream); }

}  // extern "C"
