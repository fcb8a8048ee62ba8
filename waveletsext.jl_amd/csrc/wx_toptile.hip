// wx_toptile.hip -- tree bookkeeping and launcher of the multi-level top pass (wx_toptile.h); forward kernels are instantiated here,
// inverse kernels in wx_toptile_i.hip.
#include "wx_toptile.h"
#include <cstdlib>

bool wx_top_levels_ok(int F)
{
    static const bool off = wx_getenv("WX_TOPTILE") && atoi(wx_getenv("WX_TOPTILE")) == 0;
    return !off && F >= 2 && F <= 20 && (F & 1) == 0;
}

bool wx_top_tree(int NL, unsigned split, unsigned deep, const int *W2, WxTopTree *P, bool wpd)
{
    if (NL < 1 || NL > 4) return false;
    // which nodes exist: the root, and the children of split nodes
    unsigned ex = 1u;
    for (int i = 1; i < (1 << NL); ++i)
        if (((ex >> (i - 1)) & 1u) && ((split >> (i - 1)) & 1u)) ex |= (1u << (2 * i - 1)) | (1u << (2 * i));
    P->split = split & ex & ((1u << ((1 << NL) - 1)) - 1u);
    if (!(P->split & 1u)) return false;                    // the root is not decomposed: nothing to do in this pass
    P->deep = deep;
    for (int l = 0; l < 5; ++l) { P->ns[l] = P->nf[l] = 0; P->sp[l] = 0; P->fl[l] = 0; }
    for (int l = 1; l <= NL; ++l) {
        for (int j = 0; j < (1 << (l - 1)); ++j)
            if ((P->split >> ((1 << (l - 1)) - 1 + j)) & 1u) P->sp[l] |= (unsigned)j << (4 * P->ns[l]++);
        for (int j = 0; j < (1 << l); ++j) {
            const int h = (1 << l) - 1 + j;
            // wpd: every node of every depth leaves (to its slice) whether it is decomposed further or not
            if (((ex >> h) & 1u) && (wpd || l == NL || !((P->split >> h) & 1u))) P->fl[l] |= (unsigned long long)j << (4 * P->nf[l]++);
        }
    }
    P->lstride = 0;
    for (int l = 0; l < 7; ++l) P->pf[l] = 0;
    for (int l = 1; l <= NL; ++l) P->pf[l + 1] = P->pf[l] + P->nf[l] * W2[l];
    for (int l = NL + 2; l < 7; ++l) P->pf[l] = P->pf[NL + 1];
    return true;
}

// workgroups that are resident at once (each walks its share of the tiles, the next tile's input in registers)
int64_t wx_top_grid(int64_t ntiles, size_t lds)
{
    // per device (a process may drive several GPUs): the CU count of the CURRENT device, cached per index
    static std::atomic<int> cus_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    int cus = cus_of[dev & 63].load(std::memory_order_relaxed);
    if (!cus) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount; else (void)hipGetLastError();
        if (cus <= 0) cus = 256;
        cus_of[dev & 63].store(cus, std::memory_order_relaxed);
    }
    static const int env = wx_getenv("WX_TOPTILE_WGS") ? atoi(wx_getenv("WX_TOPTILE_WGS")) : 0;
    int per = (int)((160 * 1024) / (lds + 512));
    if (per < 1) per = 1;
    if (per > 8) per = 8;
    if (env > 0) per = env;
    const int64_t g = (int64_t)cus * per;
    return ntiles < g ? ntiles : g;
}

template <typename T, int F, int NL>
static int launch_top_fwd(const T *src, T *dst, T *deep, int64_t n, int64_t batch, int64_t ss, int64_t ds, int64_t dps, unsigned split,
                          unsigned deepmask, const WxFilt &filt, hipStream_t st, int64_t lstride = 0)
{
    constexpr int TS = wx_tt_ts<T, F, NL, false>();
    typedef WxTTGeo<F, NL, TS, false> G;
    if (n < TS) return wx_set_error(WX_EUNSUPPORTED, "top levels: signal shorter than a tile");
    int W2[5] = {0, 0, 0, 0, 0};
    for (int l = 0; l <= NL; ++l) W2[l] = G::W(l) / 2;
    WxTopTree P;
    if (!wx_top_tree(NL, split, deepmask, W2, &P, lstride != 0)) return wx_set_error(WX_EARG, "top levels: the root of the pass is not decomposed");
    P.lstride = lstride;
    constexpr size_t lds = G::lds_bytes(sizeof(T));
    auto kern = k_top_tile_fwd<T, F, NL, TS>;
    if (lds > 64 * 1024) {
        // per instantiation AND per device: the attribute belongs to the function on one device (a process-wide flag made
        // the launch on a second GPU fail with more than 64 KiB of LDS)
        static std::atomic<uint64_t> raised{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
        const uint64_t bit = (uint64_t)1 << (dev & 63);
        if (dev > 63 || !(raised.load(std::memory_order_acquire) & bit)) {
            WX_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            raised.fetch_or(bit, std::memory_order_release);
        }
    }
    const int64_t ntiles = batch * ((n >> NL) / G::TL);
    if (ntiles >= ((int64_t)1 << 31)) return wx_set_error(WX_EUNSUPPORTED, "top levels: too many tiles");
    hipLaunchKernelGGL(kern, dim3((unsigned)wx_top_grid(ntiles, lds)), dim3(WX_TT_NT), lds, st, src, dst, deep, (int)n, ss, ds, dps,
                       (unsigned)ntiles, P, filt);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template <typename T>
int wx_top_launch_inv(const T *src, T *dst, const T *deep, int64_t n, int NL, int64_t batch, int64_t ss, int64_t ds, int64_t dps, unsigned split,
                      unsigned deepmask, const WxFilt &filt, hipStream_t st);     // wx_toptile_i.hip

template <typename T>
int wx_dev_top_levels(bool inverse, const T *src, T *dst, T *deep, int64_t n, int NL, int64_t batch, int64_t sstride, int64_t dstride,
                      int64_t deepstride, unsigned split, unsigned deepmask, const WxFilt &filt, hipStream_t st)
{
    if (batch == 0) return WX_OK;
    if (!wx_top_levels_ok(filt.F)) return wx_set_error(WX_EUNSUPPORTED, "top levels: filter length not instantiated");
    if (n >= ((int64_t)1 << 30) || (n & (n - 1)) || NL < 1 || NL > 4) return wx_set_error(WX_EUNSUPPORTED, "top levels: length / level count");
    if (inverse) return wx_top_launch_inv<T>(src, dst, deep, n, NL, batch, sstride, dstride, deepstride, split, deepmask, filt, st);
#define WX_TT_NL(FF)                                                                                                                  \
    switch (NL) {                                                                                                                     \
    case 1: return launch_top_fwd<T, FF, 1>(src, dst, deep, n, batch, sstride, dstride, deepstride, split, deepmask, filt, st);        \
    case 2: return launch_top_fwd<T, FF, 2>(src, dst, deep, n, batch, sstride, dstride, deepstride, split, deepmask, filt, st);        \
    case 3: return launch_top_fwd<T, FF, 3>(src, dst, deep, n, batch, sstride, dstride, deepstride, split, deepmask, filt, st);        \
    default: return launch_top_fwd<T, FF, 4>(src, dst, deep, n, batch, sstride, dstride, deepstride, split, deepmask, filt, st);       \
    }
    switch (filt.F) {
    case 2: WX_TT_NL(2) case 4: WX_TT_NL(4) case 6: WX_TT_NL(6) case 8: WX_TT_NL(8) case 10: WX_TT_NL(10)
    case 12: WX_TT_NL(12) case 14: WX_TT_NL(14) case 16: WX_TT_NL(16) case 18: WX_TT_NL(18) case 20: WX_TT_NL(20)
    }
#undef WX_TT_NL
    return wx_set_error(WX_EUNSUPPORTED, "top levels: filter length not instantiated");
}
// wpd of the top NL levels (full tree) in one pass: slices 0 .. NL of every signal's packet table
template <typename T>
int wx_dev_top_levels_wpd(const T *src, T *dst, int64_t n, int NL, int64_t batch, int64_t sstride, int64_t dstride, int64_t lstride,
                          const WxFilt &filt, hipStream_t st)
{
    if (batch == 0) return WX_OK;
    if (!wx_top_levels_ok(filt.F)) return wx_set_error(WX_EUNSUPPORTED, "top levels: filter length not instantiated");
    if (n >= ((int64_t)1 << 30) || (n & (n - 1)) || NL < 1 || NL > 4 || lstride <= 0) return wx_set_error(WX_EUNSUPPORTED, "top levels (wpd): length / level count");
#define WX_TT_NLW(FF)                                                                                                                 \
    switch (NL) {                                                                                                                     \
    case 1: return launch_top_fwd<T, FF, 1>(src, dst, (T *)nullptr, n, batch, sstride, dstride, 0, 0xffffffffu, 0u, filt, st, lstride); \
    case 2: return launch_top_fwd<T, FF, 2>(src, dst, (T *)nullptr, n, batch, sstride, dstride, 0, 0xffffffffu, 0u, filt, st, lstride); \
    case 3: return launch_top_fwd<T, FF, 3>(src, dst, (T *)nullptr, n, batch, sstride, dstride, 0, 0xffffffffu, 0u, filt, st, lstride); \
    default: return launch_top_fwd<T, FF, 4>(src, dst, (T *)nullptr, n, batch, sstride, dstride, 0, 0xffffffffu, 0u, filt, st, lstride); \
    }
    switch (filt.F) {
    case 2: WX_TT_NLW(2) case 4: WX_TT_NLW(4) case 6: WX_TT_NLW(6) case 8: WX_TT_NLW(8) case 10: WX_TT_NLW(10)
    case 12: WX_TT_NLW(12) case 14: WX_TT_NLW(14) case 16: WX_TT_NLW(16) case 18: WX_TT_NLW(18) case 20: WX_TT_NLW(20)
    }
#undef WX_TT_NLW
    return wx_set_error(WX_EUNSUPPORTED, "top levels: filter length not instantiated");
}
template int wx_dev_top_levels_wpd<double>(const double *, double *, int64_t, int, int64_t, int64_t, int64_t, int64_t, const WxFilt &, hipStream_t);
template int wx_dev_top_levels_wpd<float>(const float *, float *, int64_t, int, int64_t, int64_t, int64_t, int64_t, const WxFilt &, hipStream_t);
template int wx_dev_top_levels<double>(bool, const double *, double *, double *, int64_t, int, int64_t, int64_t, int64_t, int64_t, unsigned,
                                       unsigned, const WxFilt &, hipStream_t);
template int wx_dev_top_levels<float>(bool, const float *, float *, float *, int64_t, int, int64_t, int64_t, int64_t, int64_t, unsigned, unsigned,
                                      const WxFilt &, hipStream_t);
