// wx_toptile_i.hip -- inverse kernels of the multi-level top pass (wx_toptile.h), their own translation unit for the parallel build.
#include "wx_toptile.h"

template <typename T, int F, int NL>
static int launch_top_inv(const T *src, T *dst, const T *deep, int64_t n, int64_t batch, int64_t ss, int64_t ds, int64_t dps, unsigned split,
                          unsigned deepmask, const WxFilt &filt, hipStream_t st)
{
    constexpr int TS = wx_tt_ts<T, F, NL, true>();
    typedef WxTTGeo<F, NL, TS, true> G;
    if (n < TS) return wx_set_error(WX_EUNSUPPORTED, "top levels: signal shorter than a tile");
    int W2[5] = {0, 0, 0, 0, 0};
    for (int l = 0; l <= NL; ++l) W2[l] = G::W(l) / 2;
    WxTopTree P;
    if (!wx_top_tree(NL, split, deepmask, W2, &P)) return wx_set_error(WX_EARG, "top levels: the root of the pass is not decomposed");
    constexpr size_t lds = G::lds_bytes(sizeof(T));
    auto kern = k_top_tile_inv<T, F, NL, TS>;
    if (lds > 64 * 1024) {
        static std::atomic<uint64_t> raised{0};                  // per instantiation and per device (see wx_toptile.hip)
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
                      unsigned deepmask, const WxFilt &filt, hipStream_t st)
{
#define WX_TT_NL(FF)                                                                                              \
    switch (NL) {                                                                                                 \
    case 1: return launch_top_inv<T, FF, 1>(src, dst, deep, n, batch, ss, ds, dps, split, deepmask, filt, st);     \
    case 2: return launch_top_inv<T, FF, 2>(src, dst, deep, n, batch, ss, ds, dps, split, deepmask, filt, st);     \
    case 3: return launch_top_inv<T, FF, 3>(src, dst, deep, n, batch, ss, ds, dps, split, deepmask, filt, st);     \
    default: return launch_top_inv<T, FF, 4>(src, dst, deep, n, batch, ss, ds, dps, split, deepmask, filt, st);    \
    }
    switch (filt.F) {
    case 2: WX_TT_NL(2) case 4: WX_TT_NL(4) case 6: WX_TT_NL(6) case 8: WX_TT_NL(8) case 10: WX_TT_NL(10)
    case 12: WX_TT_NL(12) case 14: WX_TT_NL(14) case 16: WX_TT_NL(16) case 18: WX_TT_NL(18) case 20: WX_TT_NL(20)
    }
#undef WX_TT_NL
    return wx_set_error(WX_EUNSUPPORTED, "top levels: filter length not instantiated");
}
template int wx_top_launch_inv<double>(const double *, double *, const double *, int64_t, int, int64_t, int64_t, int64_t, int64_t, unsigned, unsigned,
                                       const WxFilt &, hipStream_t);
template int wx_top_launch_inv<float>(const float *, float *, const float *, int64_t, int, int64_t, int64_t, int64_t, int64_t, unsigned, unsigned,
                                      const WxFilt &, hipStream_t);
