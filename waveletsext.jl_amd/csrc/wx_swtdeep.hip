// wx_swtdeep.hip -- launchers of the lane-local deep levels of swpt / acwpt (device code and the design notes: wx_swtdeep.h)
#include "wx_swtdeep.h"

// number of levels the lane-local kernel takes off the end of a depth-L swpt / acwpt of n-sample Float64 signals (0 = none)
int wx_swpt_deep_levels(int64_t n, int L, int F, bool ac, size_t esz)
{
    static const bool off = wx_getenv("WX_SWPT_DEEP") && atoi(wx_getenv("WX_SWPT_DEEP")) == 0;
    if (off || esz != 8 || n < 1024 || (n & (n - 1))) return 0;
    switch (F) { case 2: case 4: case 6: case 8: case 10: case 12: case 16: case 18: case 20: break; default: return 0; }
    int log2n = 0;
    while (((int64_t)1 << (log2n + 1)) <= n) ++log2n;
    const int D0 = log2n - 4;
    const int LP = L - D0;
    return (LP >= 1 && LP <= 4) ? LP : 0;
}

int wx_swpd_deep_fwd_impl(double *xw, int64_t n, int L, int64_t batch, const WxFilt &filt, const WxAcFilt *ac, hipStream_t st);   // wx_swtdeep_w.hip

int wx_swpt_deep_fwd(double *xw, int64_t n, int L, int64_t batch, const WxFilt &filt, const WxAcFilt *ac, bool wpd, hipStream_t st)
{
    if (wpd) return wx_swpd_deep_fwd_impl(xw, n, L, batch, filt, ac, st);
    const bool isac = ac != nullptr;
    const int F = isac ? ac->F : filt.F;
    const int LP = wx_swpt_deep_levels(n, L, F, isac, 8);
    if (!LP) return wx_set_error(WX_EHIP, "swpt deep levels: not applicable");
    int log2n = 0;
    while (((int64_t)1 << (log2n + 1)) <= n) ++log2n;
    const int D0 = log2n - 4;
    sd_kern k = nullptr;
#define WX_SD(FF) case FF: k = isac ? sd_pick<FF, true, false>(LP) : sd_pick<FF, false, false>(LP); break;
    switch (F) { WX_SD(2) WX_SD(4) WX_SD(6) WX_SD(8) WX_SD(10) WX_SD(12) WX_SD(16) WX_SD(18) WX_SD(20) }
#undef WX_SD
    WxAcFilt acz;
    if (isac) acz = *ac; else { acz.F = 0; acz.c1 = 0; }
    const int64_t gx = ((int64_t)1 << D0) * ((int64_t)1 << (D0 - 6));
    int64_t gy = batch > 65535 ? 65535 : batch;
    hipLaunchKernelGGL(k, dim3((unsigned)gx, (unsigned)gy), dim3(64), 0, st, xw, log2n, L, batch, filt, acz);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

