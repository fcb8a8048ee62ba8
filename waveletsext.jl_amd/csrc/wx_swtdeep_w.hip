// wx_swtdeep_w.hip -- launcher of the lane-local deep levels of swpd / acwpd (heap tables; device code: wx_swtdeep.h)
#include "wx_swtdeep.h"

int wx_swpd_deep_fwd_impl(double *xw, int64_t n, int L, int64_t batch, const WxFilt &filt, const WxAcFilt *ac, hipStream_t st)
{
    const bool isac = ac != nullptr;
    const int F = isac ? ac->F : filt.F;
    const int LP = wx_swpt_deep_levels(n, L, F, isac, 8);
    if (!LP) return wx_set_error(WX_EHIP, "swpt deep levels: not applicable");
    int log2n = 0;
    while (((int64_t)1 << (log2n + 1)) <= n) ++log2n;
    const int D0 = log2n - 4;
    sd_kern k = nullptr;
#define WX_SD(FF) case FF: k = isac ? sd_pick<FF, true, true>(LP) : sd_pick<FF, false, true>(LP); break;
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

