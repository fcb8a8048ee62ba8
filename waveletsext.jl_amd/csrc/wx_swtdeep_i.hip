// wx_swtdeep_i.hip -- launcher of the lane-local deep levels of the average-based iswpt (device code: wx_swtdeep.h)
#include "wx_swtdeep.h"

// the deepest LP = L - (log2(n) - 4) levels of the average-based iswpt: leaves (n, src_cols) -> nodes of depth L - LP (n, dst_cols)
int wx_swpt_deep_inv(const double *src, int64_t src_cols, double *dst, int64_t dst_cols, int64_t n, int L, int LP, int64_t batch,
                     const WxFilt &filt, hipStream_t st)
{
    int log2n = 0;
    while (((int64_t)1 << (log2n + 1)) <= n) ++log2n;
    const int D0 = log2n - 4;
    if (LP < 1 || LP > 4 || L - LP != D0) return wx_set_error(WX_EHIP, "iswpt deep levels: inconsistent plan");
    sd_ikern k = nullptr;
#define WX_SDI(FF) case FF: k = sd_ipick<FF>(LP); break;
    switch (filt.F) { WX_SDI(2) WX_SDI(4) WX_SDI(6) WX_SDI(8) WX_SDI(10) WX_SDI(12) WX_SDI(16) WX_SDI(18) WX_SDI(20) default: break; }
#undef WX_SDI
    if (!k) return wx_set_error(WX_EHIP, "iswpt deep levels: no instantiation for this filter length");
    const int64_t gx = ((int64_t)1 << D0) * ((int64_t)1 << (D0 - 6));
    const int64_t gy = batch > 65535 ? 65535 : batch;
    hipLaunchKernelGGL(k, dim3((unsigned)gx, (unsigned)gy), dim3(64), 0, st, src, src_cols, dst, dst_cols, log2n, batch, filt);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
