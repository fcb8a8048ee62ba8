// wx_lattice_8kti.hip -- launcher of the one-pass tree-driven kernels for 8192-sample Float64 signals (wx_lattice_8kt.h), iwpt
#include "wx_lattice_8kt.h"
bool wx_lattice_factor(const WxFilt &filt, int L, bool inverse, WxLat *out);
// x, y: (8192, batch) dense.  The root is split; child c (0 = approximation, 1 = detail) is a leaf when dstatus_c is NULL / depth_c is 0,
// otherwise dstatus_c holds the 4095 status bytes of its subtree (heap order from the child) of depth depth_c (device memory).
// 0 = not applicable, 1 = launched, < 0 = error
int wx_lattice_tree8k_inv_f64(const double *x, double *y, int64_t batch, const WxFilt &filt, const uint8_t *dstatus0, int depth0,
          const uint8_t *dstatus1, int depth1, hipStream_t st)
{
    static const bool off = wx_getenv("WX_LATTICE_8K") && atoi(wx_getenv("WX_LATTICE_8K")) == 0;
    if (off || filt.F < 2 || filt.F > 20 || batch <= 0 || batch > 0x7fffffff || x == y || depth0 > 12 || depth1 > 12) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, 1, true, &cw.c)) return 0;
    {
        WxLat one;
        if (!wx_lattice_factor(filt, 1, false, &one)) return 0;
        const long double g = one.g0;                           // product of the cosines of one level
        long double acc = 1;
        for (int l = 0; l <= 12; ++l) { cw.gl[l] = (double)acc; acc *= true ? 1 / g : g; }
    }
    cw.tail_bsig = 0;
    WxScratch scr(st);
    const WxLatTreeSc *t0 = nullptr, *t1 = nullptr;
    int rc = wx_lat8k_child_tab(dstatus0, depth0, scr, st, &t0);
    if (rc == WX_OK) rc = wx_lat8k_child_tab(dstatus1, depth1, scr, st, &t1);
    if (rc != WX_OK) return rc;
#define WX_GO8T(NSS)                                                                                                      \
    case NSS: hipLaunchKernelGGL((k_lat_iwpt_treesc8k_f64<NSS, 2>), dim3((unsigned)batch), dim3(128), 0, st, x, y, batch, cw, t0, t1, filt); break;
    switch (wx_lat_stages(filt.F)) {
        WX_GO8T(1) WX_GO8T(2) WX_GO8T(4) WX_GO8T(6) WX_GO8T(8) WX_GO8T(10)
    default: return 0;
    }
#undef WX_GO8T
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice tree launch (8192 samples)", __FILE__, __LINE__);
    return 1;
}
