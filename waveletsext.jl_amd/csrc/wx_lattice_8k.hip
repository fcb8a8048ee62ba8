// wx_lattice_8k.hip -- launchers of the one-pass 8192-sample kernels (wx_lattice_8k.h), forward
#include "wx_lattice_8k.h"
bool wx_lattice_factor(const WxFilt &filt, int L, bool inverse, WxLat *out);
// 0 = not applicable, 1 = launched, < 0 = error.  x: (8192, batch) dense, y likewise, L = 7 .. 13 levels
int wx_lattice_wpt8k_f64(const double *x, double *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    static const bool off = wx_getenv("WX_LATTICE_8K") && atoi(wx_getenv("WX_LATTICE_8K")) == 0;
    if (off || L < 7 || L > 13 || filt.F < 2 || filt.F > 20 || batch <= 0 || batch > 0x7fffffff || x == y) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    WxLat cf;
    if (!wx_lattice_factor(filt, L - 1, false, &cf)) return 0;
#define WX_GO8(NSS)                                                                                                      \
    case NSS: hipLaunchKernelGGL((k_lat_wpt8k_f64<NSS, 2>), dim3((unsigned)batch), dim3(128), 0, st, x, y, L - 1, batch, cf, filt); break;
    switch (wx_lat_stages(filt.F)) {
        WX_GO8(1) WX_GO8(2) WX_GO8(4) WX_GO8(6) WX_GO8(8) WX_GO8(10)
    default: return 0;
    }
#undef WX_GO8
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpt launch (8192 samples)", __FILE__, __LINE__);
    return 1;
}
