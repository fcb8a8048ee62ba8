// wx_lattice_8ki.hip -- launchers of the one-pass 8192-sample kernels (wx_lattice_8k.h), inverse
#include "wx_lattice_8k.h"
bool wx_lattice_factor(const WxFilt &filt, int L, bool inverse, WxLat *out);
// xw: leaves of signal b at xw + b in_stride (dense or the last column of packet tables), y: (8192, batch)
int wx_lattice_iwpt8k_f64(const double *xw, double *y, int L, int64_t batch, int64_t in_stride, const WxFilt &filt, hipStream_t st)
{
    static const bool off = wx_getenv("WX_LATTICE_8K") && atoi(wx_getenv("WX_LATTICE_8K")) == 0;
    if (off || L < 7 || L > 13 || filt.F < 2 || filt.F > 20 || batch <= 0 || batch > 0x7fffffff || xw == y) return 0;
    if ((reinterpret_cast<uintptr_t>(xw) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    if (in_stride < 8192 || (in_stride & 3)) return 0;
    WxLat cf;
    if (!wx_lattice_factor(filt, L - 1, true, &cf)) return 0;
#define WX_GO8(NSS)                                                                                                      \
    case NSS: hipLaunchKernelGGL((k_lat_iwpt8k_f64<NSS, 2>), dim3((unsigned)batch), dim3(128), 0, st, xw, y, L - 1, batch, in_stride, cf, filt); break;
    switch (wx_lat_stages(filt.F)) {
        WX_GO8(1) WX_GO8(2) WX_GO8(4) WX_GO8(6) WX_GO8(8) WX_GO8(10)
    default: return 0;
    }
#undef WX_GO8
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice iwpt launch (8192 samples)", __FILE__, __LINE__);
    return 1;
}
