// wx_lattice_f32.hip -- Float32 full trees on the lattice kernels.  Round 5: pairs of signals in Float32 arithmetic
// (wx_lattice_sg32.h, lat_f2v); what those launchers decline (a single signal, an odd batch in place) keeps the round-4 form:
// Float32 in memory, Float64 in the registers (k_lat_wpt_f64<NS, WPE, float>, k_lat_iwpt_f64<NS, WPE, float> of wx_lattice_dev.h: the same element offsets with
// 8 bytes per lane, conversions at the two ends).  Half the bytes of the Float64 transform at the same arithmetic: the
// kernels are bound by FP64 issue and LDS here, not by HBM.  The result is the Float64 transform rounded once to Float32
// (the reference computes in Float32 throughout: the difference is inside the 1e-5 tolerance of the Float32 path).
#include "wx_lattice_dev.h"

#define WX_G32(k) int wx_lattice_g32_##k(bool, const float *, float *, int64_t, int, int64_t, int64_t, const WxFilt &, hipStream_t);
WX_G32(0) WX_G32(1) WX_G32(2) WX_G32(3) WX_G32(5) WX_G32(6)
#undef WX_G32

// 0 = not applicable, 1 = launched, < 0 = error
int wx_lattice_f32(bool inverse, const float *x, float *y, int64_t n, int L, int64_t batch, int64_t in_stride, const WxFilt &filt,
                   hipStream_t st)
{
    static const bool off = (wx_getenv("WX_LATTICE") && atoi(wx_getenv("WX_LATTICE")) == 0) ||
                            (wx_getenv("WX_LATTICE_F32") && atoi(wx_getenv("WX_LATTICE_F32")) == 0);
    if (off) return 0;
    // shorter signals: 2^SH of them per wavefront (wx_lattice_sg32.h)
    static const bool off_g = wx_getenv("WX_LATTICE_G32") && atoi(wx_getenv("WX_LATTICE_G32")) == 0;
    if (n != 4096 && !off_g) {
        switch (n) {
        case 2048: return wx_lattice_g32_1(inverse, x, y, n, L, batch, in_stride, filt, st);
        case 1024: return wx_lattice_g32_2(inverse, x, y, n, L, batch, in_stride, filt, st);
        case 512: return wx_lattice_g32_3(inverse, x, y, n, L, batch, in_stride, filt, st);
        case 256: return 0;     // full trees of 256 Float32 samples go through the masked tree kernels (wx_api.hip, wx_dwt1d.hip); the dedicated unit is not built since round 6
        case 128: return wx_lattice_g32_5(inverse, x, y, n, L, batch, in_stride, filt, st);
        case 64: return wx_lattice_g32_6(inverse, x, y, n, L, batch, in_stride, filt, st);
        default: return 0;
        }
    }
    if (n == 4096 && !off_g) {                               // pairs of signals in Float32 arithmetic; 0 = not taken (odd batch in place, one signal)
        const int r = wx_lattice_g32_0(inverse, x, y, n, L, batch, in_stride, filt, st);
        if (r) return r;
    }
    if (n != 4096 || L < 6 || L > 12 || filt.F < 4 || batch <= 0 || batch > 0x7fffffff) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return 0;
    if (inverse && (in_stride & 3)) return 0;
    WxLat cf;
    if (!wx_lattice_factor(filt, L, inverse, &cf)) return 0;
#define WX_GOF(NSS)                                                                                                  \
    case NSS:                                                                                                        \
        if (inverse)                                                                                                 \
            hipLaunchKernelGGL((k_lat_iwpt_f64<NSS, 2, float>), dim3((unsigned)batch), dim3(64), 0, st, x, y, L, batch, in_stride, cf); \
        else                                                                                                         \
            hipLaunchKernelGGL((k_lat_wpt_f64<NSS, 3, float>), dim3((unsigned)batch), dim3(64), 0, st, x, y, L, batch, cf); \
        break;
    switch (wx_lat_stages(filt.F)) {
        WX_GOF(2) WX_GOF(4) WX_GOF(6) WX_GOF(8) WX_GOF(10)
    default: return 0;
    }
#undef WX_GOF
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice wpt launch (Float32 memory)", __FILE__, __LINE__);
    return 1;
}
