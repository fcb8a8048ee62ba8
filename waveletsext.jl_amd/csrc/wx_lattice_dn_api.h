// launchers of the one-pass denoise kernels (wx_lattice_dn.h), signals of 4096 >> k samples: 0 = not applicable, 1 = launched, < 0 = error
#pragma once
#include "wx_host.h"
#define WX_DN_DECL(k)                                                                                                                              \
    int wx_lattice_denoise##k##_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, int th_kind, double scale,    \
                                    int undersmooth, double *sigma, int coefs, hipStream_t st);
WX_DN_DECL(0) WX_DN_DECL(1) WX_DN_DECL(2) WX_DN_DECL(3) WX_DN_DECL(4) WX_DN_DECL(5) WX_DN_DECL(6)
#undef WX_DN_DECL
