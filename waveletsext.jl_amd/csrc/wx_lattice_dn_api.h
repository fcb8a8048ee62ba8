// launchers of the one-pass denoise kernels (wx_lattice_dn.h): 0 = not applicable, 1 = launched, < 0 = error
#pragma once
#include "wx_host.h"
int wx_lattice_denoise0_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, int th_kind, double scale, int undersmooth,
                            double *sigma, hipStream_t st);
int wx_lattice_denoise1_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, int th_kind, double scale, int undersmooth,
                            double *sigma, hipStream_t st);
int wx_lattice_denoise2_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, int th_kind, double scale, int undersmooth,
                            double *sigma, hipStream_t st);
