// one-pass denoiseall of 512-sample Float64 signals (wx_lattice_dn.h)
#define WX_DN_SH 3
#define WX_DN_FN wx_lattice_denoise3_f64
#include "wx_lattice_dn_l.h"
