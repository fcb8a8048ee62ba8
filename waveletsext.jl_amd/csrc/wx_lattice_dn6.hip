// one-pass denoiseall of 64-sample Float64 signals (wx_lattice_dn.h)
#define WX_DN_SH 6
#define WX_DN_FN wx_lattice_denoise6_f64
#include "wx_lattice_dn_l.h"
