// one-pass denoiseall of 1024-sample Float64 signals (wx_lattice_dn.h)
#define WX_DN_SH 2
#define WX_DN_FN wx_lattice_denoise2_f64
#include "wx_lattice_dn_l.h"
