// one-pass denoiseall of 2048-sample Float64 signals (wx_lattice_dn.h)
#define WX_DN_SH 1
#define WX_DN_FN wx_lattice_denoise1_f64
#include "wx_lattice_dn_l.h"
