// one-pass denoiseall of 4096-sample Float64 signals (wx_lattice_dn.h)
#define WX_DN_SH 0
#define WX_DN_FN wx_lattice_denoise0_f64
#include "wx_lattice_dn_l.h"
