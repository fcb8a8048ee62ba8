// one-pass denoiseall of 256-sample Float64 signals (wx_lattice_dn.h)
#define WX_DN_SH 4
#define WX_DN_FN wx_lattice_denoise4_f64
#include "wx_lattice_dn_l.h"
