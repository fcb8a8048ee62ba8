// wx_lattice_sg32_5.hip -- Float32 signals of 128 samples on the interleaved lattice kernels (wx_lattice_sg32.h)
#define WX_G32_SH 5
#define WX_G32_FN wx_lattice_g32_5
#include "wx_lattice_sg32.h"
