// wx_lattice_sg32_6.hip -- Float32 signals of 64 samples on the interleaved lattice kernels (wx_lattice_sg32.h)
#define WX_G32_SH 6
#define WX_G32_FN wx_lattice_g32_6
#include "wx_lattice_sg32.h"
