// wx_lattice_sg32_4.hip -- Float32 signals of 256 samples on the interleaved lattice kernels (wx_lattice_sg32.h)
#define WX_G32_SH 4
#define WX_G32_FN wx_lattice_g32_4
#include "wx_lattice_sg32.h"
