// wx_lattice_sg32_3.hip -- Float32 signals of 512 samples on the interleaved lattice kernels (wx_lattice_sg32.h)
#define WX_G32_SH 3
#define WX_G32_FN wx_lattice_g32_3
#include "wx_lattice_sg32.h"
