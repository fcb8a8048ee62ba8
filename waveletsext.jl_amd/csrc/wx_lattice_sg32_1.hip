// wx_lattice_sg32_1.hip -- Float32 signals of 2048 samples on the interleaved lattice kernels (wx_lattice_sg32.h)
#define WX_G32_SH 1
#define WX_G32_NSMAX 8
#define WX_G32_FN wx_lattice_g32_1
#include "wx_lattice_sg32.h"
