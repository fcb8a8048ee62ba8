// wx_lattice_sg32_2.hip -- Float32 signals of 1024 samples on the interleaved lattice kernels (wx_lattice_sg32.h)
#define WX_G32_SH 2
#define WX_G32_NSMAX 8
#define WX_G32_FN wx_lattice_g32_2
#include "wx_lattice_sg32.h"
