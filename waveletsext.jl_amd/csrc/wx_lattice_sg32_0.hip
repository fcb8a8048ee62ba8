// Float32 signals of 4096 samples: pairs of signals per wavefront (wx_lattice_sg32.h), every filter the lattice factors
#define WX_G32_SH 0
#define WX_G32_NSMAX 10
#define WX_G32_FN wx_lattice_g32_0
#include "wx_lattice_sg32.h"
