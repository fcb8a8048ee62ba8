// tree-driven lattice kernels for Float32 signals of 2048 samples, inverse (wx_lattice_tree32.h): one translation unit per length and direction
#define WX_LAT_TREE_SH 1
#define WX_LAT_TREE_INV 1
#define WX_LAT_TREE_FN wx_lattice_tree32_1i
#include "wx_lattice_tree32.h"
