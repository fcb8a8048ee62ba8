// tree-driven lattice kernels for Float32 signals of 4096 samples, forward (wx_lattice_tree32.h): one translation unit per length and direction
#define WX_LAT_TREE_SH 0
#define WX_LAT_TREE_INV 0
#define WX_LAT_TREE_FN wx_lattice_tree32_0f
#include "wx_lattice_tree32.h"
