// tree-driven lattice kernels for Float32 signals of 1024 samples, forward (wx_lattice_tree32.h): one translation unit per length and direction
#define WX_LAT_TREE_SH 2
#define WX_LAT_TREE_INV 0
#define WX_LAT_TREE_FN wx_lattice_tree32_2f
#include "wx_lattice_tree32.h"
