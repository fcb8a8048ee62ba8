// tree-driven lattice kernels for signals of 64 samples, inverse (wx_lattice_tree_s.h)
#define WX_LAT_TREES_SH 6
#define WX_LAT_TREES_INV true
#define WX_LAT_TREES_FN(T) wx_lattice_trees_6i_##T
#include "wx_lattice_tree_s.h"
