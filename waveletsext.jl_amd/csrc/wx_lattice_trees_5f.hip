// tree-driven lattice kernels for signals of 128 samples, forward (wx_lattice_tree_s.h)
#define WX_LAT_TREES_SH 5
#define WX_LAT_TREES_INV false
#define WX_LAT_TREES_FN(T) wx_lattice_trees_5f_##T
#include "wx_lattice_tree_s.h"
