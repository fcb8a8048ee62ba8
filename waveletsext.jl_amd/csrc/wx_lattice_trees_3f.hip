// tree-driven lattice kernels for signals of 512 samples, forward (wx_lattice_tree_s.h)
#define WX_LAT_TREES_SH 3
#define WX_LAT_TREES_INV false
#define WX_LAT_TREES_FN(T) wx_lattice_trees_3f_##T
#include "wx_lattice_tree_s.h"
