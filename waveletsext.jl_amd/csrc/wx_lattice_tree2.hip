// tree-driven lattice kernels for Float64 signals of 1024 samples (see wx_lattice_tree.h)
#define WX_LAT_TREE_SH 2
#define WX_LAT_TREE_FN wx_lattice_tree2_f64
#include "wx_lattice_tree.h"
