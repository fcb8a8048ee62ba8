// tree-driven lattice kernels, forward, Float64 signals of 4096 samples (see wx_lattice_tree.h)
#define WX_LAT_TREE_SH 0
#define WX_LAT_TREE_INV 0
#define WX_LAT_TREE_FN wx_lattice_tree0f_f64
#include "wx_lattice_tree.h"
