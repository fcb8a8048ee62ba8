// row pass of the 2-D full-tree transforms on the lattice kernels (wx_lattice_rows.h): images of 1024 columns, forward
#define WX_ROWS_SH 2
#define WX_ROWS_INV false
#define WX_ROWS_FN(T) wx_lattice_rows_2f_##T
#include "wx_lattice_rows.h"
