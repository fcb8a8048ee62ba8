// row pass of the 2-D full-tree transforms on the lattice kernels (wx_lattice_rows.h): images of 256 columns, inverse
#define WX_ROWS_SH 4
#define WX_ROWS_INV true
#define WX_ROWS_FN(T) wx_lattice_rows_4i_##T
#include "wx_lattice_rows.h"
