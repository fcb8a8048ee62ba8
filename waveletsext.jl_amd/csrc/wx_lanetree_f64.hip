// wx_lanetree_f64.hip -- the lane-per-signal tree kernels (wx_lanetree.h) for double signals.
#include "wx_lanetree.h"

int wx_lane_tree_f64(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, const WxLaneTree &tree, const WxFilt &filt, hipStream_t st)
{
    switch (filt.F) {
#define WX_LT(FF) case FF: return wx_lane_dispatch<double, FF>(inverse, x, y, n, L, batch, tree, filt, st);
        WX_LT(2) WX_LT(4) WX_LT(6) WX_LT(8) WX_LT(10) WX_LT(12) WX_LT(14) WX_LT(16) WX_LT(18) WX_LT(20)
#undef WX_LT
    }
    return wx_set_error(WX_EUNSUPPORTED, "lane-per-signal tree kernel: filter length");
}
