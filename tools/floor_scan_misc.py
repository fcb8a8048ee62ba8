"""The callers' side of the path at several lengths: denoiseall (dwt -> per-signal MAD -> threshold -> idwt, Denoising.jl:651-712; from the signals and from their coefficients),
bestbasistreeall(wpdall(x), BB()) (BestBasis.jl:253-262) and getbasiscoefall along one tree (Utils.jl:199-225), Float64, db4, batches of about 1 GiB of
signal (and packet tables of 1 GiB).  Times in ms and effective GB/s on signal-read-once + written-once bytes (denoise) or on the table's bytes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from floor_scan import timed, HBM_PEAK  # noqa: E402


def scan(lengths=None):
    import numpy as np
    import torch
    import waveletsext_jl_amd as wx
    wt = wx.wavelet(wx.WT.db4)
    for n in lengths or [64, 256, 1024, 4096, 16384]:
        B = (1 << 30) // (n * 8)
        x = wx.jl_empty((n, B), torch.float64, "cuda")
        x.normal_()
        L = wx.maxtransformlevels(n)
        t = timed(torch, lambda: wx.denoiseall(x, "sig", wt))
        print("f64 n %6d denoiseall(sig, dwt)      %7.3f ms (%4.1f %% of peak on 2 x signal bytes)" % (n, t, 100 * 2.0 * n * B * 8 / (t * 1e-3) / HBM_PEAK), flush=True)
        xw = wx.dwtall(x, wt)
        t = timed(torch, lambda: wx.denoiseall(xw, "dwt", wt))
        print("f64 n %6d denoiseall(dwt)           %7.3f ms (%4.1f %% of peak on 2 x signal bytes)" % (n, t, 100 * 2.0 * n * B * 8 / (t * 1e-3) / HBM_PEAK), flush=True)
        del x, xw
        torch.cuda.empty_cache()
        Bq = max((1 << 30) // (n * (L + 1) * 8), 1)          # tables of 1 GiB (0.25 GiB until round 6: launch-bound at every length)
        xq = wx.jl_empty((n, Bq), torch.float64, "cuda")
        xq.normal_()
        tab = wx.wpdall(xq, wt, L)
        tb = n * (L + 1) * Bq * 8
        t = timed(torch, lambda: wx.bestbasistreeall(tab, wx.BB()))
        print("f64 n %6d bestbasistreeall(BB)       %7.3f ms (%4.1f %% on the table's bytes)" % (n, t, 100 * tb / (t * 1e-3) / HBM_PEAK), flush=True)
        tree = np.asarray(wx.maketree(n, L, "dwt"), dtype=bool)
        t = timed(torch, lambda: wx.getbasiscoefall(tab, tree))
        print("f64 n %6d getbasiscoefall(pyramid)   %7.3f ms (%4.1f %% on 2 x signal bytes)" % (n, t, 100 * 2.0 * n * Bq * 8 / (t * 1e-3) / HBM_PEAK), flush=True)
        t = timed(torch, lambda: wx.bestbasistree(tab, wx.JBB()))
        print("f64 n %6d bestbasistree(JBB)         %7.3f ms (%4.1f %% on the table's bytes)" % (n, t, 100 * tb / (t * 1e-3) / HBM_PEAK), flush=True)
        labels = [i % 3 for i in range(Bq)]
        f = wx.LocalDiscriminantBasis(wt=wt, n_features=10)
        t = timed(torch, lambda: wx.fit_transform(f, xq, labels), calls=2, batches=2)
        print("f64 n %6d LDB fit_transform          %7.3f ms (%4.1f %% on the table's bytes)" % (n, t, 100 * tb / (t * 1e-3) / HBM_PEAK), flush=True)
        del xq, tab
        torch.cuda.empty_cache()


if __name__ == "__main__":
    scan([int(v) for v in sys.argv[1:]] or None)
