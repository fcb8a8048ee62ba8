"""2-D companion of tools/floor_scan.py: square images 64 ... 1024, Float32 and Float64, 1 GiB batches: full quad trees (full depth and
L = 3), pyramids (dwtall / idwtall, full depth and L = 3) and wpdall (L = 3), as fractions of the 8 TB/s HBM peak on the algorithmic
bytes (image read once + written once; wpd: image + L + 1 slices).  Same timing as floor_scan.py (best of 3 batches of 5 calls)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from floor_scan import timed, HBM_PEAK  # noqa: E402


def scan(wname="db4", sizes=(64, 128, 256, 512, 1024), out=None):
    import torch
    import waveletsext_jl_amd as wx
    wt = wx.wavelet(getattr(wx.WT, wname))
    rows = []
    for dt, esz, dn in ((torch.float32, 4, "f32"), (torch.float64, 8, "f64")):
        for m in sizes:
            B = (1 << 30) // (m * m * esz)
            x = wx.jl_empty((m, m, B), dt, "cuda")
            x.normal_()
            Lmax = wx.maxtransformlevels(m)
            gb = 2.0 * m * m * B * esz
            cases = [("full tree L=%d" % Lmax, lambda a: wx.wptall(a, wt, Lmax), lambda a: wx.iwptall(a, wt, Lmax)),
                     ("full tree L=3", lambda a: wx.wptall(a, wt, 3), lambda a: wx.iwptall(a, wt, 3)),
                     ("pyramid L=%d" % Lmax, lambda a: wx.dwtall(a, wt, Lmax), lambda a: wx.idwtall(a, wt, Lmax)),
                     ("pyramid L=3", lambda a: wx.dwtall(a, wt, 3), lambda a: wx.idwtall(a, wt, 3))]
            for name, fwd, inv in cases:
                tf = timed(torch, lambda: fwd(x))
                y = fwd(x)
                ti = timed(torch, lambda: inv(y))
                err = float((inv(y) - x).abs().max() / x.abs().max())
                rows.append(dict(dtype=dn, m=m, case=name, fwd_ms=tf, inv_ms=ti, fwd_frac=gb / (tf * 1e-3) / HBM_PEAK,
                                 inv_frac=gb / (ti * 1e-3) / HBM_PEAK, roundtrip=err))
                line = "%s %4dx%-4d %-16s fwd %6.3f ms (%4.1f %%)  inv %6.3f ms (%4.1f %%)  rt %.0e" % (
                    dn, m, m, name, tf, 100 * rows[-1]["fwd_frac"], ti, 100 * rows[-1]["inv_frac"], err)
                print(line, flush=True)
                if out is not None:
                    out.write(line + "\n")
                del y
            # wpd, L = 3, a quarter of the batch (the table is (L + 1) images)
            xq = x[..., : max(B // 4, 1)]
            gbw = m * m * xq.shape[-1] * esz * (1 + 4)
            tw = timed(torch, lambda: wx.wpdall(xq, wt, 3))
            yw = wx.wpdall(xq, wt, 3)
            tiw = timed(torch, lambda: wx.iwpdall(yw, wt, 3))
            line = "%s %4dx%-4d %-16s fwd %6.3f ms (%4.1f %%)  inv %6.3f ms (%4.1f %% of 2 images)" % (
                dn, m, m, "wpdall L=3", tw, 100 * gbw / (tw * 1e-3) / HBM_PEAK, tiw, 100 * 2.0 * m * m * xq.shape[-1] * esz / (tiw * 1e-3) / HBM_PEAK)
            print(line, flush=True)
            if out is not None:
                out.write(line + "\n")
            del x, xq, yw
            torch.cuda.empty_cache()
    return rows


if __name__ == "__main__":
    # python tools/floor_scan2d.py [wavelet [side ...]]
    scan(sys.argv[1] if len(sys.argv) > 1 else "db4", tuple(int(v) for v in sys.argv[2:]) or (64, 128, 256, 512, 1024))
