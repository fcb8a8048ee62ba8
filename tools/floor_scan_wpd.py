"""Companion of tools/floor_scan.py for the north-star operation itself at every length: wpdall (signal -> the (n, L+1, B) packet table,
DWT.jl:164-209 via dwt/dwt_all.jl:262-281) and iwpdall of the full tree (the deepest slice -> signal, dwt_all.jl:323-342), Float64 and
Float32, db4, full depth and depth 4; batches sized so that the TABLE is about 1 GiB.  Fractions of the 8 TB/s HBM peak on the algorithmic
bytes (forward: the signal read once + L + 1 slices written; inverse: one slice read + the signal written)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from floor_scan import timed, HBM_PEAK  # noqa: E402


def scan(wname="db4", lengths=None, out=None):
    import torch
    import waveletsext_jl_amd as wx
    wt = wx.wavelet(getattr(wx.WT, wname))
    rows = []
    for dt, esz, dn in ((torch.float64, 8, "f64"), (torch.float32, 4, "f32")):
        for n in lengths or [64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]:
            Lmax = wx.maxtransformlevels(n)
            for L in sorted({4, Lmax}):
                B = max((1 << 30) // (n * (L + 1) * esz), 1)
                x = wx.jl_empty((n, B), dt, "cuda")
                x.normal_()
                tf = timed(torch, lambda: wx.wpdall(x, wt, L))
                y = wx.wpdall(x, wt, L)
                ti = timed(torch, lambda: wx.iwpdall(y, wt, L))
                err = float((wx.iwpdall(y, wt, L) - x).abs().max() / x.abs().max())
                gf, gi = n * B * esz * (L + 2), 2.0 * n * B * esz
                rows.append(dict(dtype=dn, n=n, L=L, fwd_ms=tf, inv_ms=ti, fwd_frac=gf / (tf * 1e-3) / HBM_PEAK, inv_frac=gi / (ti * 1e-3) / HBM_PEAK, roundtrip=err))
                line = "%s n %6d wpd L=%-2d  fwd %6.3f ms (%4.1f %%)  inv %6.3f ms (%4.1f %%)  rt %.0e" % (
                    dn, n, L, tf, 100 * rows[-1]["fwd_frac"], ti, 100 * rows[-1]["inv_frac"], err)
                print(line, flush=True)
                if out is not None:
                    out.write(line + "\n")
                del x, y
                torch.cuda.empty_cache()
    return rows


if __name__ == "__main__":
    scan(sys.argv[1] if len(sys.argv) > 1 else "db4", [int(v) for v in sys.argv[2:]] or None)
