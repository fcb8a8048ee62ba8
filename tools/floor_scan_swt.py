"""Redundant transforms around the benchmarked shapes (SURVEY section 8 rows for sdwt / swpt / swpd and acdwt / acwpt / acwpd): signal lengths
256 ... 16384, depth 4 and 8 (and 10 for the dwt forms), Float64, db4 and Haar; output tables of about 1 GiB.  Fractions of the 8 TB/s HBM peak
on the algorithmic bytes: forward = the signal read + the table written; inverse = the table (or, for iswpt, its 2^L leaf columns) read + the
signal written.  swt/swt_one_level.jl, SWT.jl:23-222; acwt/acwt_one_level.jl, ACWT.jl."""
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
    esz = 8
    for n in lengths or [256, 1024, 4096, 16384]:
        for kind, L in (("sdwt", 4), ("sdwt", 10 if n >= 1024 else 8), ("swpd", 4), ("swpd", 8), ("swpt", 4), ("swpt", 8), ("acdwt", 4), ("acwpd", 4), ("acwpd", 8)):
            cols = {"sdwt": L + 1, "acdwt": L + 1, "swpd": (2 << L) - 1, "acwpd": (2 << L) - 1, "swpt": 1 << L}[kind]
            B = max((1 << 30) // (n * cols * esz), 1)
            x = wx.jl_empty((n, B), torch.float64, "cuda")
            x.normal_()
            fwd = {"sdwt": wx.sdwtall, "swpd": wx.swpdall, "swpt": wx.swptall, "acdwt": wx.acdwtall, "acwpd": wx.acwpdall}[kind]
            inv = {"sdwt": wx.isdwtall, "swpd": None, "swpt": wx.iswptall, "acdwt": wx.iacdwtall, "acwpd": None}[kind]
            tf = timed(torch, lambda: fwd(x, wt, L))
            y = fwd(x, wt, L)
            gf = n * B * esz * (1 + cols)
            line = "f64 n %6d %-6s L=%-2d  fwd %7.3f ms (%4.1f %%)" % (n, kind, L, tf, 100 * gf / (tf * 1e-3) / HBM_PEAK)
            if inv is not None:
                ti = timed(torch, lambda: inv(y, wt))
                err = float((inv(y, wt) - x).abs().max() / x.abs().max())
                line += "  inv %7.3f ms (%4.1f %%)  rt %.0e" % (ti, 100 * gf / (ti * 1e-3) / HBM_PEAK, err)
            print(line, flush=True)
            if out is not None:
                out.write(line + "\n")
            rows.append(line)
            del x, y
            torch.cuda.empty_cache()
    return rows


if __name__ == "__main__":
    scan(sys.argv[1] if len(sys.argv) > 1 else "db4", [int(v) for v in sys.argv[2:]] or None)
