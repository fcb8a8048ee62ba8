"""timing of the lattice kernels on the target workload (diagnostics): python tools/lattice_time.py [filters...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waveletsext_jl_amd as wx
from waveletsext_jl_amd.dwt import Arg, _wpt_batched


def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


n, B = 4096, 65536
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
y1 = wx.jl_empty((n, B), torch.float64, "cuda")
x2 = wx.jl_empty((n, B), torch.float64, "cuda")
cases = [a.split(":") for a in sys.argv[1:]] or [("db4", "10"), ("db2", "10"), ("db8", "12")]
for name, L in cases:
    L = int(L)
    wt = wx.wavelet(getattr(wx.WT, name))
    f = t(lambda: _wpt_batched("wx_wpt", Arg(x), Arg(y1), 1, wt, L, None))
    i = t(lambda: _wpt_batched("wx_iwpt", Arg(y1), Arg(x2), 1, wt, L, None))
    err = float((x2 - x).abs().max())
    gb = 16.0 * n * B / 1e6
    print("%-5s L=%2d  wpt %.3f ms (%.1f %%)  iwpt %.3f ms (%.1f %%)  roundtrip %.1e" % (name, L, f, gb / f / 80, i, gb / i / 80, err))
