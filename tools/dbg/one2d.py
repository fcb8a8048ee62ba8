"""one 2-D case for a profiler: python one2d.py f32|f64 m L full|pyr|wpd [fwd|inv|both]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
dn, m, L, kind = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
which = sys.argv[5] if len(sys.argv) > 5 else "both"
dt, esz = (torch.float32, 4) if dn == "f32" else (torch.float64, 8)
wt = wx.wavelet(getattr(wx.WT, os.environ.get("WNAME", "db4")))
B = (1 << 30) // (m * m * esz)
if kind == "wpd":
    B = max(B // 4, 1)
x = wx.jl_empty((m, m, B), dt, "cuda"); x.normal_()
fwd, inv = {"full": (wx.wptall, wx.iwptall), "pyr": (wx.dwtall, wx.idwtall), "wpd": (wx.wpdall, wx.iwpdall)}[kind]
y = fwd(x, wt, L)
for _ in range(6):
    if which in ("fwd", "both"):
        y = fwd(x, wt, L)
    if which in ("inv", "both"):
        z = inv(y, wt, L)
torch.cuda.synchronize()
