"""GPU diagnostics for the lattice kernels (wx_lattice.hip): parity against the oracle at n = 4096 for every filter /
depth the kernels take, then timings of the target workload.  python tools/lattice_check.py [quick]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import waveletsext_jl_amd as wx
from waveletsext_jl_amd.dwt import Arg, _wpt_batched, _iwpd_batched
from oracle import wx_oracle as O


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rng = np.random.default_rng(7)
n = 4096
worst = 0.0
for name in ("db2", "db3", "db4", "db5", "db6", "db8", "coif6", "db10"):
    wt = wx.wavelet(getattr(wx.WT, name))
    q = np.asarray(wt.qmf, dtype=np.float64)
    for L in (6, 7, 8, 9, 10, 11, 12):
        B = 3
        x = rng.standard_normal((B, n))
        ref = np.stack([O.wpt(x[b], q, L) for b in range(B)])
        xd = torch.from_numpy(x).cuda()
        got = wx.wptall(xd.T, wt, L).T.cpu().numpy() if hasattr(wx, "wptall") else None
        e1 = np.abs(got - ref).max() / np.abs(ref).max()
        back = wx.iwptall(torch.from_numpy(ref).cuda().T, wt, L).T.cpu().numpy()
        e2 = np.abs(back - x).max() / np.abs(x).max()
        worst = max(worst, e1, e2)
        flag = "" if max(e1, e2) < 1e-12 else "   <<<<<< FAIL"
        print(f"{name:6s} L={L:2d} fwd {e1:.2e} inv {e2:.2e}{flag}")
print("worst", worst)
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0)
B = 65536
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
y1 = wx.jl_empty((n, B), torch.float64, "cuda")
for name, L in (("db4", 10), ("db2", 10), ("db8", 10), ("db8", 12), ("db4", 12), ("coif6", 10)):
    wt = wx.wavelet(getattr(wx.WT, name))
    f = t(lambda: _wpt_batched("wx_wpt", Arg(x), Arg(y1), 1, wt, L, None))
    i = t(lambda: _wpt_batched("wx_iwpt", Arg(y1), Arg(x), 1, wt, L, None))
    gb = 16.0 * n * B / 1e6
    print("%-5s L=%2d  wpt %.3f ms (%.2f TB/s = %.1f %%)  iwpt %.3f ms (%.2f TB/s = %.1f %%)" % (
        name, L, f, gb / f / 1e3, gb / f / 80, i, gb / i / 1e3, gb / i / 80))
