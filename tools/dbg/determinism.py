import sys, os
sys.path.insert(0, os.getcwd())
import torch
import waveletsext_jl_amd as wx
ok = True
for wname in ("haar", "db4", "db8"):
    wt = wx.wavelet(getattr(wx.WT, wname))
    for n, B, L in ((64, 200001, 6), (256, 100003, 8), (1024, 70001, 10), (2048, 30001, 11), (4096, 20000, 12), (8192, 5001, 12)):
        if n < 1024 and wname == "db8" and False:
            continue
        x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_(generator=torch.Generator(device="cuda").manual_seed(n))
        y0 = wx.wptall(x, wt, L)
        for _ in range(3):
            y = wx.wptall(x, wt, L)
            if not torch.equal(y, y0): ok = False; print("NONDET fwd", wname, n)
        xr = wx.iwptall(y0, wt, L)
        err = float((xr - x).abs().max())
        e0, e1 = (x * x).sum(dim=0), (y0 * y0).sum(dim=0)
        en = float(((e1 - e0).abs() / e0).max())
        if err > 1e-11 or en > 1e-11: ok = False; print("BAD", wname, n, err, en)
        del x, y0, y, xr
    x = wx.jl_empty((1024, 3001), torch.float64, "cuda"); x.normal_()
    s0 = wx.swptall(x, wt, 10)
    for _ in range(2):
        if not torch.equal(wx.swptall(x, wt, 10), s0): ok = False; print("NONDET swpt", wname)
    back = wx.iswptall(s0, wt)
    if float((back - x).abs().max()) > 1e-10: ok = False; print("BAD iswpt", wname, float((back - x).abs().max()))
    del x, s0, back
print("all good" if ok else "PROBLEMS")
