import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np, torch
import waveletsext_jl_amd as wx, wx_oracle as O
rng = np.random.default_rng(9)
for n in (1024,):
    for wname in ("db2", "db4", "coif6"):
        wt = wx.wavelet(getattr(wx.WT, wname))
        x = np.asfortranarray(rng.standard_normal((n, 2)))
        for L in (6, 7, 9, 10):
            exp = np.stack([O.swpd(x[:, b], wt.qmf, L) for b in range(2)], axis=-1)
            e = np.abs(wx.swpdall(x, wt, L) - exp).max() / np.abs(exp).max()
            expa = np.stack([O.acwpd(x[:, b], wt.qmf, L) for b in range(2)], axis=-1)
            e2 = np.abs(wx.acwpdall(x, wt, L) - expa).max() / np.abs(expa).max()
            if e > 1e-12 or e2 > 1e-12:
                print("FAIL", n, wname, L, e, e2)
print("done")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
wt = wx.wavelet(wx.WT.db4)
x = wx.jl_empty((1024, 1024), torch.float64, "cuda"); x.normal_()
gb = 8e-9 * 1024 * 1024 * 2047
print("swpdall n 1024 L 10 B 1024 (%.1f GB): %.2f ms   acwpdall %.2f ms" % (gb, t(lambda: wx.swpdall(x, wt, 10)), t(lambda: wx.acwpdall(x, wt, 10))))
