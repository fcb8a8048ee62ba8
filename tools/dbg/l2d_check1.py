import sys, os
os.environ["WX_LATTICE2D_DEBUG"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np
import waveletsext_jl_amd as wx
from oracle import wx_oracle as O
rng = np.random.default_rng(5)
wt = wx.wavelet(wx.WT.db2); q = np.asarray(wt.qmf)
x = np.asfortranarray(rng.standard_normal((512, 512, 1)).astype(np.float32))
got = wx.wptall(x, wt, 6)[:, :, 0]          # Z[j + 512 o] as a (512, 512) Fortran array: got[j, o]
ref = np.stack([O.wpt(x[:, j, 0].astype(np.float64), q, 6) for j in range(512)], axis=0)   # ref[j, o]
d = np.abs(got - ref)
print("err", d.max() / np.abs(ref).max())
bad = d > 1e-4
print("bad count", bad.sum(), "of", bad.size)
print("bad j (columns):", np.where(bad.any(axis=1))[0][:40])
print("bad o (positions):", np.where(bad.any(axis=0))[0][:64])
print("ratio got/ref at a few good/bad spots:", (got[:3, :8] / ref[:3, :8]))
