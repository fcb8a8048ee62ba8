"""Randomised comparison of the one-pass denoise entry points with the oracle (not part of the test-suite; GPU box, repo root):
python3 tools/dbg/fuzz_denoise.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import waveletsext_jl_amd as wx
from oracle import wx_oracle as oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
TH = {"hard": wx.HardTH, "soft": wx.SoftTH, "semisoft": wx.SemiSoftTH, "stein": wx.SteinTH}
worst = 0.0
for case in range(cases):
    n = int(2 ** rng.integers(6, 13))
    B = int(rng.integers(1, 200)) if n <= 1024 else int(rng.integers(1, 40))
    wname = str(rng.choice(["haar", "db2", "db3", "db4", "db5", "db6", "db7", "db8", "coif2", "coif4", "db10"]))
    wt = wx.wavelet(getattr(wx.WT, wname))
    Lmax = wx.maxtransformlevels(n)
    L = int(rng.integers(1, Lmax + 1))
    smooth = str(rng.choice(["regular", "undersmooth"]))
    thname = str(rng.choice(["hard", "soft", "semisoft", "hard", "soft"]))
    kind = str(rng.choice(["noisy", "ties", "sparse", "heavy", "scaled"]))
    x = rng.standard_normal((n, B))
    if kind == "ties":
        x = np.round(x * 3) / 3
    elif kind == "sparse":
        x = np.where(rng.random((n, B)) < 0.05, x, 0.0)
    elif kind == "heavy":
        x = rng.standard_cauchy((n, B))
    elif kind == "scaled":
        x = x * 10.0 ** rng.integers(-150, 150)
    x = np.asfortranarray(x + (np.sin(np.arange(n) / 7.0)[:, None] if kind == "noisy" else 0.0))
    dnt = wx.VisuShrink(n, TH[thname]())
    route = str(rng.choice(["sig", "dwt"]))
    xin = x if route == "sig" else np.asfortranarray(wx.to_numpy(wx.dwtall(x, wt, L)))
    Y = wx.to_numpy(wx.denoiseall(xin, route, wt, L=L, dnt=dnt, smooth=smooth))
    for i in sorted({0, B // 2, B - 1}):
        exp = oracle.denoise(np.asfortranarray(xin[:, i]), route, wt.qmf, L=L, th=thname, t=dnt.t, smooth=smooth)
        den = max(np.abs(exp).max(), np.abs(xin[:, i]).max(), 1e-300)
        err = np.abs(Y[:, i] - exp).max() / den
        worst = max(worst, err)
        if not err <= 1e-9:
            print("MISMATCH", dict(case=case, n=n, B=B, w=wname, L=L, smooth=smooth, th=thname, kind=kind, route=route, i=i, err=err), flush=True)
print("cases %d worst relative error %.2e" % (cases, worst))
