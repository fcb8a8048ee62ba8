"""Diagnostic only: phase shares of the fused forward kernel from the -DWX_STAMPS build
(tools/bin/libwx_stamps.so).  Never used for reported timings."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waveletsext_jl_amd as wx
from waveletsext_jl_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libwx_stamps.so")
L = _lib.lib()
wl = sys.argv[1] if len(sys.argv) > 1 else "target"
n, B = 4096, 65536
wt, Lv, kind = (wx.wavelet(wx.WT.db4), 10, "wpt") if wl == "target" else (wx.wavelet(wx.WT.db8), 12, "wpd")
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
buf = (ctypes.c_ulonglong * 8)()
for rep in range(2):
    y = wx.wptall(x, wt, Lv) if kind == "wpt" else wx.wpdall(x, wt, Lv)
    torch.cuda.synchronize()
    L.wx_debug_read_stamps(buf, 1)
v = list(buf)
waves = v[5]
tot = v[4]
print(wl, "waves", waves, "cycles/wave total %.3e" % (tot / waves))
for name, val in zip(("stage", "level compute", "level barrier", "final"), v[:4]):
    print("  %-14s %5.1f %%" % (name, 100.0 * val / tot))
