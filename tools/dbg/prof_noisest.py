"""noisest of a batch of dwt-shaped Float64 signals (1 GiB) at one length: tools/dbg/prof_any.sh tools/dbg/prof_noisest.py <n>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
from waveletsext_jl_amd import denoising as dn

n = int(sys.argv[1])
B = (1 << 30) // (n * 8)
x = wx.jl_empty((n, B), torch.float64, "cuda")
x.normal_()
for _ in range(6):
    s = dn._noisest(dn.Arg(x), True, "dwt", None, on_device=True)
torch.cuda.synchronize()
