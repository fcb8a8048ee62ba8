"""Config 3's inverse is bimodal from process to process (7.0 / 8.2 ms per 64-signal chunk, rocprofv3: the kernel k_haar_iswpt
itself 6.45 / 7.59 ms -- profiles/r05_cfg3_inverse.md).  Does the placement of the 32 GiB leaf table decide it?  One process,
the table allocated several times (freed in between, optionally with a spacer allocation that shifts it): time per inverse and
forward, the table's address."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx

wt = wx.wavelet(wx.WT.haar)
n, L, B = 16384, 12, 64
dev = torch.device("cuda", 0)
x = wx.jl_empty((n, B), torch.float64, dev); x.normal_()


def timeit(fn, reps=6):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


spacers = [0, 0, 1 << 20, 3 << 20, 64 << 20, 1 << 30, 0, 0]
keep = []
for i, sp in enumerate(spacers):
    if sp:
        keep.append(torch.empty(sp, dtype=torch.uint8, device=dev))
    xw = wx.swptall(x, wt, L)
    torch.cuda.synchronize()
    f = timeit(lambda: wx.swptall(x, wt, L) if False else None) if False else None
    inv = timeit(lambda: wx.iswptall(xw, wt))
    print("alloc %d spacer %10d  table at 0x%x (mod 1 GiB: 0x%08x, mod 2 MiB: 0x%06x)  inverse %.3f ms" % (
        i, sp, xw.data_ptr(), xw.data_ptr() & ((1 << 30) - 1), xw.data_ptr() & ((2 << 20) - 1), inv), flush=True)
    del xw
    torch.cuda.empty_cache()
    if i == 5:
        keep.clear()
        torch.cuda.empty_cache()
