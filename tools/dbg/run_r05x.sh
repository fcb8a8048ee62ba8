#!/bin/bash
O=gpurun_out/r05x; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_lattice2d64.py tests/test_gpu_dwt2d.py tests/test_gpu_pyr2d_small.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -15 $O/pytest.log | cut -c1-300
timeout 600 python tools/floor_scan2d.py db4 64 2>&1 | grep -v amdgpu | tee $O/floor2d_64.txt
python - <<PY
import sys; sys.path.insert(0, "tools")
import torch, waveletsext_jl_amd as wx
from floor_scan import timed
wt = wx.wavelet(wx.WT.db4)
for dt, esz in ((torch.float64, 8), (torch.float32, 4)):
    B = (1 << 30) // (4096 * esz)
    x = wx.jl_empty((64, 64, B), dt, "cuda"); x.normal_()
    for L in (1, 2, 3, 4, 5, 6):
        y = wx.dwtall(x, wt, L)
        print(dt, L, "fwd %.3f inv %.3f" % (timed(torch, lambda: wx.dwtall(x, wt, L)), timed(torch, lambda: wx.idwtall(y, wt, L))), flush=True)
PY
