#!/bin/bash
O=gpurun_out/r05v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dwt1d.py tests/test_gpu_dwt_long.py tests/test_gpu_toptile.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python tools/floor_scan.py db4 both 8192 16384 32768 65536 2>&1 | grep "full tree L=4\|full tree L=1[3-6]" | tee $O/floor_long.txt
