#!/bin/bash
# lattice row pass with the XCD-aware mapping: parity of every geometry, A/B of the shortest runs taken
O=gpurun_out/r05r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_lattice2d64.py tests/test_gpu_dwt2d.py tests/test_gpu_2d_shapes.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -5 $O/pytest.log
for sh in 2 3 4 6; do
  echo "--- WX_LATROWS_MINSH=$sh"
  WX_KNOBS=1 WX_LATROWS_MINSH=$sh timeout 600 python tools/floor_scan2d.py db4 128 256 512 1024 2>&1 | grep "full tree" | tee $O/floor2d_minsh$sh.txt
done
