#!/bin/bash
# short-signal tree kernels (wx_lattice_tree_s.h): parity and the floor cells of 64 .. 512 samples, with and without them
O=gpurun_out/r05t; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_lattice_trees_short.py tests/test_gpu_lattice_tree.py tests/test_gpu_lattice_tree32.py tests/test_gpu_smalltree.py tests/test_gpu_lattice_8k.py tests/test_gpu_dwt1d.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -8 $O/pytest.log
timeout 900 python tools/floor_scan.py db4 both 64 128 256 512 2>&1 | grep -v "full tree" | tee $O/floor_new.txt
echo "--- WX_LATTICE_TREES=0"
WX_KNOBS=1 WX_LATTICE_TREES=0 timeout 900 python tools/floor_scan.py db4 both 64 128 256 512 2>&1 | grep -v "full tree" | tee $O/floor_old.txt
