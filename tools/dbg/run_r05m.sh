#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05m
mkdir -p $O
python -m pytest tests/test_gpu_lattice_8k.py tests/test_gpu_lattice_tree.py tests/test_gpu_dwt_long.py tests/test_gpu_lattice.py tests/test_gpu_dwt1d.py tests/test_gpu_denoise.py -m gpu -q > $O/pytest_8kt.log 2>&1; echo "pytest rc $?"; tail -12 $O/pytest_8kt.log
python tools/floor_scan.py db4 f64 8192 > $O/floor_f64_db4.txt 2>&1; cat $O/floor_f64_db4.txt
