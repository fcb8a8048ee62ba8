#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05j
mkdir -p $O
python -m pytest tests/test_gpu_lattice_8k.py tests/test_gpu_lattice.py tests/test_gpu_dwt_long.py tests/test_gpu_toptile.py tests/test_gpu_dwt1d.py tests/test_gpu_lattice_pairs.py -m gpu -q > $O/pytest_8k.log 2>&1; echo "pytest rc $?"; tail -12 $O/pytest_8k.log
python tools/floor_scan.py db4 f64 4096 8192 16384 > $O/floor_f64_db4.txt 2>&1; cat $O/floor_f64_db4.txt
python tools/floor_scan.py db8 f64 8192 > $O/floor_f64_db8.txt 2>&1; cat $O/floor_f64_db8.txt
python tools/floor_scan.py haar f64 8192 > $O/floor_f64_haar.txt 2>&1; cat $O/floor_f64_haar.txt
