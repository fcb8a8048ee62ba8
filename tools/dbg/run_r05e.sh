#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05e
mkdir -p $O
tools/dbg/pkfma_probe > $O/pkfma_probe.txt 2>&1; cat $O/pkfma_probe.txt
python -m pytest tests/test_gpu_lattice_pairs.py tests/test_gpu_lattice_g32.py tests/test_gpu_lattice_tree32.py tests/test_gpu_lattice.py tests/test_gpu_dwt_long.py tests/test_gpu_toptile.py tests/test_gpu_fuzz.py tests/test_gpu_denoise.py tests/test_gpu_2d_shapes.py -m gpu -x -q > $O/pytest_f32.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest_f32.log
python tools/floor_scan.py db4 f32 256 1024 4096 16384 > $O/floor_f32_db4.txt 2>&1; cat $O/floor_f32_db4.txt
