#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05f
mkdir -p $O
python -m pytest tests/test_gpu_lattice_pairs.py tests/test_gpu_lattice_g32.py tests/test_gpu_lattice_tree32.py tests/test_gpu_lattice.py tests/test_gpu_dwt_long.py tests/test_gpu_toptile.py tests/test_gpu_fuzz.py tests/test_gpu_denoise.py tests/test_gpu_2d_shapes.py tests/test_gpu_dwt2d.py -m gpu -x -q > $O/pytest_f32.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest_f32.log
python tools/floor_scan.py db4 f32 64 128 256 512 1024 2048 4096 8192 > $O/floor_f32_db4.txt 2>&1; cat $O/floor_f32_db4.txt
python tools/floor_scan.py db2 f32 1024 4096 > $O/floor_f32_db2.txt 2>&1; cat $O/floor_f32_db2.txt
python tools/floor_scan.py db8 f32 1024 4096 > $O/floor_f32_db8.txt 2>&1; cat $O/floor_f32_db8.txt
