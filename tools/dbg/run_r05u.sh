#!/bin/bash
# round bits for 4-byte elements (lat_round_bit, LAT_ORD32) + short pyramids whole on the tree kernels: full suite, Float32 floors, PMC of target_f32
O=gpurun_out/r05u; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 900 python tools/floor_scan.py db4 f32 64 128 256 512 1024 2048 4096 2>&1 | grep -v amdgpu.ids | tee $O/floor_f32.txt
timeout 900 python tools/floor_scan.py db4 f64 64 128 256 512 2>&1 | grep "pyramid" | tee $O/floor_f64_pyr.txt
timeout 600 python tools/floor_scan2d.py db4 64 128 2>&1 | grep "^f32" | tee $O/floor2d_f32.txt
bash tools/profile.sh r05u target_f32 pmc > /dev/null 2>&1
python3 - <<'PY'
import sys, json, os
sys.path.insert(0, 'tools')
import summarize_prof
summarize_prof.main('gpurun_out/prof_r05u/target_f32', 'gpurun_out/r05u/r05_target_f32')
PY
sed -n 1,12p gpurun_out/r05u/r05_target_f32.md; grep "k_lat" gpurun_out/r05u/r05_target_f32.md | tail -2
find gpurun_out/prof_r05u -name "*.csv" -size +2M -delete
