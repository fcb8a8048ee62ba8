#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05d
mkdir -p $O
python -m pytest tests/test_gpu_lattice_g32.py tests/test_gpu_lattice.py tests/test_gpu_dwt1d.py tests/test_gpu_fuzz.py tests/test_gpu_gathertrees.py -m gpu -x -q > $O/pytest_f32.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest_f32.log
python tools/floor_scan.py db4 f32 64 256 1024 2048 4096 > $O/floor_f32_db4.txt 2>&1; cat $O/floor_f32_db4.txt
python tools/floor_scan.py db8 f32 1024 4096 > $O/floor_f32_db8.txt 2>&1; cat $O/floor_f32_db8.txt
WX_KNOBS=1 WX_HOST_TRACE=1 python tools/host_path_time.py > $O/host_trace.txt 2>&1; cat $O/host_trace.txt
WX_KNOBS=1 WX_HOST_TRACE=1 python bench.py --steps 3 --warmup 1 --no-also --cpu-seconds 1 > $O/bench_trace.json 2> $O/bench_trace.err; grep "wx host" $O/bench_trace.err | tail -12
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r05d/bench_trace.json") if l.startswith("{")][-1])
print("pcie", j.get("pcie_inclusive"))
PY
