#!/bin/bash
# final evidence of round 5 (second take, after the short-signal trees, the quad-tree kernel and the wpd changes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" gpurun_out/r05_pytest_gpu.log | tail -1
timeout 1800 python -m pytest tests/test_gpu_perf_floor.py -q -m perf > gpurun_out/r05_pytest_perf.log 2>&1; echo "perf rc $?"; tail -2 gpurun_out/r05_pytest_perf.log
{ echo "# tools/floor_scan_wpd.py db4, one MI355X: wpdall / iwpdall of the full tree, tables of about 1 GiB; fractions of 8 TB/s on the algorithmic bytes"; timeout 900 python tools/floor_scan_wpd.py db4 2>&1 | grep "^f"; } > gpurun_out/r05_floor_wpd.txt; tail -3 gpurun_out/r05_floor_wpd.txt
WX_EVIDENCE_NOTEST=1 WX_EVIDENCE_WORKLOADS="cfg2 wpt_db8 target target_f32 target_n2048 target_n1024 target_haar tree_random tree_pyramid cfg3 cfg3_sdwt swpt_db4 cfg4 cfg4_256 cfg4_1024 cfg5 bb ldb siwt dwt_long" bash tools/refresh_evidence.sh r05d r05 2>&1 | tail -5
timeout 900 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; echo "bench rc $?"; tail -c 300 gpurun_out/r05_bench_default.json
