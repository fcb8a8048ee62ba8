#!/bin/bash
# final evidence of round 6 (GPU box, repo root): parity log, floor tables, profiles + bench lines of every workload, default line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r06fin
O=gpurun_out/r06fin
timeout 2400 python -m pytest tests -q -m gpu > $O/r06_pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/r06_pytest_gpu.log | tail -1
timeout 1800 python -m pytest tests/test_gpu_perf_floor.py -q -m perf > $O/r06_pytest_perf.log 2>&1; echo "perf rc $?"; tail -2 $O/r06_pytest_perf.log
cp gpurun_out/r06_floor.txt gpurun_out/r06_floor2d.txt $O/ 2>/dev/null
{ echo "# tools/floor_scan_wpd.py db4, one MI355X: wpdall / iwpdall of the full tree, tables of about 1 GiB; fractions of 8 TB/s on the algorithmic bytes"; timeout 900 python tools/floor_scan_wpd.py db4 2>&1 | grep "^f"; } > $O/r06_floor_wpd.txt; tail -2 $O/r06_floor_wpd.txt
{ echo "# tools/floor_scan.py db8 (config 2's filter), one MI355X, 1 GiB batches"; timeout 1500 python tools/floor_scan.py db8 2>&1 | grep "^f"; } > $O/r06_floor_db8.txt; tail -2 $O/r06_floor_db8.txt
{ echo "# tools/floor_scan.py db3 / db5 / db7 (filters that run on the next even stage count since round 6), one MI355X, 1 GiB batches, lengths 256 / 4096"; for w in db3 db4 db5 db6 db7 db8; do echo "# $w"; timeout 600 python tools/floor_scan.py $w f64 256 4096 2>&1 | grep "^f64"; done; } > $O/r06_floor_oddstages.txt; tail -3 $O/r06_floor_oddstages.txt
{ echo "# tools/floor_scan_wpd.py db8"; timeout 900 python tools/floor_scan_wpd.py db8 2>&1 | grep "^f"; } > $O/r06_floor_wpd_db8.txt
{ echo "# tools/floor_scan_swt.py, one MI355X"; timeout 1200 python tools/floor_scan_swt.py 2>&1 | grep "^f"; } > $O/r06_floor_swt.txt; tail -2 $O/r06_floor_swt.txt
{ echo "# tools/floor_scan_misc.py: the callers' side at several lengths, Float64, db4, 1 GiB of signals / 1 GiB tables, one MI355X (round-6 build)"; timeout 1200 python tools/floor_scan_misc.py 64 128 256 512 1024 2048 4096 16384 2>&1 | grep "^f"; } > $O/r06_floor_misc.txt; tail -3 $O/r06_floor_misc.txt
WX_EVIDENCE_NOTEST=1 WX_EVIDENCE_WORKLOADS="cfg2 denoise wpt_db8 target target_f32 target_n2048 target_n1024 target_haar tree_random tree_pyramid cfg3 cfg3_sdwt swpt_db4 cfg4 cfg4_256 cfg4_1024 cfg5 bb ldb siwt dwt_long" bash tools/refresh_evidence.sh r06d r06 2>&1 | tail -5
timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/r06_bench_default.err; echo "bench rc $?"; tail -c 300 $O/r06_bench_default.json
