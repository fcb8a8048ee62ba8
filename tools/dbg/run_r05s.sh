#!/bin/bash
# final evidence of round 5: the full GPU suite, the knobs-off run is the default (WX_KNOBS unset), the throughput floors, the rocprofv3 +
# PMC passes and a bench line per workload (tools/refresh_evidence.sh), the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r05_pytest_gpu.log
timeout 1800 python -m pytest tests/test_gpu_perf_floor.py -q -m perf > gpurun_out/r05_pytest_perf.log 2>&1; echo "perf rc $?"; tail -3 gpurun_out/r05_pytest_perf.log
WX_EVIDENCE_NOTEST=1 WX_EVIDENCE_WORKLOADS="cfg2 wpt_db8 target target_f32 target_n2048 target_n1024 target_haar tree_random tree_pyramid cfg3 cfg3_sdwt swpt_db4 cfg4 cfg4_256 cfg4_1024 cfg5 bb ldb siwt dwt_long" bash tools/refresh_evidence.sh r05c r05 2>&1 | tail -30
timeout 900 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; echo "bench rc $?"; tail -c 600 gpurun_out/r05_bench_default.json
