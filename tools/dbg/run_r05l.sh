#!/bin/bash
# final evidence of the round on the final build: the whole GPU suite, the floors, and the bench lines / profiles of the workloads whose kernels
# were rebuilt after tools/refresh_evidence.sh r05a (the lattice header was refactored: cfg2, target, trees; the 1-D long paths)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05l
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
python -m pytest -m perf tests/test_gpu_perf_floor.py -q > $O/perf_floor.log 2>&1; tail -3 $O/perf_floor.log
WX_EVIDENCE_NOTEST=1 WX_EVIDENCE_WORKLOADS="cfg2 target wpt_db8 tree_random tree_pyramid target_n2048 target_n1024 cfg5 dwt_long" bash tools/refresh_evidence.sh r05b r05 > $O/refresh.log 2>&1; tail -12 $O/refresh.log
