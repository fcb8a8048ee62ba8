#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05i
mkdir -p $O
cd $R
bash tools/refresh_evidence.sh r05a r05 > $O/refresh.log 2>&1; tail -30 $O/refresh.log
python -m pytest -m perf tests/test_gpu_perf_floor.py -q > $O/perf_floor.log 2>&1; tail -5 $O/perf_floor.log
