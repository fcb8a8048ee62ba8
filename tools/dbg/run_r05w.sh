#!/bin/bash
# final tables of the round: full GPU suite, the throughput floors (1-D and 2-D tables), the default bench line
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" gpurun_out/r05_pytest_gpu.log | tail -1
timeout 1800 python -m pytest tests/test_gpu_perf_floor.py -q -m perf > gpurun_out/r05_pytest_perf.log 2>&1; echo "perf rc $?"; tail -2 gpurun_out/r05_pytest_perf.log
timeout 900 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; echo "bench rc $?"; tail -c 300 gpurun_out/r05_bench_default.json
