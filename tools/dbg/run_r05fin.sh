#!/bin/bash
# bench lines and rocprofv3 + PMC summaries of the FINAL build of round 5 (same workloads as run_r05z.sh; the kernels they time did not change)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
WX_EVIDENCE_NOTEST=1 WX_EVIDENCE_WORKLOADS="cfg2 wpt_db8 target target_f32 target_n2048 target_n1024 target_haar tree_random tree_pyramid cfg3 cfg3_sdwt swpt_db4 cfg4 cfg4_256 cfg4_1024 cfg5 bb ldb siwt dwt_long" bash tools/refresh_evidence.sh r05e r05 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; echo "bench rc $?"; tail -c 200 gpurun_out/r05_bench_default.json
