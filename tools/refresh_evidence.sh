#!/bin/bash
# usage (GPU box, repo root): bash tools/refresh_evidence.sh <tag> [round-prefix, default r02]
# full GPU test log, the three rocprofv3 passes per workload, then one bench line per workload (with the CPU
# baseline) reading the traffic table just measured; everything lands under gpurun_out/ and
# tools/collect_evidence.py files it under profiles/
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
mkdir -p $R/gpurun_out/bench_$TAG
cd $R
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu_$TAG.log | tail -1
WL=${WX_EVIDENCE_WORKLOADS:-cfg2 target target_n2048 target_n1024 target_haar tree_random tree_pyramid cfg3 cfg3_sdwt swpt_db4 cfg4 cfg4_256 cfg4_1024 cfg5 bb ldb siwt dwt_long}   # subset: only the workloads whose kernels changed
for w in $WL; do
  bash tools/profile.sh $TAG $w pmc > /dev/null 2>&1
done
python tools/collect_evidence.py $TAG ${2:-r02} > /dev/null 2>&1      # profiles/traffic.json of this build, read by bench.py
for w in $WL; do
  timeout 600 python bench.py --workload $w > gpurun_out/bench_$TAG/$w.json 2> gpurun_out/bench_$TAG/$w.err
  tail -c 200 gpurun_out/bench_$TAG/$w.json; echo
done
ls gpurun_out/prof_$TAG
