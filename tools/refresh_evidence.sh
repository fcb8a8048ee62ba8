#!/bin/bash
# usage (GPU box, repo root): bash tools/refresh_evidence.sh <tag>
# full GPU test log, one bench line per workload (with the CPU baseline) and the three rocprofv3 passes
# per workload, all under gpurun_out/; tools/collect_evidence.py then files them under profiles/
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
mkdir -p $R/gpurun_out/bench_$TAG
cd $R
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -2 gpurun_out/pytest_gpu_$TAG.log
for w in cfg2 target cfg3 cfg4 cfg5 bb; do
  timeout 600 python bench.py --workload $w > gpurun_out/bench_$TAG/$w.json 2> gpurun_out/bench_$TAG/$w.err
  tail -c 300 gpurun_out/bench_$TAG/$w.json; echo
  bash tools/profile.sh $TAG $w pmc > /dev/null 2>&1
done
ls gpurun_out/prof_$TAG
