#!/bin/bash
# usage (GPU box, repo root): bash tools/refresh_evidence.sh <tag> [round-prefix, default r02] (WX_EVIDENCE_NOTEST=1 skips the test-suite)
# full GPU test log, the three rocprofv3 passes per workload, then one bench line per workload (with the CPU
# baseline) reading the traffic table just measured; everything lands under gpurun_out/ and
# tools/collect_evidence.py files it under profiles/
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
mkdir -p $R/gpurun_out/bench_$TAG
cd $R
if [ -z "$WX_EVIDENCE_NOTEST" ]; then timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu_$TAG.log | tail -1; fi
WL=${WX_EVIDENCE_WORKLOADS:-cfg2 wpt_db8 target target_n2048 target_n1024 target_haar tree_random tree_pyramid cfg3 cfg3_sdwt swpt_db4 cfg4 cfg4_256 cfg4_1024 cfg5 bb ldb siwt dwt_long}   # subset: only the workloads whose kernels changed
for w in $WL; do
  bash tools/profile.sh $TAG $w pmc > /dev/null 2>&1
done
python tools/collect_evidence.py $TAG ${2:-r02} > /dev/null 2>&1      # profiles/traffic.json of this build, read by bench.py
for w in $WL; do
  NA=--no-also; [ $w = cfg2 ] && NA=
  timeout 600 python bench.py --workload $w $NA > gpurun_out/bench_$TAG/$w.json 2> gpurun_out/bench_$TAG/$w.err
  tail -c 200 gpurun_out/bench_$TAG/$w.json; echo
done
ls gpurun_out/prof_$TAG
python tools/collect_evidence.py $TAG ${2:-r02} > /dev/null 2>&1      # again: files the bench lines
# the GPU box only hands gpurun_out/ back: a copy of what collect_evidence.py filed under profiles/
P=${2:-r02}
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${P}_*.md profiles/${P}_*.json profiles/traffic.json gpurun_out/profiles_$TAG/ 2>/dev/null
# the traces themselves stay on the box (gpurun_out is limited to 64 MiB)
find gpurun_out/prof_$TAG -name "*.csv" -size +2M -delete
