#!/bin/bash
# usage (GPU box, repo root): bash tools/dbg/trace_cfg3.sh <tag> [workload] [extra bench args]
# HIP API + kernel trace of a short bench run (where does the time between the launches of one inverse leg go);
# tools/dbg/trace_gaps.py turns the two CSVs into a timeline.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; WL=${2:-cfg3}; shift; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --workload $WL --steps 3 --warmup 2 --no-cpu --no-also "$@" > $O/bench.json 2> $O/run.err
cd $R
python3 tools/dbg/trace_gaps.py $O/trace > $O/gaps.txt 2>&1
tail -60 $O/gaps.txt
