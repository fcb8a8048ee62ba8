#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile.sh <tag> <workload> [pmc]
# writes gpurun_out/prof_<tag>/<workload>/{trace,pmc_fetch,pmc_write}; summarise with tools/summarize_prof.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$1/$2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --workload $2 --steps 20 --warmup 3 --no-cpu > $O/bench_trace.json 2> $O/trace.err
if [ "$3" = "pmc" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --workload $2 --steps 2 --warmup 1 --no-cpu > $O/bench_fetch.json 2> $O/fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --workload $2 --steps 2 --warmup 1 --no-cpu > $O/bench_write.json 2> $O/write.err
fi
cd $R && python3 tools/summarize_prof.py $O $O/summary > /dev/null 2>&1; head -40 $O/summary.md
