#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_bytes.sh <tag> <workload> [kernel-substring]
# memory-side request counters of a workload's kernels in separate rocprofv3 --pmc passes (only --kernel-trace beside them): the
# derived FETCH_SIZE / WRITE_SIZE and the raw L2 -> fabric requests they come from, to tell 32- from 64- and 128-byte requests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmcb_$1_$2
mkdir -p $O
W=$2
K=${3:-k_}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-also > $O/p$i.json 2> $O/p$i.err
done
cd $R && python3 tools/pmc_summary.py $O $K
