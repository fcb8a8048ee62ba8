#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_issue.sh <tag> <workload> [bench args]   -- issue / wait counters of a workload's
# kernels, separate rocprofv3 --pmc passes (no trace domains beside --kernel-trace); summary by tools/pmc_summary.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$1_$2
mkdir -p $O
W=$2
shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-also "$@" > $O/p$i.json 2> $O/p$i.err
done
cd $R && python3 tools/pmc_summary.py $O k_lat
