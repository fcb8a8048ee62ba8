#!/bin/bash
# usage (GPU box, repo root): bash tools/fp64_issue.sh <tag>
# Is the F = 16 lattice kernel bound by FP64 issue?  GRBM_GUI_ACTIVE (GPU-clock cycles of the launch) and SQ_INSTS_VALU per launch for
# the lattice kernels of config 2 / its wpt form / the db4 target, and for the pure FP64-FMA loop of tools/ubench.hip, in separate
# rocprofv3 --pmc passes (only --kernel-trace beside them); summary by tools/fp64_issue.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/fp64_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in cfg2 wpt_db8 target; do
  i=0
  for set in "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${w}_p$i -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu --no-also > $O/${w}_p$i.json 2> $O/${w}_p$i.err
  done
done
i=0
for set in "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/ubench_p$i -- $R/tools/bin/ubench > $O/ubench_p$i.log 2> $O/ubench_p$i.err
done
cd $R && python3 tools/fp64_issue.py $O
