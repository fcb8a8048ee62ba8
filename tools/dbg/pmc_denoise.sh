#!/bin/bash
# executed-instruction counters of the one-pass denoise kernel: bash tools/dbg/pmc_denoise.sh <n>  -> gpurun_out/dn_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=${1:-4096}
: > $R/gpurun_out/dn_pmc.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_ANY"; do
  d=$R/gpurun_out/dnpmc_$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $R/tools/dbg/prof_denoise.py $n > $d.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  python3 - "$f" >> $R/gpurun_out/dn_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'denoise' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print("%-24s %16.0f (mean of %d launches)" % (k, sum(v) / len(v), len(v)))
PY
done
cat $R/gpurun_out/dn_pmc.txt
