#!/bin/bash
# usage: bash tools/pmc_sq.sh <tag> <workload>   -- SQ activity counters (two passes) for one workload
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $O/a -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --workload $2 > $O/a.json 2> $O/a.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --workload $2 > $O/b.json 2> $O/b.err
cd $R && python3 - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % sub):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k in acc:
            if "wx" in k or "k_" in k:
                print(k, {c: "%.3g" % (v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
