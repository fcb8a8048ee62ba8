#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_probe.sh <tag> <workload> "<CTR1 CTR2 ...>" ["<CTR...>" ...]
# one rocprofv3 --pmc pass per quoted counter group over `bench.py --workload <w> --steps 2`; prints per-kernel averages
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; W=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  O=$R/gpurun_out/pmcprobe_$TAG/g$i
  mkdir -p $O
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu > $O/bench.json 2> $O/err.txt
  i=$((i+1))
done
cd $R && python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcprobe_$TAG/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_" in k[:60] and "at::" not in k:
            agg[k.replace("(anonymous namespace)::", "").replace("void ", "")[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %.4g  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
