#!/bin/bash
# usage (GPU box, repo root): bash tools/dbg/prof_script.sh <tag> <script.py> [args...]: rocprofv3 kernel stats of a python script
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/"$@" > $O/run.log 2> $O/run.err
cd $R
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:25]:
    print("%-110s calls %5s avg %9.3f ms total %9.2f ms" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
