#!/bin/bash
# kernel breakdown of a script per argument: bash tools/dbg/prof_any.sh <script.py> <arg> [<arg> ...] -> gpurun_out/any_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=$1; shift
rm -f $R/gpurun_out/any_stats.txt
for n in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/any$n -- python3 $R/$S $n > $R/gpurun_out/any$n.log 2>&1
  echo "== $S $n" >> $R/gpurun_out/any_stats.txt
  f=$(find $R/gpurun_out/any$n -name '*kernel_stats.csv' | head -1)
  python3 - "$f" >> $R/gpurun_out/any_stats.txt <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print("%-110s %4s %9.1f us" % (r['Name'].replace('(anonymous namespace)::', '')[:110], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
cat $R/gpurun_out/any_stats.txt
