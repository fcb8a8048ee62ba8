#!/bin/bash
# kernel breakdown of denoiseall(x, :sig) per length -> gpurun_out/dn_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f $R/gpurun_out/dn_stats.txt
for n in ${@:-64 256 1024 2048 4096}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dn$n -- python3 $R/tools/dbg/prof_denoise.py $n > $R/gpurun_out/dn$n.log 2>&1
  echo "== n = $n" >> $R/gpurun_out/dn_stats.txt
  f=$(find $R/gpurun_out/dn$n -name '*kernel_stats.csv' | head -1)
  python3 - "$f" >> $R/gpurun_out/dn_stats.txt <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-100s %3s %9.1f us" % (r['Name'].replace('(anonymous namespace)::', '')[:100], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
cat $R/gpurun_out/dn_stats.txt
