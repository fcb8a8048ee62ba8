#!/bin/bash
# GPU run r05b: (1) host-path probe, (2) cfg3 inverse: run-to-run spread of one resident chunk, plain and under rocprofv3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05b
mkdir -p $O
timeout 600 tools/dbg/hostpath_probe 3.5 > $O/hostpath_probe.txt 2>&1
cat $O/hostpath_probe.txt
for i in 1 2 3 4 5 6; do
  python bench.py --workload cfg3 --batch 64 --steps 10 --no-cpu --no-also > $O/cfg3_plain_$i.json 2>/dev/null
done
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3_prof_$i -- python3 $R/bench.py --workload cfg3 --batch 64 --steps 10 --no-cpu --no-also > $O/cfg3_prof_$i.json 2>/dev/null
done
cd $R
python3 - <<'PY'
import json, glob, csv, os
O = "gpurun_out/r05b"
for f in sorted(glob.glob(O + "/cfg3_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), "fwd %.3f ms  inv %.3f ms (frac %.3f)" % (j["roofline"]["avg_launch_ms"], j["inverse"]["avg_launch_ms"], j["inverse"]["frac"]))
    except Exception as e:
        print(f, "ERR", e)
for d in sorted(glob.glob(O + "/cfg3_prof_?")):
    fs = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if fs:
        rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: -float(r["TotalDurationNs"]))
        print(os.path.basename(d), " | ".join("%s x%s avg %.3f" % (r["Name"][:28], r["Calls"], float(r["AverageNs"]) / 1e6) for r in rows[:4]))
PY
# keep the merge small
find $O -name "*kernel_trace.csv" -delete
