cd /tmp && export TMPDIR=/tmp
for w in dwt_long cfg5; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kn_$w -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 2 --no-cpu --no-also > $GRAFT_REPO_ROOT/gpurun_out/kn_$w.json 2>/dev/null
f=$(find $GRAFT_REPO_ROOT/gpurun_out/kn_$w -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("%-100s calls %5s avg %9.3f ms total %9.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
done
