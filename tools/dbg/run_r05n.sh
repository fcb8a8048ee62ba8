#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05n
mkdir -p $O
python -m pytest tests/test_gpu_cfg5_partials.py tests/test_gpu_bench_geometry.py tests/test_gpu_redundant.py tests/test_gpu_golden.py tests/test_gpu_stress.py tests/test_gpu_lattice_8k.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 $R/bench.py --workload cfg5 --batch 8192 --steps 5 --warmup 2 --no-cpu --no-also > $O/bench_cfg5_prof.json 2>/dev/null
cd $R
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/r05n/prof_cfg5/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:6]:
    print("%-100s calls %5s avg %9.3f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e6))
j = json.loads([l for l in open("gpurun_out/r05n/bench_cfg5_prof.json") if l.startswith("{")][-1])
print("cfg5 8192: fwd avg", j["roofline"]["avg_launch_ms"], "frac", j["roofline"]["frac"], "gap", j["config"].get("min_rel_cost_gap"))
PY
find $O -name "*kernel_trace.csv" -delete
python bench.py --workload cfg5 --no-cpu --no-also > $O/bench_cfg5_full.json 2>/dev/null; python3 -c "
import json; j=json.loads([l for l in open('gpurun_out/r05n/bench_cfg5_full.json') if l.startswith('{')][-1]); print('cfg5 full', j['ms_per_step'], j['roofline']['frac'], j['config'].get('min_rel_cost_gap'), j['config'].get('tree_split_nodes'))"
