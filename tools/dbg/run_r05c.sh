#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05c
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
python tools/host_path_time.py > $O/host_path.txt 2>&1; cat $O/host_path.txt
for i in 1 2 3; do python tools/dbg/cfg3_placement.py > $O/cfg3_placement_$i.txt 2>&1; cat $O/cfg3_placement_$i.txt; done
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r05c/bench_default.json") if l.startswith("{")][-1])
print("value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"])
print("pcie", j.get("pcie_inclusive"))
for k, v in j.get("also", {}).items():
    print(k, {a: (round(v[a].get("avg_launch_ms", 0), 3), round(v[a].get("frac", 0), 3)) for a in ("fwd", "inv") if a in v}, v.get("min_rel_cost_gap"))
PY
