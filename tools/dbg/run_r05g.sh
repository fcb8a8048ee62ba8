#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05g
mkdir -p $O
python tools/dbg/pair_dbg.py > $O/pair_dbg.txt 2>&1; cat $O/pair_dbg.txt
python -m pytest tests/test_gpu_lattice_pairs.py tests/test_gpu_lattice_g32.py tests/test_gpu_lattice_tree32.py tests/test_gpu_lattice.py -m gpu -q > $O/pytest_f32.log 2>&1; echo "pytest rc $?"; tail -25 $O/pytest_f32.log
python -m pytest tests/test_gpu_cfg5_partials.py tests/test_gpu_bench_geometry.py tests/test_gpu_redundant.py tests/test_gpu_golden.py tests/test_gpu_stress.py -m gpu -x -q > $O/pytest_cfg5.log 2>&1; echo "pytest cfg5 rc $?"; tail -8 $O/pytest_cfg5.log
python bench.py --workload cfg5 --batch 8192 --steps 5 --warmup 2 --no-cpu --no-also > $O/bench_cfg5.json 2>$O/bench_cfg5.err; tail -c 600 $O/bench_cfg5.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r05g/bench_cfg5.json") if l.startswith("{")][-1])
print("cfg5 8192 signals: ms_per_step", j["ms_per_step"], "fwd avg", j["roofline"]["avg_launch_ms"], "frac", j["roofline"]["frac"], j["config"].get("min_rel_cost_gap"))
PY
