#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05h
mkdir -p $O
python -m pytest tests/test_gpu_lattice_pairs.py tests/test_gpu_lattice_g32.py tests/test_gpu_lattice_tree32.py tests/test_gpu_lattice.py tests/test_gpu_lattice_tree.py tests/test_gpu_dwt_long.py tests/test_gpu_toptile.py tests/test_gpu_fuzz.py tests/test_gpu_denoise.py tests/test_gpu_2d_shapes.py tests/test_gpu_dwt2d.py tests/test_gpu_dwt1d.py -m gpu -q > $O/pytest_f32.log 2>&1; echo "pytest rc $?"; tail -12 $O/pytest_f32.log
python tools/floor_scan.py db4 f32 64 256 1024 4096 > $O/floor_f32_db4.txt 2>&1; cat $O/floor_f32_db4.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 $R/bench.py --workload cfg5 --batch 8192 --steps 5 --warmup 2 --no-cpu --no-also > $O/bench_cfg5_prof.json 2>/dev/null
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r05h/prof_cfg5/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print("%-100s calls %5s avg %9.3f ms total %9.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
find $O -name "*kernel_trace.csv" -delete
