#!/bin/bash
# 64 x 64 one-pass kernels and the lattice row pass: parity, A/B timings, kernel breakdown of the Float64 2-D cells
O=gpurun_out/r05q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_lattice2d64.py tests/test_gpu_lattice_pairs.py tests/test_gpu_lattice.py tests/test_gpu_dwt2d.py tests/test_gpu_2d_shapes.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 600 python tools/floor_scan2d.py db4 64 128 256 512 > $O/floor2d_new.txt 2>&1
WX_KNOBS=1 WX_LATROWS=0 WX_NO_2D64=1 timeout 600 python tools/floor_scan2d.py db4 64 128 256 512 > $O/floor2d_old.txt 2>&1
grep "full tree" $O/floor2d_new.txt; echo ---; grep "full tree" $O/floor2d_old.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_f64full -o p -- python3 $GRAFT_REPO_ROOT/tools/dbg/prof2d.py f64full 256 512 1024 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_f64full -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print("%-100s calls %5s  avg %10.1f us  total %6.2f %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY

echo "--- f32 pyramids"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_f32pyr -o p -- python3 $GRAFT_REPO_ROOT/tools/dbg/prof2d.py f32pyr 256 1024 > $GRAFT_REPO_ROOT/$O/prof2.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_f32pyr -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print("%-100s calls %5s  avg %10.1f us  total %6.2f %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
