python -m pytest tests/test_gpu_dwt2d.py tests/test_gpu_bench_geometry.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
for c in "f32 1024 10 full" "f32 256 8 full" "f64 256 8 full" "f64 1024 10 full" "f32 64 6 full"; do
  t=$(echo $c | tr ' ' '_')
  echo "== $c"; bash tools/dbg/prof_script.sh p2d_$t tools/dbg/one2d.py $c | grep rows_fused
  echo "== $c REGL=0"; WX_ROWS_REGL=0 bash tools/dbg/prof_script.sh p2d0_$t tools/dbg/one2d.py $c | grep rows_fused
done
