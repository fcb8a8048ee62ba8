timeout 600 python -m pytest tests/test_gpu_dwt2d.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
for cfg in "32 32" "16 16" "16 20"; do
  set -- $cfg
  echo -n "R=$1 S=$2: "
  WX_KNOBS=1 WX_ROWS_R=$1 WX_ROWS_S=$2 timeout 300 python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['avg_launch_ms'], d['inverse']['avg_launch_ms'], d['roundtrip_rel_err'])" 2>&1 | tail -1
done
