timeout 600 python -m pytest tests/test_gpu_redundant.py -x -q -m gpu 2>&1 | tail -3
for cfg in "16" "8" "32" "64"; do
  echo "INV LDS=$cfg"
  WX_KNOBS=1 WX_SWTINV_LDS_KIB=$cfg timeout 300 python bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['inverse']['avg_launch_ms'], d['roofline']['avg_launch_ms'], d['roundtrip_rel_err'])"
done
