for cfg in "16 512" "16 256" "32 512" "8 256" "8 512" "64 512"; do
  set -- $cfg
  echo "LDS=$1 NT=$2"
  WX_SWTINV_LDS_KIB=$1 WX_SWTINV_NT=$2 timeout 300 python bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['inverse']['avg_launch_ms'], d['roofline']['avg_launch_ms'], d['roundtrip_rel_err'])"
done
timeout 600 python -m pytest tests/test_gpu_redundant.py -x -q -m gpu 2>&1 | tail -3
