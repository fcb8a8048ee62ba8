#!/bin/bash
# config 3's inverse (k_haar_iswpt<5,1>) over consecutive processes for block orders / residencies
for cfg in "0 0" "1 0" "0 2" "1 2" "0 4"; do
  set -- $cfg
  line="order=$1 wpe=$2:"
  for i in 1 2 3 4 5 6; do
    WX_KNOBS=1 WX_HAAR_ISWT_ORDER=$1 WX_HAAR_ISWT_WPE=$2 python bench.py --workload cfg3 --batch 64 --steps 10 --no-cpu --no-also 2>/dev/null | tail -1 > /tmp/c3.json
    line="$line $(python -c "import json; d=json.load(open('/tmp/c3.json')); print('%.2f' % d['inverse']['avg_launch_ms'])")"
  done
  echo "$line"
done
