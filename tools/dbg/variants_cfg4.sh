#!/bin/bash
# usage: bash tools/dbg/variants_cfg4.sh <lib-name>...   -- forward / inverse launch times of config 4 per library variant
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib=$PWD/tools/dbg/lib/libwx_$v.so
  WX_HIP_LIB=$lib python bench.py --workload cfg4 --no-cpu --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', 'fwd %.3f ms' % d['roofline']['avg_launch_ms'], 'inv %.3f ms' % d['inverse']['avg_launch_ms'], 'rt err %.2e' % d['roundtrip_rel_err'])"
done
