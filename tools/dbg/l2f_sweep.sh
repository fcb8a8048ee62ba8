#!/bin/bash
# config 4 through the fused 2-D launch for ring geometries (G units per group, second pass D groups behind, ring of K groups)
mkdir -p gpurun_out/r06
for cfg in "8 8 16" "8 4 8" "8 12 24" "16 4 8" "16 8 16" "4 16 24" "32 2 4" "32 3 6" "64 2 4" "8 8 12" "2 16 24"; do
  set -- $cfg
  WX_KNOBS=1 WX_L2F_G=$1 WX_L2F_D=$2 WX_L2F_K=$3 timeout 300 python bench.py --workload cfg4 --no-cpu --no-also 2>/dev/null | tail -1 > /tmp/l2f.json
  python -c "
import json; d=json.loads(open('/tmp/l2f.json').read())
print('G=$1 D=$2 K=$3 (ring %d MiB): step %.3f ms  fwd %.3f ms (%.3f)  inv %.3f ms (%.3f)' % ($1*$3, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['inverse']['avg_launch_ms'], d['inverse']['frac']))"
done
