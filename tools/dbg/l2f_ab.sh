#!/bin/bash
# fused 2-D launch against the two launches, per geometry (A/B by knob on one box)
for w in cfg4 cfg4_256 cfg4_1024; do for f in 1 0; do
  WX_KNOBS=1 WX_L2D_FUSED=$f timeout 300 python bench.py --workload $w --no-cpu --no-also 2>/dev/null | tail -1 > /tmp/l2f.json
  python -c "
import json; d=json.loads(open('/tmp/l2f.json').read())
print('$w fused=$f: step %.3f ms  fwd %.3f ms (%.3f)  inv %.3f ms (%.3f)' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['inverse']['avg_launch_ms'], d['inverse']['frac']))"
done; done
