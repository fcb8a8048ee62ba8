#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05p
mkdir -p $O
python bench.py --workload target_f32 --no-cpu --no-also 2>/dev/null | python3 -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('target_f32', j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['inverse']['avg_launch_ms'], j['inverse']['frac'], j['roundtrip_rel_err'], j['roofline']['kernel'])"
python tools/floor_scan.py db4 f32 256 1024 4096 2>&1 | grep "full tree"
python -m pytest tests/test_gpu_lattice_pairs.py tests/test_gpu_lattice_g32.py -m gpu -q 2>&1 | tail -2
