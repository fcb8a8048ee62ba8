#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05o
mkdir -p $O
python bench.py --workload target_f32 --no-cpu --no-also > $O/bench_target_f32.json 2>$O/err.txt; tail -c 300 $O/err.txt
python3 -c "
import json; j=json.loads([l for l in open('gpurun_out/r05o/bench_target_f32.json') if l.startswith('{')][-1]); print('target_f32', j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['inverse']['avg_launch_ms'], j['inverse']['frac'], j['roundtrip_rel_err'])"
bash tools/pmc_issue.sh r05o target_f32 > $O/pmc_target_f32.txt 2>&1; cat $O/pmc_target_f32.txt | tail -20
bash tools/pmc_issue.sh r05o target > $O/pmc_target.txt 2>&1; cat $O/pmc_target.txt | tail -20
find gpurun_out/pmc_r05o_* -name "*.csv" -size +1M -delete
