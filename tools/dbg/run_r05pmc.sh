#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof_r05w/wpd2d64; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/dbg/prof2d_wpd.py > $O/out.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/dbg/prof2d_wpd.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/dbg/prof2d_wpd.py > /dev/null 2>&1
cd $R && python3 tools/summarize_prof.py gpurun_out/prof_r05w/wpd2d64 gpurun_out/prof_r05w/r05_wpd2d64 > /dev/null 2>&1; grep "k_lat2d64_wpd\|HBM bytes" gpurun_out/prof_r05w/r05_wpd2d64.md | head; grep algorithmic $O/out.txt
find gpurun_out/prof_r05w -name "*.csv" -size +1M -delete
