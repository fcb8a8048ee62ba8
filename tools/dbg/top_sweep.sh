#!/bin/bash
# sweep of the resident workgroups per CU of the top pass on the GPU box: bash tools/dbg/top_sweep.sh
for w in 1 2 3 4 6; do
  echo "== WX_TOPTILE_WGS=$w"
  WX_TOPTILE_WGS=$w python tools/dbg/long_dwt_time.py db4 2>/dev/null | grep -v "n  4096"
done
