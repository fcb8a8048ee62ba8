# usage (GPU box): bash tools/dbg/rows_sweep.sh  -- strip height / LDS pitch of the 2-D row pass (WX_ROWS_R, WX_ROWS_S), rocprofv3 kernel times
i=0
for c in "f32 256 8" "f32 64 6" "f32 1024 10" "f64 256 8" "f64 64 6"; do
  for rs in "0 0" "32 32" "32 36" "32 40" "16 16" "16 20" "16 24" "8 8" "8 12" "64 64"; do
    set -- $rs
    i=$((i+1))
    if [ $1 = 0 ]; then unset WX_ROWS_R WX_ROWS_S; else export WX_KNOBS=1 WX_ROWS_R=$1 WX_ROWS_S=$2; fi
    echo "== $c R=$1 S=$2: $(bash tools/dbg/prof_script.sh rs_$i tools/dbg/one2d.py $c full | grep rows_fused | sed -e 's/.*avg *//' -e 's/ ms total.*//' | tr '\n' ' ')"
  done
done
