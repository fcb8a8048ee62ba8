export WX_ROWS_P4=0
for rs in "32 32" "32 36" "32 40" "32 48" "16 16" "16 20" "16 24" "64 64"; do
  set -- $rs
  echo "== f32 256 L8 R=$1 S=$2"; WX_ROWS_R=$1 WX_ROWS_S=$2 bash tools/dbg/prof_script.sh p2d_s tools/dbg/one2d.py f32 256 8 full | grep rows_fused | cut -c1-40,150-220
done
for rs in "16 24" "16 16" "16 20" "16 28" "8 12" "8 8" "32 40"; do
  set -- $rs
  echo "== f64 256 L8 R=$1 S=$2"; WX_ROWS_R=$1 WX_ROWS_S=$2 bash tools/dbg/prof_script.sh p2d_s tools/dbg/one2d.py f64 256 8 full | grep rows_fused | cut -c1-40,150-220
done
bash tools/dbg/pmc_script.sh p2d_pmc k_rows_fused tools/dbg/one2d.py f32 256 8 full fwd | grep -E "k_rows|LDS|VALU|WAVE_CYCLES|BUSY|WAIT" 
