mkdir -p gpurun_out/r05a
python -m pytest tests/test_gpu_cfg5_partials.py tests/test_gpu_bench_geometry.py -m gpu -x -q > gpurun_out/r05a/pytest_cfg5.log 2>&1; echo "pytest rc $?"
tail -5 gpurun_out/r05a/pytest_cfg5.log
bash tools/dbg/trace_cfg3.sh r05a/trace_cfg3 cfg3 --batch 64 
python bench.py > gpurun_out/r05a/bench_default.json 2> gpurun_out/r05a/bench_default.err; echo "bench rc $?"
python bench.py --workload cfg3 --batch 64 --steps 10 --no-cpu --no-also > gpurun_out/r05a/bench_cfg3_b64.json 2>&1
python tools/bench_table.py gpurun_out/r05a/bench_default.json 2>/dev/null | head -30
