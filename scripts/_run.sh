mkdir -p gpurun_out/r2
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2/t_all.log 2>&1
python scripts/drift_table.py --out gpurun_out/r2/drift > gpurun_out/r2/drift.log 2>&1
python bench.py --per-op > gpurun_out/r2/bench_d.json 2> gpurun_out/r2/bench_d.err
