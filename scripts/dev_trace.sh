OUT=gpurun_out/r03_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity --no-modes --graph 0 --reps 1 > $OUT/bench_traced.json 2> $OUT/trace.err
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03_trace/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-90s calls %5s avg %9.1f us total %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
