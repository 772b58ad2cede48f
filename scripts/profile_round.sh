#!/bin/bash
# Profiles of the benchmark workload on one MI355X (run on the GPU box from the repo root):
#     bash scripts/profile_round.sh gpurun_out/r02_final
# then  python scripts/summarize_profile.py gpurun_out/r02_final profiles/r02_final
# Passes (counters in their own runs, never combined with other trace domains):
#   bench.json            python bench.py                                   (the line the driver records; un-profiled)
#   per_op.txt            python bench.py --per-op ...                      (hipEvent time of every launch: alone / in the step)
#   trace/                rocprofv3 --kernel-trace --stats                  (kernel durations in the timed loop, eager launches)
#   trace_graph/          the same of `bench.py --graph 1`                  (the hipGraph REPLAY: what the bench line times)
#   pmc_fetch/ pmc_write/ rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE           (HBM-side traffic per dispatch)
#   pmc_sq/               rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE              (wave cycles, waits, matrix-pipe busy, LDS conflicts, clock)
OUT=${1:-gpurun_out/r06_final}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --graph 0 > $OUT/bench_per_op.json 2> $OUT/per_op.txt
COMMON="--steps 5 --warmup 3 --reps 3 --no-cpu-baseline --no-parity --no-modes --stream 0 --in-flight 1"      # (kernel durations: one pipeline; bench.json above is the default two-in-flight line)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $COMMON --graph 0 > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_graph -- python3 bench.py $COMMON --graph 1 > $OUT/bench_traced_graph.json 2> $OUT/trace_graph.err
PM="--steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-parity --no-modes --stream 0 --graph 0 --no-detect --in-flight 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $PM > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py $PM > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py $PM > /dev/null 2> $OUT/pmc_sq.err
find $OUT -name "*.csv" | head -20
