#!/bin/bash
# rocprofv3 passes of one BASELINE configuration (run on the GPU box from the repo root):
#     bash scripts/profile_config.sh 4 gpurun_out/r05_cfg4
# then  python scripts/summarize_config_profile.py gpurun_out/r05_cfg4 profiles/r05_cfg4
#   bench.json              python3 bench.py --config N                       (the line, un-profiled)
#   per_op.txt              python3 bench.py --config N --per-op ...          (hipEvent time of every launch, alone / in the step)
#   trace_graph/            rocprofv3 --kernel-trace --stats of the hipGraph replay (what the line times)
#   pmc_fetch/ pmc_write/   rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, each in its own run with --kernel-trace only
CFG=${1:-4}
OUT=${2:-gpurun_out/r06_cfg$CFG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --config $CFG > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --config $CFG --per-op --no-cpu-baseline --no-parity --no-modes --stream 0 --graph 0 --no-frame-loop > $OUT/bench_per_op.json 2> $OUT/per_op.txt
COMMON="--config $CFG --steps 5 --warmup 3 --reps 3 --no-cpu-baseline --no-parity --no-modes --stream 0 --no-frame-loop --in-flight 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_graph -- python3 bench.py $COMMON --graph 1 > $OUT/bench_traced_graph.json 2> $OUT/trace_graph.err
PM="--config $CFG --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-parity --no-modes --stream 0 --graph 0 --no-detect --no-frame-loop --in-flight 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $PM > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py $PM > /dev/null 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head; tail -2 $OUT/bench.err
