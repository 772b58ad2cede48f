#!/bin/bash
# Instruction mix and active / wait cycles of every conv dispatch of one eager forward (two rocprofv3 --pmc passes, counters only):
#     OUT=gpurun_out/pmc_mix bash scripts/pmc_instruction_mix.sh        (on the GPU box, from the repo root)
export OUT=${OUT:-gpurun_out/pmc_mix}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
PM="--steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-modes --stream 0 --graph 0 --no-detect --reps 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py $PM > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL --kernel-trace --output-format csv -d $OUT/b -- python3 bench.py $PM > /dev/null 2> $OUT/b.err
python3 - <<'PY'
import csv, glob, collections, os
for sub in ("a", "b"):
    f = glob.glob(os.environ["OUT"] + "/%s/**/*counter_collection.csv" % sub, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    agg = collections.OrderedDict()
    for r in rows:
        k = r["Kernel_Name"]
        if "conv3x3" not in k: continue
        agg.setdefault((r["Dispatch_Id"], k[11:64]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    items = list(agg.items())[-18:]
    for (d, k), c in items:
        if "pp_kernel" in k or "128, 16" in k or "128, 32" in k:
            print(d, k, " ".join("%s=%.3g" % (n.replace("SQ_", ""), v) for n, v in sorted(c.items())))
PY
