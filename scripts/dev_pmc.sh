OUT=gpurun_out/r03_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
PM="--steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-modes --graph 0 --no-detect --reps 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py $PM > /dev/null 2> $OUT/pmc_sq.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r03_pmc/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# per dispatch: kernel name + counters; keep the LAST forward's conv dispatches
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if "conv3x3" not in k: continue
    key = (r["Dispatch_Id"], k[:60])
    agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
items = list(agg.items())[-18:]
for (d, k), c in items:
    wc = c.get("SQ_WAVE_CYCLES", 0)
    g = c.get("GRBM_GUI_ACTIVE", 1)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g / 8 * 1024) if g else 0
    print("%-8s %-60s mfma_busy %.3f wait_any %.2f wait_inst %.2f active %.2f ldsconf %.3f gui/8 %.0f" % (d, k, busy, c.get("SQ_WAIT_ANY", 0) / wc if wc else 0,
          c.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0, c.get("SQ_ACTIVE_INST_ANY", 0) / wc if wc else 0, c.get("SQ_LDS_BANK_CONFLICT", 0) / (g / 8 * 256) if g else 0, g / 8))
PY
