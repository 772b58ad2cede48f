run() { timeout 600 python bench.py --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run base
TDRN_LATE_SIDE=2 run late2
TDRN_LATE_SIDE=2 TDRN_SIDE_GRID=192 run late2_g192
TDRN_LATE_SIDE=2 TDRN_SIDE_GRID=128 run late2_g128
TDRN_SIDE_GRID=192 run g192
TDRN_LATE_SIDE=0 run late0
done
