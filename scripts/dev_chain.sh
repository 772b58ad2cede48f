timeout 60 python scripts/dev_chain1.py 32 2>&1 | grep -c equal
run() { timeout 150 python bench.py --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
run chain
TDRN_CHAIN=0 run nochain
done
timeout 120 python scripts/dev_timeline.py 2>&1 | grep -v amdgpu.ids | sed -n 18,60p
