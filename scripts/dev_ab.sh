# harness + three interleaved bench lines (bench.py default graph mode, no extras)
bash scripts/dev_conv_check.sh | tail -12
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --no-parity --no-modes --stream 0 --reps 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
