echo "== plain"; timeout 300 python scripts/dev_dbg_modes.py "bf16,fp16,fp32" x 2>&1 | grep "replayed\|ok \|done\|Fatal"
echo "== bf16,bf16,fp32,fp16,fp32"; timeout 300 python scripts/dev_dbg_modes.py "bf16,bf16,fp32,fp16,fp32" x 2>&1 | grep "replayed\|ok \|done\|Fatal"
