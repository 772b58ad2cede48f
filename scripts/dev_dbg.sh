for V in 5 6; do echo "== GPU_MAX_HW_QUEUES=$V"; GPU_MAX_HW_QUEUES=$V timeout 300 python scripts/dev_stream.py 2>&1 | grep "resident graph\|pipe slots=2 full\|no copies\|pipe slots=4"; done
echo "== TDRN_STREAMS=1"; TDRN_STREAMS=1 timeout 300 python scripts/dev_stream.py 2>&1 | grep "resident graph\|pipe slots=2 full\|no copies\|pipe slots=4"
