timeout -k 10 800 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -4
for cfg in 0 4x2 4x3 2x3; do echo "== pc=$cfg"; SM_SPMMA_PC=$cfg timeout -k 10 200 python tools/sweep.py --unique --only spmma 2>&1 | grep -E "spmma" | awk '{printf "%s/%s/%s:%s ", $1,$2,$3,$7} END{print ""}'; done
