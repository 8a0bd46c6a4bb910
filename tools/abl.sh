timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout 300 -k "gemm or lane_maps" 2>&1 | tail -3
timeout -k 10 300 python tools/sweep.py --unique --only gemm 2>&1 | tail -20
