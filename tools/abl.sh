timeout -k 10 800 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -2
timeout -k 10 300 python tools/sweep.py --unique --only spmma,gemm_rm,gemm,compress 2>&1 | tail -75
