import sys, os
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
def run(m, n, k, b):
    A = torch.empty(b * m * k, dtype=torch.float32, device=dev); sm.fill_uniform(A, 1, -1.0, 1.0)
    B = torch.empty(k * n, dtype=torch.float32, device=dev); sm.fill_uniform(B, 2, -1.0, 1.0)
    C = torch.empty(b * m * n, dtype=torch.float32, device=dev)
    f = lambda: sm.gemm_rowmajor(A, B, C, m, n, k, batch=b)
    ms = sm.graph_time_ms(f, iters=10)
    print(f"{m}x{n}x{k} b={b}: {ms:.3f} ms  {2.0*m*n*k*b/ms/1e9:.1f} TF/s", flush=True)
for s in [(4096, 4096, 4096, 1), (8192, 8192, 2048, 1), (3136, 128, 1152, 32), (784, 256, 2304, 32), (12544, 64, 576, 32), (196, 512, 4608, 32)]:
    run(*s)
