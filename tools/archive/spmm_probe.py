#!/usr/bin/env python3
"""Device time of the two SpMM baselines (COO with workspace, Blocked-ELL batched) on a few ResNet-50 shapes at
b = 32, fp32, 90 % sparse COO / 50 % dense 2x2 Blocked-ELL as the drivers build them: tools/spmm_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
sm = ge.load_package()
L = sm.lib()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
for (m, n, k, b) in [(784, 256, 2304, 32), (12544, 64, 576, 32), (196, 512, 4608, 32), (3136, 128, 1152, 32)]:
    dense = rng.uniform(0, 1, (m, k)) < 0.1
    r, c = np.nonzero(dense)
    nnz = r.size
    dr, dc = torch.from_numpy(r.astype(np.int32)).to(dev), torch.from_numpy(c.astype(np.int32)).to(dev)
    dv = torch.rand(nnz, dtype=torch.float32, device=dev) - 0.5
    B = torch.rand(b * k * n, dtype=torch.float32, device=dev) - 0.5
    C = torch.zeros(b * m * n, dtype=torch.float32, device=dev)
    nb = ctypes.c_size_t(0)
    L.sm_spmm_coo_workspace_size(m, ctypes.byref(nb))
    ws = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
    def f():
        rc = L.sm_spmm_coo_f32_ws(m, k, nnz, n, b, dr.data_ptr(), dc.data_ptr(), dv.data_ptr(), B.data_ptr(), C.data_ptr(), 1.0, 0.0,
                                  ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    ms = sm.graph_time_ms(f, iters=5)
    gb = (B.numel() + C.numel()) * 4 / 1e9
    print(f"coo {m}x{n}x{k} b={b} nnz={nnz}: {ms:.3f} ms  {2.0*nnz*n*b/ms/1e9:.2f} TF/s  {gb/ms*1e3:.0f} GB/s (B read + C write)", flush=True)

# Blocked-ELL (2 x 2 blocks, half of the block columns present), all batches in one submission
PtrArr = None
for (m, n, k, b) in [(784, 256, 2304, 32), (3136, 128, 1152, 32), (196, 512, 4608, 32)]:
    bs, ell_cols = 2, k // 2
    bcols = ell_cols // bs
    keep = []
    for _ in range(b):
        ci = np.stack([np.sort(rng.choice(k // bs, bcols, replace=False)) for _ in range(m // bs)]).astype(np.int64)
        keep.append((torch.rand(m * ell_cols, dtype=torch.float32, device=dev) - 0.5, torch.from_numpy(ci.reshape(-1)).to(dev)))
    B = torch.rand(k * n, dtype=torch.float32, device=dev) - 0.5
    Cs = [torch.zeros(m * n, dtype=torch.float32, device=dev) for _ in range(b)]
    nb = ctypes.c_size_t(0)
    L.sm_spmm_bell_batched_workspace_size(m, k, b, ctypes.byref(nb))
    ws = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
    PA = ctypes.c_void_p * b
    pv, pi, pc = PA(*[v.data_ptr() for v, _ in keep]), PA(*[i.data_ptr() for _, i in keep]), PA(*[c.data_ptr() for c in Cs])
    def g():
        rc = L.sm_spmm_bell_batched_f32(pv, pi, m, k, bs, ell_cols, B.data_ptr(), pc, n, b, 1.0, 0.0, ws.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    g(); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5): g()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3   # (host pointer tables: not graph-capturable, wall clock over 5 calls)
    print(f"bell {m}x{n}x{k} b={b}: {ms:.3f} ms  {2.0*m*n*k*b/ms/1e9:.2f} dense-TF/s  {1.0*m*n*k*b/ms/1e9:.2f} TF/s of stored values", flush=True)
