#!/usr/bin/env python3
"""coo_probe.py [reps]: BASELINE config 5 (density-0.1 COO x dense, fp32, b = 32) on four ResNet-50 shapes through
sm_spmm_coo_f32_packed and sm_spmm_coo_f32_ws, for rocprofv3 --kernel-trace --stats (per-kernel times of the re-ordering
passes and of the product)."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
L_ = sm.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
for (m, n, k, b) in [(784, 256, 2304, 32), (12544, 64, 576, 32), (196, 512, 4608, 32), (3136, 128, 1152, 32)]:
    idx = (torch.rand(m, k, generator=g, device=dev) < 0.1).nonzero()
    r, c = idx[:, 0].to(torch.int32).contiguous(), idx[:, 1].to(torch.int32).contiguous()
    nnz = int(r.numel())
    v = (torch.rand(nnz, generator=g, device=dev) * 2 - 1).float()
    B = torch.empty(b * k * n, dtype=torch.float32, device=dev)
    sm.fill_uniform(B, 55 + n, -1.0, 1.0)
    C = torch.empty(b * m * n, dtype=torch.float32, device=dev)
    nb = ctypes.c_size_t(0)
    L_.sm_spmm_coo_packed_workspace_size(m, nnz, ctypes.byref(nb))
    ws = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
    for _ in range(reps):
        assert L_.sm_spmm_coo_f32_packed(m, k, nnz, n, b, r.data_ptr(), c.data_ptr(), v.data_ptr(), B.data_ptr(), C.data_ptr(), 1.0, 0.0,
                                          ws.data_ptr(), nb.value, None) == 0
    nb1 = ctypes.c_size_t(0)
    L_.sm_spmm_coo_workspace_size(m, ctypes.byref(nb1))
    ws1 = torch.zeros(nb1.value, dtype=torch.uint8, device=dev)
    for _ in range(reps):
        assert L_.sm_spmm_coo_f32_ws(m, k, nnz, n, b, r.data_ptr(), c.data_ptr(), v.data_ptr(), B.data_ptr(), C.data_ptr(), 1.0, 0.0,
                                     ws1.data_ptr(), None) == 0
    torch.cuda.synchronize()
    print(m, n, k, b, nnz, "flag", int(ws[:4].view(torch.int32)[0]))
