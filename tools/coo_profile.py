#!/usr/bin/env python3
"""sm_spmm_coo_f32_fast a few times on chosen config-5 shapes, for `rocprofv3 --kernel-trace --stats -- python3 tools/coo_profile.py m,n,k ...`
(per-kernel durations of the call's scan / scatter / image / matrix kernels); env COO_ABLATE -> SM_COO_ABLATE of the tuning library."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
sm = ge.load_package()
L = sm.lib()
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(12544, 64, 576), (196, 512, 4608), (3136, 128, 1152)]
b = 32
for (m, n, k) in shapes:
    rng = np.random.default_rng(m + k)
    d = rng.uniform(0, 1, (m, k)) < 0.1
    r, c = np.nonzero(d)
    v = rng.uniform(-1, 1, r.size).astype(np.float32)
    dr, dc, dv = torch.from_numpy(r.astype(np.int32)).cuda(), torch.from_numpy(c.astype(np.int32)).cuda(), torch.from_numpy(v).cuda()
    B = torch.empty(b * k * n, dtype=torch.float32, device="cuda"); sm.fill_uniform(B, 3, -1.0, 1.0)
    C = torch.empty(b * m * n, dtype=torch.float32, device="cuda")
    nb = ctypes.c_size_t(0); L.sm_spmm_coo_fast_workspace_size(m, k, n, b, ctypes.byref(nb))
    ws = torch.zeros(nb.value, dtype=torch.uint8, device="cuda")
    for it in range(4):
        rc = L.sm_spmm_coo_f32_fast(m, k, r.size, n, b, dr.data_ptr(), dc.data_ptr(), dv.data_ptr(), B.data_ptr(), C.data_ptr(), 1.0, 0.0, ws.data_ptr(), nb.value, None)
        assert rc == 0
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for it in range(5):
        L.sm_spmm_coo_f32_fast(m, k, r.size, n, b, dr.data_ptr(), dc.data_ptr(), dv.data_ptr(), B.data_ptr(), C.data_ptr(), 1.0, 0.0, ws.data_ptr(), nb.value, None)
    t1.record(); torch.cuda.synchronize()
    print(f"{m}x{n}x{k}: {t0.elapsed_time(t1) / 5 * 1e3:.1f} us per call (eager)", flush=True)
