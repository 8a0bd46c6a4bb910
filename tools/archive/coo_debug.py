#!/usr/bin/env python3
"""debug aid: sm_spmm_coo_f32_fast (beta == 0) on integer data against numpy, error map by (row block, column block)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, importlib
import __graft_entry__ as ge
sm = ge.load_package()
L = sm.lib()
def run(m, n, k, b, dens, seed=1, strips=False):
    rng = np.random.default_rng(seed)
    d = rng.uniform(0, 1, (m, k)) < dens
    r, c = np.nonzero(d); r = r.astype(np.int32); c = c.astype(np.int32)
    v = rng.integers(1, 9, r.size).astype(np.float32)
    B = rng.integers(-8, 9, b * k * n).astype(np.float32)
    A = np.zeros((m, k)); np.add.at(A, (r, c), v)
    want = np.concatenate([(B[i*k*n:(i+1)*k*n].astype(np.float64).reshape(n, k) @ A.T).reshape(-1) for i in range(b)])
    nb = ctypes.c_size_t(0); L.sm_spmm_coo_fast_workspace_size(m, k, n, b, ctypes.byref(nb))
    ws = torch.zeros(nb.value, dtype=torch.uint8, device="cuda")
    t = lambda a: torch.from_numpy(a).cuda()
    dr, dc, dv, dB = t(r), t(c), t(v), t(B)
    dC = torch.full((b*m*n,), 3.0, device="cuda")
    rc = L.sm_spmm_coo_f32_fast(m, k, r.size, n, b, dr.data_ptr(), dc.data_ptr(), dv.data_ptr(), dB.data_ptr(), dC.data_ptr(), 1.0, 0.0, ws.data_ptr(), nb.value, None)
    torch.cuda.synchronize()
    got = dC.cpu().numpy().astype(np.float64)
    bad = (got != want).reshape(b * n, m)
    strips3 = sum(int((d[:, j:j+4].sum(axis=1) > 2).sum()) for j in range(0, k - k % 4, 4))
    print(f"{m}x{n}x{k}x{b} dens {dens}: form {L.sm_spmm_coo_fast_form(m,k,r.size,n,b,0.0)} rc {rc} wrong {int(bad.sum())} of {bad.size}; strips with >2: {strips3}")
    ru = lambda x, a: (x + a - 1) // a * a
    kc = ru(k, 64); nst = kc // 64; tm = (m + 127) // 128
    o = 256 + ru(m * kc * 4, 256); o_hi = o; o += ru(nst * m * 64, 256); o_lo = o; o += ru(nst * m * 64, 256); o_meta = o; o += ru(nst * m * 8, 256); o_cnt = o
    o += ru(tm * nst * 4, 256); o_list = o
    w = ws.cpu().numpy()
    cnt = w[o_cnt:o_cnt + tm * nst * 4].view(np.int32)
    print("   hdr", w[:20].view(np.int32), "rcount", cnt[:8], "sum", cnt.sum())
    if cnt.sum():
        b0 = int(np.nonzero(cnt)[0][0])
        ent = w[o_list + b0 * 2048: o_list + b0 * 2048 + 8 * min(4, cnt[b0])].view(np.uint32).reshape(-1, 2)
        for e in ent:
            print("   bucket", b0, "row", e[0] & 127, "k", e[0] >> 8, "val", np.array([e[1]], dtype=np.uint32).view(np.float32)[0])
    if bad.any():
        rows = np.nonzero(bad.any(axis=0))[0]; cols = np.nonzero(bad.any(axis=1))[0]
        print("   wrong rows:", rows[:40], "... n", rows.size, " wrong cols:", cols[:40], "n", cols.size)
        i, j = np.argwhere(bad)[0]
        print("   first: col", i, "row", j, "got", got.reshape(b*n, m)[i, j], "want", want.reshape(b*n, m)[i, j])
for a in [(8, 8, 64, 1, 0.1), (32, 16, 64, 1, 0.1), (128, 16, 128, 1, 0.1), (132, 16, 64, 1, 0.02), (256, 33, 128, 3, 0.15), (388, 150, 328, 2, 0.17), (260, 72, 147, 3, 0.12)]:
    run(*a)
