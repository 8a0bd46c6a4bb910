"""Host-side baselines: the oracle (or the reference build under oracle/_ref) timed on the box's cores -- the checker being measured, never the product."""
import time


def config1_cpu(ge):
    """BASELINE config 1 (examples/sparsify.cu:43-47 path, no GPU): one 512 x 512 x 512 fp32 layer on the host, timed in
    full -- the positional sparsify, the magnitude prune to 2:4 (STRIP), compress, the dense GEMM and the 2:4 product, all the
    oracle's arithmetic (`port`).  Seeded U(0,1) operands; best of 5 after one warm-up each."""
    import numpy as np
    orc = ge.load_oracle()
    m = n = k = 512
    rng = np.random.default_rng(0x5EED)
    A = rng.uniform(0, 1, m * k).astype(np.float32)
    B = rng.uniform(0, 1, k * n).astype(np.float32)
    C = np.zeros(m * n, dtype=np.float32)

    def best(fn, reps=5):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts) * 1e3

    w, mask = A.copy(), np.ones(m * k, dtype=np.uint64)
    t_pos = best(lambda: orc.sparsify_positional(w, mask, m, k, 0.5))
    Au = A.view(np.uint32)
    t_prune = best(lambda: orc.prune24(Au, m, k, k, orc.STRIP))
    P = orc.prune24(Au, m, k, k, orc.STRIP)
    t_cmp = best(lambda: orc.compress24(P, m, k, k))
    t_gemm = best(lambda: orc.cpu_gemm_f32(A, B, C, m, n, k))
    t_sp = best(lambda: orc.cpu_spmma_f32(A, B, C, m, n, k))
    fl = 2.0 * m * n * k
    return {"m": m, "n": n, "k": k, "b": 1, "dtype": "f32", "kind": "port", "cores": orc.num_threads(),
            "threads": "dense GEMM and 2:4 product: OpenMP over rows on `cores` threads; sparsify / prune / compress: 1 thread",
            "sparsify_positional_ms": t_pos, "prune24_strip_ms": t_prune, "compress24_ms": t_cmp,
            "dense_gemm_ms": t_gemm, "dense_gemm_gfs": fl / t_gemm / 1e6,
            "spmma_2to4_ms": t_sp, "spmma_2to4_eff_gfs": fl / t_sp / 1e6,
            "prune_compress_dense_gemm_ms": t_prune + t_cmp + t_gemm,
            "note": "untuned restatement (naive loops, no cache blocking): a reported baseline, not a target"}


def cpu_baseline(ge, shapes):
    """The oracle's arithmetic ('port': fp32 accumulate, OpenMP over rows) on the host cores, on a
    bounded sample: one batch (b = 1) of every unique (m,n,k) of the table, repeated; both the dense
    product and the 2:4 path (STRIP selection fused with the two kept MACs per strip)."""
    import numpy as np
    orc = ge.load_oracle()
    uniq = sorted(set((m, n, k) for m, n, k, _ in shapes))
    rng = np.random.default_rng(0x5EED)
    reps = 64
    fl = t_dense = t_sparse = 0.0
    for (m, n, k) in uniq:
        r = m
        A = rng.uniform(0, 1, r * k).astype(np.float32)
        B = rng.uniform(0, 1, k * n).astype(np.float32)
        C = np.zeros(r * n, dtype=np.float32)
        orc.cpu_gemm_f32(A, B, C, r, n, k)  # warm
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.cpu_gemm_f32(A, B, C, r, n, k)
        t_dense += time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.cpu_spmma_f32(A, B, C, r, n, k)
        t_sparse += time.perf_counter() - t0
        fl += 2.0 * r * n * k * reps
    return {"value": fl / t_sparse / 1e9, "unit": "GF/s", "cores": orc.num_threads(), "kind": "port",
            "dense_value": fl / t_dense / 1e9,
            "sample": f"untuned oracle port (sm_cpu_spmma_f32 = value, sm_cpu_gemm_f32 = dense_value; naive OpenMP row loops), fp32, b=1 of each of the "
                      f"{len(uniq)} unique shapes x {reps} reps ({fl / 1e9:.1f} dense-equivalent GFLOP, {t_dense + t_sparse:.1f} s of CPU work)"}
