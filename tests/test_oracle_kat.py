"""CPU tests: the oracle against the hand-derived known answers in tests/golden/kat.json and against
independent pure-Python restatements / properties.  (The reference has no tests or golden vectors of
its own -- SURVEY.md section 4 -- so these KATs are what pins the oracle.)"""
import itertools
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))

DTYPES = [np.float16, np.float32]


def bits(a):
    return a.view({2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


# ---------------------------------------------------------------------------------------------
# positional sparsify (reference include/sparsify.me/sparsify.hxx:32-81)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", KAT["positional"], ids=lambda c: f"{c['m']}x{c['n']}_sf{c['sf']}")
@pytest.mark.parametrize("dtype", [np.float16, np.float32, np.float64])
def test_positional_kat(orc, case, dtype):
    w = np.array(case["weights_in"], dtype=dtype)
    mask = np.full(w.size, 77, dtype=np.uint64)
    orc.sparsify_positional(w, mask, case["m"], case["n"], case["sf"])
    assert w.tolist() == [float(x) for x in case["weights_out"]]
    assert mask.tolist() == case["mask_out"]


def test_positional_counts(orc):
    c = KAT["positional_counts"][0]
    m, n = c["m"], c["n"]
    w = np.arange(1, m * n + 1, dtype=np.float32)
    mask = np.zeros(m * n, dtype=np.uint64)
    orc.sparsify_positional(w, mask, m, n, c["sf"])
    touched = 4 * c["blocks"]
    assert touched + c["untouched_tail"] == m * n
    assert int((w[:touched] == 0).sum()) == 2 * c["blocks"]
    assert np.array_equal(w[touched:], np.arange(touched + 1, m * n + 1, dtype=np.float32))
    assert mask[touched:].min() == 1 and int(mask[:touched].sum()) == 2 * c["blocks"]


def positional_py(w, m, n, blk_m, blk_n, sf):
    """Literal pure-Python transcription of the reference lambda's control flow."""
    w = list(w)
    mask = [1] * (m * n)
    blk = blk_m * blk_n
    nz = int(np.floor(np.float32(blk) * np.float32(sf)))
    for b in range((m // blk_m) * (n // blk_n)):
        g, done = b * blk, 0
        for h in range(blk_m):
            for ww in range(blk_n):
                if done == nz:
                    break
                w[g + h + ww * blk_n] = 0
                mask[g + h + ww * blk_n] = 0
                done += 1
    return w, mask


@settings(max_examples=60, deadline=None)
@given(m=st.integers(0, 9), n=st.integers(0, 9), sf=st.sampled_from([0.0, 0.25, 0.3, 0.5, 0.75, 0.99, 1.0]))
def test_positional_matches_python(orc, m, n, sf):
    w = np.arange(1, m * n + 1, dtype=np.float32)
    mask = np.zeros(m * n, dtype=np.uint64)
    orc.sparsify_positional(w, mask, m, n, sf)
    ew, em = positional_py(np.arange(1, m * n + 1), m, n, 2, 2, sf)
    assert w.tolist() == [float(x) for x in ew] and mask.tolist() == em


def test_positional_other_block_shapes(orc):
    # 1x4 blocks: idx = g + w*4 leaves the block (sparsify.hxx:60); reproduced while in bounds ...
    m, n = 4, 8
    w = np.arange(1, m * n + 1, dtype=np.float32)
    mask = np.zeros(m * n, dtype=np.uint64)
    orc.sparsify_positional(w, mask, m, n, 0.25, blk_m=1, blk_n=4)  # nz = 1 -> offset 0 only
    ew, em = positional_py(np.arange(1, m * n + 1), m, n, 1, 4, 0.25)
    assert w.tolist() == [float(x) for x in ew] and mask.tolist() == em
    # ... and refused where the reference would write outside the buffer
    with pytest.raises(ValueError):
        orc.sparsify_positional(w, mask, m, n, 1.0, blk_m=1, blk_n=4)


# ---------------------------------------------------------------------------------------------
# STRIP / TILE selection
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", KAT["strip"], ids=lambda c: str(c["in"]))
@pytest.mark.parametrize("dtype", DTYPES)
def test_strip_kat(orc, case, dtype):
    a = np.array(case["in"], dtype=dtype)
    out = orc.prune24(bits(a), 1, 4, 4, orc.STRIP).view(dtype)
    assert np.array_equal(out, np.array(case["out"], dtype=dtype))
    blob = orc.compress24(bits(a), 1, 4, 4)
    kc, meta_off, total = orc.compress24_layout(1, 4, a.dtype.itemsize)
    assert (kc, blob.size) == (64, total)
    assert blob[meta_off] & 0xF == case["nibble"]
    vals = blob[: 2 * a.dtype.itemsize].view(dtype)
    assert np.array_equal(vals, a[case["keep"]])
    # padding strips: zero values, nibble 0x4
    assert blob[meta_off] >> 4 == 4 and set(blob[meta_off + 1: meta_off + 8].tolist()) == {0x44}
    assert not blob[2 * a.dtype.itemsize: meta_off].any()


def tile_bruteforce(t):
    best, bestmask = None, None
    pairs = list(itertools.combinations(range(4), 2))
    for choice in itertools.product(pairs, repeat=4):  # lexicographic, same order as the oracle
        cnt = [0] * 4
        for pr in choice:
            for c in pr:
                cnt[c] += 1
        if cnt != [2, 2, 2, 2]:
            continue
        s = sum(abs(float(t[r][c])) for r, pr in enumerate(choice) for c in pr)
        if best is None or s > best:
            best, bestmask = s, choice
    return bestmask


@pytest.mark.parametrize("case", KAT["tile"], ids=["cols01", "allones", "perm"])
@pytest.mark.parametrize("dtype", DTYPES)
def test_tile_kat(orc, case, dtype):
    a = np.array(case["in"], dtype=dtype)
    out = orc.prune24(bits(a).reshape(-1), 4, 4, 4, orc.TILE).view(dtype).reshape(4, 4)
    assert out.tolist() == case["out"]
    choice = tile_bruteforce(case["in"])  # independent confirmation of the hand answer
    exp = [[case["in"][r][c] if c in choice[r] else 0.0 for c in range(4)] for r in range(4)]
    assert exp == case["out"]


@settings(max_examples=200, deadline=None)
@given(data=st.lists(st.integers(-8, 8), min_size=16, max_size=16))
def test_tile_matches_bruteforce_on_small_integers(orc, data):
    # small integers: every candidate score is exact in fp32, so fp64 brute force must agree
    t = np.array(data, dtype=np.float32).reshape(4, 4)
    out = orc.prune24(bits(t).reshape(-1), 4, 4, 4, orc.TILE).view(np.float32).reshape(4, 4)
    choice = tile_bruteforce(t)
    exp = np.array([[t[r][c] if c in choice[r] else 0.0 for c in range(4)] for r in range(4)], dtype=np.float32)
    assert np.array_equal(out, exp)
    kept = out != 0
    assert kept.sum(0).max() <= 2 and kept.sum(1).max() <= 2


def strip_py(v):
    key = [abs(float(x)) for x in v]
    order = sorted(range(4), key=lambda i: (-key[i], i))
    return sorted(order[:2])


@settings(max_examples=200, deadline=None)
@given(m=st.integers(1, 7), k=st.integers(1, 70), seed=st.integers(0, 2**31 - 1), dt=st.sampled_from([0, 1]),
       ties=st.booleans())
def test_strip_properties(orc, m, k, seed, dt, ties):
    dtype = DTYPES[dt]
    rng = np.random.default_rng(seed)
    A = rng.integers(-3, 4, size=(m, k)).astype(dtype) if ties else rng.uniform(-1, 1, size=(m, k)).astype(dtype)
    ld = k + int(rng.integers(0, 3))
    buf = np.zeros(m * ld, dtype=dtype)
    buf.reshape(m, ld)[:, :k] = A
    P = orc.prune24(bits(buf), m, k, ld, orc.STRIP).view(dtype).reshape(m, ld)[:, :k]
    for i in range(m):
        for c in range(0, k, 4):
            v = list(A[i, c:c + 4]) + [0.0] * (4 - len(A[i, c:c + 4]))
            keep = strip_py(v)
            for t in range(min(4, k - c)):
                want = A[i, c + t] if t in keep else 0
                assert P[i, c + t] == want and (t in keep or not np.signbit(P[i, c + t]))
    assert orc.prune24_check(bits(np.ascontiguousarray(P).reshape(-1)), m, k, k) == 0
    # idempotent, and compress of the pruned matrix == compress of the original (same selection)
    Pc = np.ascontiguousarray(P).reshape(-1)
    assert np.array_equal(orc.prune24(bits(Pc), m, k, k, orc.STRIP), bits(Pc))
    blobP = orc.compress24(bits(Pc), m, k, k)
    blobA = orc.compress24(bits(buf), m, k, ld)
    assert np.array_equal(blobP, blobA)
    # round trip
    D = orc.decompress24(blobP, m, k, k, bits(Pc).dtype)
    assert np.array_equal(D, bits(Pc))


def test_strip_special_values(orc):
    inf, nan = np.inf, np.nan
    a = np.array([1.0, nan, inf, 2.0, -0.0, 0.0, -0.0, 0.0, nan, nan, nan, 1.0], dtype=np.float16)
    out = orc.prune24(bits(a), 1, 12, 12, orc.STRIP).view(np.float16)
    assert np.isnan(out[1]) and out[2] == inf and out[0] == 0 and out[3] == 0      # NaN > inf > finite
    assert bits(out)[4:8].tolist() == [0x8000, 0x0000, 0, 0]                       # ties keep index 0,1; -0 kept as is
    assert np.isnan(out[8]) and np.isnan(out[9]) and out[10] == 0 and out[11] == 0  # NaN ties -> lower index
    assert orc.prune24_check(bits(a), 1, 12, 12) == 1   # first strip has 4 non-zeros
    assert orc.prune24_check(bits(out), 1, 12, 12) == 0


def test_prune_check_rule(orc):
    a = np.array([1, 0, 2, 0, 0, 0, 0, 0], dtype=np.float32)
    assert orc.prune24_check(bits(a), 2, 4, 4) == 0
    a[1] = 3
    assert orc.prune24_check(bits(a), 2, 4, 4) == 1
    # ragged tail: k = 3, three non-zeros in the only (partial) strip
    assert orc.prune24_check(bits(np.array([1, 2, 3], dtype=np.float32)), 1, 3, 3) == 1
    assert orc.prune24_check(bits(np.array([1, 0, 3], dtype=np.float32)), 1, 3, 3) == 0
    u = np.random.default_rng(0).uniform(0, 1, 64 * 64).astype(np.float16)
    assert orc.prune24_check(bits(u), 64, 64, 64) == 1


@pytest.mark.parametrize("alg", [0, 1])
@pytest.mark.parametrize("dtype", DTYPES)
def test_prune_in_place_and_ragged(orc, alg, dtype):
    rng = np.random.default_rng(7)
    m, k, ld = 10, 147, 150
    buf = rng.uniform(-1, 1, m * ld).astype(dtype)
    ref = orc.prune24(bits(buf), m, k, ld, alg)
    # padding columns [k, ld) are never touched
    assert np.array_equal(ref.reshape(m, ld)[:, k:], bits(buf).reshape(m, ld)[:, k:])
    P = ref.view(dtype).reshape(m, ld)[:, :k]
    nz = (P != 0)
    for c in range(0, k, 4):
        assert nz[:, c:c + 4].sum(1).max() <= 2
    if alg == 0:
        for r in range(0, m, 4):
            for c in range(0, k, 4):
                assert nz[r:r + 4, c:c + 4].sum(0).max() <= 2
    assert orc.prune24_check(ref, m, k, ld) == 0


# ---------------------------------------------------------------------------------------------
# compressed layout, batches, matmul restatements
# ---------------------------------------------------------------------------------------------
def test_compress_layout_and_batches(orc):
    m, k, batch = 5, 147, 3
    kc, meta_off, total = orc.compress24_layout(m, k, 2, batch)
    # values 15 rows x 96 halves = 2880 B -> 3072; metadata 15 x 24 = 360 B -> 512
    assert kc == 192 and meta_off == 3072 and total == 3072 + 512
    assert meta_off % 256 == 0 and meta_off >= batch * m * (kc // 2) * 2
    rng = np.random.default_rng(3)
    A = rng.uniform(-1, 1, batch * m * k).astype(np.float16)
    blob = orc.compress24(bits(A), m, k, k, batch)
    # batch b of the blob == blob of batch b alone (rows are independent)
    for b in range(batch):
        one = orc.compress24(bits(A[b * m * k:(b + 1) * m * k]), m, k, k, 1)
        kc1, mo1, _ = orc.compress24_layout(m, k, 2, 1)
        # both sections are stage-major: plane s holds, for ALL rows, the 32 kept halves (64 B) / the 8
        # metadata bytes of dense k 64s..64s+63
        M = batch * m
        for s in range(kc // 64):
            assert np.array_equal(blob[(s * M + b * m) * 64: (s * M + (b + 1) * m) * 64],
                                  one[s * m * 64: (s + 1) * m * 64])
            assert np.array_equal(blob[meta_off + (s * M + b * m) * 8: meta_off + (s * M + (b + 1) * m) * 8],
                                  one[mo1 + s * m * 8: mo1 + (s + 1) * m * 8])
    D = orc.decompress24(blob, m, k, k, np.uint16, batch)
    assert np.array_equal(D, orc.prune24(bits(A), batch * m, k, k, orc.STRIP))


@pytest.mark.parametrize("dtype", DTYPES)
def test_spmma_equals_dense_gemm_of_pruned(orc, dtype):
    rng = np.random.default_rng(11)
    m, n, k, batch = 9, 13, 50, 2
    A = rng.integers(-4, 5, batch * m * k).astype(dtype)   # integers: every sum exact in every precision
    B = rng.integers(-4, 5, k * n).astype(dtype)
    P = orc.prune24(bits(A), batch * m, k, k, orc.STRIP)
    blob = orc.compress24(P, m, k, k, batch)
    C = np.zeros(batch * m * n, dtype=dtype)
    orc.spmma(blob, bits(B), bits(C), m, n, k, batch)
    want = (P.view(dtype).astype(np.float64).reshape(batch * m, k) @ B.astype(np.float64).reshape(k, n))
    assert np.array_equal(C.astype(np.float64).reshape(batch * m, n), want)
    C2 = np.zeros_like(C)
    orc.gemm_rowmajor(P, bits(B), bits(C2), m, n, k, batch=batch)
    assert np.array_equal(C, C2)
    # alpha / beta
    C3 = np.ones_like(C)
    orc.spmma(blob, bits(B), bits(C3), m, n, k, batch, alpha=2.0, beta=-1.0)
    assert np.array_equal(C3.astype(np.float64).reshape(batch * m, n), 2 * want - 1)


@pytest.mark.parametrize("dtype", [np.float16, np.float32, np.float64])
def test_gemm_batched_column_major_identity_asymmetric(orc, dtype):
    # A = I (padded) with an ASYMMETRIC B: a row/col swap anywhere shows up
    m, n, k, batch = 6, 5, 7, 2
    A = np.zeros((k, m), dtype=dtype)              # column-major m x k, lda = m  ==  row-major [k][m]
    for i in range(min(m, k)):
        A[i, i] = 1
    Bs, Cs, wants = [], [], []
    for b in range(batch):
        Bm = np.array([[10 * l + j + 100 * b for j in range(n)] for l in range(k)], dtype=np.float64)  # k x n
        Bs.append(np.ascontiguousarray(Bm.T).astype(dtype).reshape(-1))   # column-major k x n, ldb = k
        Cs.append(np.zeros(m * n, dtype=dtype))
        wants.append((np.eye(m, k) @ Bm))
    As = [A.reshape(-1)] * batch
    orc.gemm_batched(As, Bs, Cs, m, n, k)
    for b in range(batch):
        got = Cs[b].reshape(n, m).T.astype(np.float64)   # column-major m x n, ldc = m
        assert np.array_equal(got, wants[b])


def test_fp16_conversions_roundtrip(orc):
    # f2h/h2f inside the oracle: every finite half survives widen -> narrow, and narrowing rounds to nearest even
    import ctypes
    allh = np.arange(0, 1 << 16, dtype=np.uint16)
    f = np.zeros(allh.size, dtype=np.float32)
    orc.lib().sm_widen_f16(allh.ctypes.data_as(ctypes.c_void_p), f.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(allh.size))
    npf = allh.view(np.float16).astype(np.float32)
    assert np.array_equal(f.view(np.uint32)[~np.isnan(npf)], npf.view(np.uint32)[~np.isnan(npf)])
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-70000, 70000, 20000), rng.uniform(-1e-4, 1e-4, 20000),
                        np.array([65504.0, 65519.9, 65520.0, 1e-8, 2.98e-8, 2.981e-8, 0.0, -0.0])]).astype(np.float32)
    h = np.zeros(x.size, dtype=np.uint16)
    orc.lib().sm_narrow_f16(x.ctypes.data_as(ctypes.c_void_p), h.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(x.size))
    with np.errstate(over="ignore"):
        assert np.array_equal(h, x.astype(np.float16).view(np.uint16))


def test_bf16_conversions_and_rules(orc):
    """bfloat16 in the oracle: conversions agree with torch's CPU bfloat16 (an independent implementation of round to
    nearest even), the STRIP rule is the fp16 function of the bits, the TILE rule keeps 2 per row and column with a
    maximal kept magnitude, and the bf16 matmul refs equal an fp64 matmul of the widened operands rounded once."""
    import ctypes
    import torch
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-3e38, 3e38, 5000), rng.uniform(-1, 1, 20000), rng.uniform(-1e-38, 1e-38, 2000),
                        np.array([1.0, 1.00390625, 1.005859375, 1.001953125, 3.3895313892515355e38, 3.4e38, 0.0, -0.0, np.inf, -np.inf])]).astype(np.float32)
    h = np.zeros(x.size, dtype=np.uint16)
    orc.lib().sm_narrow_bf16(x.ctypes.data_as(ctypes.c_void_p), h.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(x.size))
    want = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(h, want)
    allb = np.arange(0, 1 << 16, dtype=np.uint16)
    f = np.zeros(allb.size, dtype=np.float32)
    orc.lib().sm_widen_bf16(allb.ctypes.data_as(ctypes.c_void_p), f.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(allb.size))
    assert np.array_equal(f.view(np.uint32), allb.astype(np.uint32) << 16)
    # STRIP: same bits in, same bits out as the fp16 rule
    m, k = 37, 52
    Ab = rng.integers(0, 1 << 16, m * k).astype(np.uint16)
    assert np.array_equal(orc.prune24(Ab, m, k, k, orc.STRIP, bf16=True), orc.prune24(Ab, m, k, k, orc.STRIP))
    # TILE on bf16 magnitudes: valid pattern, never worse than STRIP's kept sum when STRIP happens to be column-valid
    A = torch.from_numpy(rng.uniform(-4, 4, (16, 16)).astype(np.float32)).to(torch.bfloat16)
    Abits = A.view(torch.int16).numpy().view(np.uint16).reshape(-1).copy()
    P = orc.prune24(Abits, 16, 16, 16, orc.TILE, bf16=True).reshape(16, 16)
    nz = P != 0
    mag = np.abs(A.to(torch.float32).numpy().astype(np.float64))
    for r0 in range(0, 16, 4):
        for c0 in range(0, 16, 4):
            t = nz[r0:r0 + 4, c0:c0 + 4]
            assert (t.sum(0) == 2).all() and (t.sum(1) == 2).all()
            # brute force over the 90 patterns
            import itertools
            pairs = list(itertools.combinations(range(4), 2))
            best = -1.0
            for combo in itertools.product(pairs, repeat=4):
                cols = np.zeros(4, int)
                for pr in combo:
                    cols[list(pr)] += 1
                if (cols == 2).all():
                    best = max(best, sum(mag[r0 + r, c0 + c] for r, pr in enumerate(combo) for c in pr))
            assert abs((mag[r0:r0 + 4, c0:c0 + 4] * t).sum() - best) <= 1e-6 * best
    # matmul refs
    m, n, k = 9, 7, 24
    Af = torch.from_numpy(rng.uniform(-1, 1, (m, k)).astype(np.float32)).to(torch.bfloat16)
    Bf = torch.from_numpy(rng.uniform(-1, 1, (k, n)).astype(np.float32)).to(torch.bfloat16)
    ab = Af.view(torch.int16).numpy().view(np.uint16).reshape(-1).copy()
    bb = Bf.view(torch.int16).numpy().view(np.uint16).reshape(-1).copy()
    C = np.zeros(m * n, dtype=np.uint16)
    orc.gemm_rowmajor(ab, bb, C, m, n, k, bf16=True)
    want = torch.from_numpy((Af.to(torch.float64).numpy() @ Bf.to(torch.float64).numpy()).astype(np.float32)).to(torch.bfloat16)
    got = torch.from_numpy(C.view(np.int16)).view(torch.bfloat16)
    assert (got.to(torch.float32) - want.reshape(-1).to(torch.float32)).abs().max() <= 2.0 ** -7 * max(1.0, float(want.to(torch.float32).abs().max()))  # at most 1 ulp (double rounding)
    blob = orc.compress24(ab, m, k, k)
    pr = orc.prune24(ab, m, k, k, orc.STRIP, bf16=True)
    C1, C2 = np.zeros(m * n, dtype=np.uint16), np.zeros(m * n, dtype=np.uint16)
    orc.spmma(blob, bb, C1, m, n, k, bf16=True)
    orc.gemm_rowmajor(pr, bb, C2, m, n, k, bf16=True)
    assert np.array_equal(C1, C2)


@pytest.mark.parametrize("cfg", [(2, 3, 9, 11, 3, 3, 1, 1, 1), (1, 4, 12, 12, 7, 7, 2, 3, 1), (2, 5, 8, 8, 1, 1, 1, 0, 1),
                                 (1, 2, 10, 9, 3, 2, 2, 0, 2), (1, 3, 224, 224, 7, 7, 2, 3, 1)])
def test_im2col_restatement_is_unfold_transposed(orc, cfg):
    """The oracle's im2col against torch's CPU unfold (what datasets/get_shapes.py:30-35 calls), transposed to rows =
    output pixels; and the (m, k) it yields are the shape tables' (get_shapes.py:66-73)."""
    import torch
    N, C, H, W, kh, kw, s, p, d = cfg
    rng = np.random.default_rng(sum(cfg))
    X = rng.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    want = torch.nn.functional.unfold(torch.from_numpy(X), (kh, kw), dilation=d, padding=p, stride=s)   # N x K x L
    want = want.transpose(1, 2).contiguous().numpy()                                                       # N x L x K
    got = orc.im2col(X.reshape(-1), N, C, H, W, kh, kw, s, p, d).reshape(want.shape)
    assert np.array_equal(got, want)
    assert want.shape[1] == orc.conv_out_size(H, kh, s, p, d) * orc.conv_out_size(W, kw, s, p, d)
    if cfg == (1, 3, 224, 224, 7, 7, 2, 3, 1):
        assert want.shape[1:] == (12544, 147)     # first row of datasets/resnet*.csv
    # 16-bit elements move as opaque values too
    Xh = X.astype(np.float16)
    got16 = orc.im2col(Xh.reshape(-1).view(np.uint16), N, C, H, W, kh, kw, s, p, d)
    assert np.array_equal(got16.view(np.float16).astype(np.float32).reshape(want.shape), want.astype(np.float16).astype(np.float32))


def test_int8_restatements(orc):
    """int8 forms in the oracle: STRIP on |x| of signed bytes (|-128| = 128 beats 127, ties keep the lower index),
    compress/decompress round trip, and the exact int32 product against numpy on the pruned operand ([n][k] B)."""
    A = np.array([[-128, 127, 3, -3], [5, -5, 5, 5], [0, 0, 0, 0], [1, -2, 2, -1]], dtype=np.int8).reshape(-1)
    P = orc.prune24(A.view(np.uint8), 4, 4, 4, orc.STRIP).view(np.int8).reshape(4, 4)
    assert P.tolist() == [[-128, 127, 0, 0], [5, -5, 0, 0], [0, 0, 0, 0], [0, -2, 2, 0]]
    assert orc.prune24_check(P.reshape(-1).view(np.uint8), 4, 4, 4) == 0
    assert orc.prune24_check(A.view(np.uint8), 4, 4, 4) == 1
    rng = np.random.default_rng(21)
    m, n, k, batch = 10, 7, 128, 2
    A = rng.integers(-128, 128, batch * m * k).astype(np.int8)
    B = rng.integers(-128, 128, n * k).astype(np.int8)                      # [n][k]
    blob = orc.compress24(A.view(np.uint8), m, k, k, batch)
    assert blob.size == orc.compress24_size(m, k, 1, batch)
    P = orc.prune24(A.view(np.uint8), batch * m, k, k, orc.STRIP)
    assert np.array_equal(orc.decompress24(blob, m, k, k, np.uint8, batch), P)
    assert np.array_equal(orc.compress24(P, m, k, k, batch), blob)
    C = np.full(batch * m * n, 5, dtype=np.int32)
    orc.spmma_i8(blob, B, C, m, n, k, batch, 0)
    want = P.view(np.int8).reshape(batch * m, k).astype(np.int64) @ B.reshape(n, k).astype(np.int64).T
    assert np.array_equal(C.reshape(batch * m, n), want)
    orc.spmma_i8(blob, B, C, m, n, k, batch, 0, accumulate=True)
    assert np.array_equal(C.reshape(batch * m, n), 2 * want)


def test_bell_and_coo_restatements(orc):
    rng = np.random.default_rng(2)
    rows, cols, bs, n = 8, 12, 2, 5
    ell_cols = cols // 2
    bcols = ell_cols // bs
    ci = np.stack([np.sort(rng.choice(cols // bs, bcols, replace=False)) for _ in range(rows // bs)]).astype(np.uint64)
    vals = rng.integers(-3, 4, (rows, ell_cols)).astype(np.float32)
    Bm = rng.integers(-3, 4, (cols, n)).astype(np.float32)
    dense = np.zeros((rows, cols))
    for i in range(rows):
        for e in range(bcols):
            c0 = int(ci[i // bs, e]) * bs
            dense[i, c0:c0 + bs] = vals[i, e * bs:(e + 1) * bs]
    C = np.zeros(rows * n, dtype=np.float32)
    orc.spmm_bell(vals.reshape(-1), ci.reshape(-1), rows, cols, bs, ell_cols, np.ascontiguousarray(Bm.T).reshape(-1), C, n)
    assert np.array_equal(C.reshape(n, rows).T, dense @ Bm)
    # COO with a duplicate coordinate, two batches
    r = np.array([0, 2, 2, 7, 2], dtype=np.int32)
    c = np.array([1, 3, 3, 11, 0], dtype=np.int32)
    v = np.array([1, 2, 3, 4, 5], dtype=np.float32)
    d2 = np.zeros((rows, cols))
    for rr, cc, vv in zip(r, c, v):
        d2[rr, cc] += vv
    B2 = rng.integers(-3, 4, (2, cols, n)).astype(np.float32)
    Bcm = np.concatenate([np.ascontiguousarray(B2[b].T).reshape(-1) for b in range(2)])
    C2 = np.zeros(2 * rows * n, dtype=np.float32)
    orc.spmm_coo(rows, cols, 5, n, 2, r, c, v, Bcm, C2)
    for b in range(2):
        assert np.array_equal(C2[b * rows * n:(b + 1) * rows * n].reshape(n, rows).T, d2 @ B2[b])


def test_tile_rule_two_level_statement_vs_exhaustive(orc):
    """The TILE rule is frozen in its two-level statement (oracle/sm_oracle.c: tile_select; what the kernels compute
    from csrc/tile_rule.inc).  It always reaches the same maximal fp32 score as round 1's exhaustive statement (first
    strictly greatest of the 90 totals), picks the same pattern whenever the sums are exact (integers; fp16 data of
    ordinary dynamic range), and can differ from it only through fp32 rounding collisions on data spanning many
    binades -- one such tile is pinned here."""
    rng = np.random.default_rng(2)
    differing = 0
    for trial in range(6000):
        kind = trial % 3
        if kind == 0:
            mag = rng.integers(0, 4, 16).astype(np.float32)                      # heavy ties, exact sums
        elif kind == 1:
            mag = np.abs(rng.uniform(-1, 1, 16)).astype(np.float16).astype(np.float32)
        else:
            mag = (2.0 ** rng.integers(-20, 20, 16) * rng.uniform(1, 2, 16)).astype(np.float32)
        ma, mb, sa, sb = orc.tile_select_both(mag)
        assert sa == sb, "the two statements must reach the same maximal score"
        for m_ in (ma, mb):   # two per row, two per column
            bitsm = [(m_ >> i) & 1 for i in range(16)]
            assert all(sum(bitsm[4 * r:4 * r + 4]) == 2 for r in range(4)) and all(sum(bitsm[c::4]) == 2 for c in range(4))
        if kind < 2:
            assert ma == mb, (mag, hex(ma), hex(mb))
        differing += ma != mb
    # a rounding collision (found by search): both patterns total 19200.771484375 in fp32
    mag = np.array([0.5947265625, 61.4375, 3690.0, 0.111328125, 15016.0, 0.00022077560424804688, 0.0033855438232421875, 6.7578125,
                    0.007778167724609375, 0.019287109375, 0.0550537109375, 4.5234375, 0.00018393993377685547, 0.01409912109375,
                    0.04827880859375, 426.5], dtype=np.float32)
    ma, mb, sa, sb = orc.tile_select_both(mag)
    assert (ma, mb) == (0xa596, 0xc396) and sa == sb == np.float32(19200.771484375)
