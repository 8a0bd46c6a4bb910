"""datasets/gen_shapes.py (row f-3: shape generator, counterpart of the reference's datasets/get_shapes.py:19-41,66-73)
must reproduce the committed shape tables byte for byte -- they are the data files the reference's sweep reads."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_shapes", os.path.join(ROOT, "datasets", "gen_shapes.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)


@pytest.mark.parametrize("name", sorted(gen.RESNETS))
def test_generated_table_is_the_committed_file(name):
    with open(os.path.join(ROOT, "datasets", name + ".csv"), "rb") as f:
        want = f.read()
    assert gen.to_csv(gen.shapes(name)).encode() == want


def test_shapes_csv_is_resnet50_with_unix_line_ends():
    with open(os.path.join(ROOT, "datasets", "shapes.csv"), "rb") as f:
        want = f.read()
    got = gen.to_csv(gen.shapes("resnet50")).replace("\r\n", "\n").encode()
    assert got.rstrip(b"\n") == want.rstrip(b"\n")


def test_conv_out_formula():
    # get_shapes.py:19-20 on the stem: 224 -> 112 (7x7, stride 2, pad 3); 3x3 stride 2 pad 1: 112 -> 56
    assert gen.conv_out(224, 7, 2, 3) == 112
    assert gen.conv_out(112, 3, 2, 1) == 56
    assert gen.conv_out(56, 1, 1, 0) == 56


def test_other_batch_and_image_sizes():
    rows = gen.shapes("resnet18", batch=8, image=128)
    assert rows[0] == (64 * 64, 64, 147, 8)
    assert all(r[3] == 8 for r in rows) and len(rows) == 17


def test_bench_serves_every_resnet50_layer_with_a_fused_variant():
    """bench.py mirrors the C-side dispatch of sm_spmma_fused_f16 (csrc/spmma_f16_fused.hip: spmma_fused16) to label its kernel
    families; since round 3 every layer of the headline table runs fused: direct (n <= 128), wide (256-column tiles), A-stationary
    (n > 256, k <= 512) or the span form (k % 64 != 0).  Guards the two dispatch tables against drifting apart."""
    import collections
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    shapes = bench.read_shapes(os.path.join(root, "datasets", "resnet50.csv"))
    assert len(shapes) == 49
    fam = collections.Counter(bench.fused_variant(n, k) for (_, n, k, _) in shapes)
    assert fam == {"direct": 17, "wide": 18, "astat": 13, "span": 1}
    # the span form's conditions as bench.py states them hold for the stem layer
    m, n, k, b = next(s for s in shapes if s[2] % 64)
    assert (m, n, k, b) == (12544, 64, 147, 32) and (b * m * k * 2) % 16 == 0 and n % 8 == 0 and n <= 128
    assert 128 * k * 2 + 1152 + (k + 63) // 64 * 64 * 64 * 2 <= 160 * 1024
