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


# published torchvision parameter totals: the convolution weights of the generated layer lists + batch-norm / bias / classifier
# parameters (counted here from the same architecture constants) must add up to them exactly
PUBLISHED_PARAMS = {"mobilenetv2": 3504872, "mobilenetv3_large": 5483032, "mobilenetv3_small": 2542856,
                    "densenet161": 28681000, "densenet201": 20013928}
CONV_LAYERS = {"mobilenetv2": 52, "mobilenetv3_large": 62, "mobilenetv3_small": 52, "densenet161": 160, "densenet201": 200}


def _total_params(name):
    L = gen.network_layers(name)
    tot = gen.conv_parameters(name)
    if name == "mobilenetv2":
        return tot + 2 * sum(l[2] for l in L if l[0] == "conv") + 1280 * 1000 + 1000
    if name in gen.MOBILENETV3:
        last = gen.MOBILENETV3[name][1]
        hidden = 1280 if name.endswith("large") else 1024
        tot += last * hidden + hidden + hidden * 1000 + 1000
        se = False
        for l in L:
            if l[0] == "squeeze":
                se = True
            elif l[0] == "unsqueeze":
                se = False
            elif l[0] == "conv":
                tot += l[2] if se else 2 * l[2]  # squeeze-excitation convolutions carry a bias, the others a batch-norm
        return tot
    growth, stages, c = gen.DENSENETS[name]
    tot += 2 * c
    for si, nl in enumerate(stages):
        for _ in range(nl):
            tot += 2 * c + 2 * 4 * growth
            c += growth
        if si + 1 < len(stages):
            tot += 2 * c
            c //= 2
    return tot + 2 * c + c * 1000 + 1000


@pytest.mark.parametrize("name", sorted(PUBLISHED_PARAMS))
def test_model_zoo_architectures_match_published_parameter_counts(name):
    """datasets/get_shapes.py:87-98 lists MobileNetV2 / V3 and DenseNet-161 / -201 beside the ResNets but commits no table
    for them (and needs torchvision).  The architecture constants here are pinned by torchvision's published totals."""
    assert _total_params(name) == PUBLISHED_PARAMS[name]
    rows = gen.shapes(name)
    assert len(rows) == CONV_LAYERS[name]
    with open(os.path.join(ROOT, "datasets", name + ".csv"), "rb") as f:
        assert gen.to_csv(rows).encode() == f.read()


def test_model_zoo_spatial_sizes_and_grouped_rows():
    r = gen.shapes("mobilenetv2")
    assert r[0] == (112 * 112, 32, 27, 32)                 # stem 3x3 stride 2
    assert r[1] == (112 * 112, 1, 9, 32 * 32)              # depthwise 3x3 on 32 channels: 32 one-column products per image
    assert r[2] == (112 * 112, 16, 32, 32)                 # linear bottleneck 1x1
    assert r[-1] == (7 * 7, 1280, 320, 32)
    # dense-equivalent multiply-adds per image of the whole network: the published ~300 M (convolutions only)
    macs = sum(m * n * k * (b // 32) for m, n, k, b in r)
    assert 295e6 < macs < 305e6
    d = gen.shapes("densenet201")
    assert d[0] == (112 * 112, 64, 147, 32) and d[1] == (56 * 56, 128, 64, 32)   # the max-pool IS applied (unlike the ResNet walk)
    assert d[-1] == (7 * 7, 32, 1152, 32)
    s = gen.shapes("mobilenetv3_small")
    assert s[2] == (1, 8, 16, 32) and s[3] == (1, 16, 8, 32)   # squeeze-excitation convolutions on the 1x1 average
    assert gen.make_divisible(72 // 4) == 24 and gen.make_divisible(16 // 4) == 8 and gen.make_divisible(960 // 4) == 240
    assert gen.shapes("mobilenetv2", image=244)[0][0] == 122 * 122   # the reference's MobileNet cell feeds 244 x 244


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
    # round 4: with the launch's tile count known (grouped launches of the table's instance counts at b = 32) the 256-row big form
    # takes the shapes whose rounds it fills at least as well -- mirror of the rule in spmma_fused16
    cnt = collections.Counter(shapes)
    fam = collections.Counter(bench.fused_variant(n, k, m, b, cnt[(m, n, k, b)]) for (m, n, k, b) in shapes)
    assert fam == {"direct": 17, "span": 1, "big": 13, "wide": 8, "astat": 10}
    assert bench.fused_variant(256, 2304, 784, 32, 6) == "wide" and bench.fused_variant(256, 1024, 784, 32, 5) == "big"
    assert bench.fused_variant(2048, 512, 196, 32, 3) == "big" and bench.fused_variant(1024, 256, 784, 32, 6) == "astat"
    assert bench.fused_variant(256, 2304, 784, 32, 1) == "wide"   # a single instance: 98 big tiles would fill 0.38 of a round
    # the span form's conditions as bench.py states them hold for the stem layer
    m, n, k, b = next(s for s in shapes if s[2] % 64)
    assert (m, n, k, b) == (12544, 64, 147, 32) and (b * m * k * 2) % 16 == 0 and n % 8 == 0 and n <= 128
    assert 128 * k * 2 + 1152 + (k + 63) // 64 * 64 * 64 * 2 <= 160 * 1024
