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
