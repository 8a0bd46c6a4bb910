#!/usr/bin/env python3
"""Shape-table generator: the (m, n, k, b) matmul shapes of a ResNet's convolutions seen as im2col products
C[m x n] = A[m x k] . B[k x n]  with  m = out_h * out_w, n = out_channels, k = in_channels * kh * kw, b = batch.

Counterpart of the reference's datasets/get_shapes.py:19-41,66-73 (which needs torchvision to enumerate the
layers; absent here and not needed: the ResNet family is five small tables).  It reproduces the reference's
walk, including its two peculiarities, so that the output is byte-identical to the committed datasets/*.csv:
  * only convolutions that are not in a `downsample` branch are visited (get_shapes.py:27);
  * the spatial size is chained from convolution to convolution only -- the stem's max-pool is never applied
    (get_shapes.py:28-40), so the first stage runs at 112 x 112 = 12544 rows, not 56 x 56.

usage: gen_shapes.py [--batch 32] [--image 224] [--out DIR] [name ...]     (names: resnet18 34 50 101 152)
Without --out the table is printed; tests/test_shapes.py checks all five against the committed files."""
import argparse
import os
import sys

# (block kind, blocks per stage)
RESNETS = {
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
    "resnet101": ("bottleneck", (3, 4, 23, 3)),
    "resnet152": ("bottleneck", (3, 8, 36, 3)),
}


def conv_out(size, kernel, stride, padding, dilation=1):
    """floor((size + 2p - d(k-1) - 1) / s + 1)   (get_shapes.py:19-20)"""
    return (size + 2 * padding - dilation * (kernel - 1) - 1) // stride + 1


def resnet_convs(name):
    """(in_channels, out_channels, kernel, stride, padding) of every non-downsample convolution, in module order."""
    kind, stages = RESNETS[name]
    convs = [(3, 64, 7, 2, 3)]  # stem
    inplanes = 64
    for si, nblocks in enumerate(stages):
        planes = 64 << si
        for bi in range(nblocks):
            stride = 2 if (si > 0 and bi == 0) else 1
            if kind == "basic":
                convs.append((inplanes, planes, 3, stride, 1))
                convs.append((planes, planes, 3, 1, 1))
                inplanes = planes
            else:  # bottleneck, stride on the 3x3 (the "v1.5" placement)
                convs.append((inplanes, planes, 1, 1, 0))
                convs.append((planes, planes, 3, stride, 1))
                convs.append((planes, planes * 4, 1, 1, 0))
                inplanes = planes * 4
    return convs


def shapes(name, batch=32, image=224):
    rows, h, w = [], image, image
    for cin, cout, ksz, stride, pad in resnet_convs(name):
        h, w = conv_out(h, ksz, stride, pad), conv_out(w, ksz, stride, pad)
        rows.append((h * w, cout, cin * ksz * ksz, batch))
    return rows


def to_csv(rows):
    # csv.writer's default dialect ends lines with \r\n (get_shapes.py:68-73)
    return "m,n,k,b\r\n" + "".join("%d,%d,%d,%d\r\n" % r for r in rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=list(RESNETS))
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--image", type=int, default=224)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    for name in a.names:
        if name not in RESNETS:
            sys.exit("unknown network %r (have: %s)" % (name, ", ".join(RESNETS)))
        text = to_csv(shapes(name, a.batch, a.image))
        if a.out:
            os.makedirs(a.out, exist_ok=True)
            with open(os.path.join(a.out, name + ".csv"), "w", newline="") as f:
                f.write(text)
        else:
            sys.stdout.write("# %s\n%s" % (name, text))


if __name__ == "__main__":
    main()
