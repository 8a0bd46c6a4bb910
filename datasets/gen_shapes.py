#!/usr/bin/env python3
"""Shape-table generator: the (m, n, k, b) matmul shapes of a network's convolutions seen as im2col products
C[m x n] = A[m x k] . B[k x n]  with  m = out_h * out_w, n = out_channels, k = in_channels * kh * kw, b = batch.

Counterpart of the reference's datasets/get_shapes.py:19-41,66-73 (which needs torchvision to enumerate the
layers; absent here and not needed: the networks of its model zoo, get_shapes.py:87-98, are small tables of
architecture constants).

ResNets (get_shapes.py:22-41, `print_resnet_conv_shapes`): the reference's walk is reproduced including its two
peculiarities, so that the output is byte-identical to the committed datasets/resnet*.csv:
  * only convolutions that are not in a `downsample` branch are visited (get_shapes.py:27);
  * the spatial size is chained from convolution to convolution only -- the stem's max-pool is never applied
    (get_shapes.py:28-40), so the first stage runs at 112 x 112 = 12544 rows, not 56 x 56.

MobileNetV2 / V3-small / V3-large and DenseNet-161 / -201 (get_shapes.py:43-64, `print_mobilenet_shapes`, and the
model zoo): the reference commits no table for them and its walk cannot produce one -- it multiplies
`weight.view(out, -1)` with the unfolded input (`:57`), which has no meaning for a grouped convolution, and it
applies container modules AND their children.  What is generated here is the evident intent: every nn.Conv2d of the
torchvision definition in module order, spatial sizes as the real forward pass sees them (pooling layers applied,
squeeze-excitation convolutions at 1 x 1).  A grouped convolution (depthwise: groups = channels) is `groups`
independent products per image, so its row is  m = out_h * out_w, n = out_channels / groups,
k = (in_channels / groups) * kh * kw, b = batch * groups  -- still the four columns every driver reads.
The reference's MobileNet cell feeds 244 x 244 images (get_shapes.py:44, probably a typo for 224); the default
here is 224 for every network, `--image 244` reproduces that size.

usage: gen_shapes.py [--batch 32] [--image 224] [--out DIR] [name ...]
Without --out the table is printed; tests/test_shapes.py checks the ResNets against the committed files and the
other tables against hand-checked layer counts / parameter counts."""
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
# growth rate, blocks per stage, stem channels (bn_size = 4)
DENSENETS = {
    "densenet161": (48, (6, 12, 36, 24), 96),
    "densenet201": (32, (6, 12, 48, 32), 64),
}
# MobileNetV2: expansion t, output channels c, repeats n, stride s of each stage
MOBILENETV2 = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))
# MobileNetV3: input channels, kernel, expanded channels, output channels, squeeze-excitation, stride of each block; last 1x1 width
MOBILENETV3 = {
    "mobilenetv3_large": (((16, 3, 16, 16, False, 1), (16, 3, 64, 24, False, 2), (24, 3, 72, 24, False, 1), (24, 5, 72, 40, True, 2),
                           (40, 5, 120, 40, True, 1), (40, 5, 120, 40, True, 1), (40, 3, 240, 80, False, 2), (80, 3, 200, 80, False, 1),
                           (80, 3, 184, 80, False, 1), (80, 3, 184, 80, False, 1), (80, 3, 480, 112, True, 1), (112, 3, 672, 112, True, 1),
                           (112, 5, 672, 160, True, 2), (160, 5, 960, 160, True, 1), (160, 5, 960, 160, True, 1)), 960),
    "mobilenetv3_small": (((16, 3, 16, 16, True, 2), (16, 3, 72, 24, False, 2), (24, 3, 88, 24, False, 1), (24, 5, 96, 40, True, 2),
                           (40, 5, 240, 40, True, 1), (40, 5, 240, 40, True, 1), (40, 5, 120, 48, True, 1), (48, 5, 144, 48, True, 1),
                           (48, 5, 288, 96, True, 2), (96, 5, 576, 96, True, 1), (96, 5, 576, 96, True, 1)), 576),
}
NETWORKS = list(RESNETS) + ["mobilenetv2"] + list(MOBILENETV3) + list(DENSENETS)


def conv_out(size, kernel, stride, padding, dilation=1):
    """floor((size + 2p - d(k-1) - 1) / s + 1)   (get_shapes.py:19-20)"""
    return (size + 2 * padding - dilation * (kernel - 1) - 1) // stride + 1


def make_divisible(v, divisor=8):
    """torchvision's channel rounding (squeeze width of a squeeze-excitation block = make_divisible(expanded / 4))"""
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    return new_v + divisor if new_v < 0.9 * v else new_v


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


# ---- the other networks: a list of layer records in module order --------------------------------------------------
#   ("conv", cin, cout, kernel, stride, padding, groups)   an nn.Conv2d
#   ("pool", kernel, stride, padding)                      max / average pooling between convolutions
#   ("squeeze",) / ("unsqueeze",)                          the squeeze-excitation branch works on the 1 x 1 global average


def mobilenetv2_layers():
    L = [("conv", 3, 32, 3, 2, 1, 1)]
    inp = 32
    for t, c, n, s in MOBILENETV2:
        for i in range(n):
            hidden = inp * t
            if t != 1:
                L.append(("conv", inp, hidden, 1, 1, 0, 1))
            L.append(("conv", hidden, hidden, 3, s if i == 0 else 1, 1, hidden))  # depthwise
            L.append(("conv", hidden, c, 1, 1, 0, 1))
            inp = c
    L.append(("conv", inp, 1280, 1, 1, 0, 1))
    return L


def mobilenetv3_layers(name):
    blocks, last = MOBILENETV3[name]
    L = [("conv", 3, 16, 3, 2, 1, 1)]
    for cin, ksz, exp, cout, se, stride in blocks:
        if exp != cin:
            L.append(("conv", cin, exp, 1, 1, 0, 1))
        L.append(("conv", exp, exp, ksz, stride, (ksz - 1) // 2, exp))  # depthwise
        if se:
            sq = make_divisible(exp // 4, 8)
            L += [("squeeze",), ("conv", exp, sq, 1, 1, 0, 1), ("conv", sq, exp, 1, 1, 0, 1), ("unsqueeze",)]
        L.append(("conv", exp, cout, 1, 1, 0, 1))
    L.append(("conv", blocks[-1][3], last, 1, 1, 0, 1))
    return L


def densenet_layers(name):
    growth, stages, c = DENSENETS[name]
    L = [("conv", 3, c, 7, 2, 3, 1), ("pool", 3, 2, 1)]
    for si, nlayers in enumerate(stages):
        for _ in range(nlayers):
            L.append(("conv", c, 4 * growth, 1, 1, 0, 1))
            L.append(("conv", 4 * growth, growth, 3, 1, 1, 1))
            c += growth
        if si + 1 < len(stages):  # transition: 1x1 to half the channels, 2x2 average pool
            L.append(("conv", c, c // 2, 1, 1, 0, 1))
            L.append(("pool", 2, 2, 0))
            c //= 2
    return L


def network_layers(name):
    if name == "mobilenetv2":
        return mobilenetv2_layers()
    if name in MOBILENETV3:
        return mobilenetv3_layers(name)
    if name in DENSENETS:
        return densenet_layers(name)
    raise KeyError(name)


def conv_parameters(name):
    """weights of the network's convolutions (no bias, no batch-norm, no classifier): a cross-check against the published sizes"""
    if name in RESNETS:
        return sum(cin * cout * k * k for cin, cout, k, _, _ in resnet_convs(name))
    return sum(l[1] // l[6] * l[2] * l[3] * l[3] for l in network_layers(name) if l[0] == "conv")


def shapes(name, batch=32, image=224):
    rows, h, w = [], image, image
    if name in RESNETS:
        for cin, cout, ksz, stride, pad in resnet_convs(name):
            h, w = conv_out(h, ksz, stride, pad), conv_out(w, ksz, stride, pad)
            rows.append((h * w, cout, cin * ksz * ksz, batch))
        return rows
    saved = None
    for l in network_layers(name):
        if l[0] == "conv":
            _, cin, cout, ksz, stride, pad, groups = l
            h, w = conv_out(h, ksz, stride, pad), conv_out(w, ksz, stride, pad)
            rows.append((h * w, cout // groups, cin // groups * ksz * ksz, batch * groups))
        elif l[0] == "pool":
            h, w = conv_out(h, l[1], l[2], l[3]), conv_out(w, l[1], l[2], l[3])
        elif l[0] == "squeeze":
            saved, h, w = (h, w), 1, 1
        elif l[0] == "unsqueeze":
            h, w = saved
    return rows


def to_csv(rows):
    # csv.writer's default dialect ends lines with \r\n (get_shapes.py:68-73)
    return "m,n,k,b\r\n" + "".join("%d,%d,%d,%d\r\n" % r for r in rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=NETWORKS)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--image", type=int, default=224)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    for name in a.names:
        if name not in NETWORKS:
            sys.exit("unknown network %r (have: %s)" % (name, ", ".join(NETWORKS)))
        text = to_csv(shapes(name, a.batch, a.image))
        if a.out:
            os.makedirs(a.out, exist_ok=True)
            with open(os.path.join(a.out, name + ".csv"), "w", newline="") as f:
                f.write(text)
        else:
            sys.stdout.write("# %s\n%s" % (name, text))


if __name__ == "__main__":
    main()
