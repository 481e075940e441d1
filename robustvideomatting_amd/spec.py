"""Architecture tables for the RVM per-frame inference path.

Single source of truth (on the Python side) for layer shapes and for the
``state_dict`` key layout the drop-in boundary has to honour.  The native
engine (csrc/engine.cpp) carries the same tables in C++; tests/test_c_abi_gpu.py
checks that the two agree through ``rvm_param_count`` / ``rvm_param_name``, and
tests/test_encoder_tables.py holds both against the reference's own literals
(tests/golden/ref_encoder_tables.json).

Reference for the numbers:
  * MobileNetV3-Large rows  - /root/reference/model/mobilenetv3.py:9-27
  * ResNet50 layout         - /root/reference/model/resnet.py:7-11
  * LR-ASPP / decoder sizes - /root/reference/model/model.py:23-33
  * refiner                 - /root/reference/model/deep_guided_filter.py:10-22
"""
from collections import OrderedDict

# (in, kernel, expanded, out, use_se, act, stride, dilation)
MBV3_ROWS = [
    (16, 3, 16, 16, False, "RE", 1, 1),
    (16, 3, 64, 24, False, "RE", 2, 1),
    (24, 3, 72, 24, False, "RE", 1, 1),
    (24, 5, 72, 40, True, "RE", 2, 1),
    (40, 5, 120, 40, True, "RE", 1, 1),
    (40, 5, 120, 40, True, "RE", 1, 1),
    (40, 3, 240, 80, False, "HS", 2, 1),
    (80, 3, 200, 80, False, "HS", 1, 1),
    (80, 3, 184, 80, False, "HS", 1, 1),
    (80, 3, 184, 80, False, "HS", 1, 1),
    (80, 3, 480, 112, True, "HS", 1, 1),
    (112, 3, 672, 112, True, "HS", 1, 1),
    (112, 5, 672, 160, True, "HS", 2, 2),
    (160, 5, 960, 160, True, "HS", 1, 2),
    (160, 5, 960, 160, True, "HS", 1, 2),
]
MBV3_STEM = 16
MBV3_LAST = 960
MBV3_BN_EPS = 1e-3
BN_EPS = 1e-5
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)

RESNET_LAYERS = [3, 4, 6, 3]
RESNET_PLANES = [64, 128, 256, 512]

VARIANTS = {
    # feature channels f1..f4 (f4 = after LR-ASPP), decoder channels
    "mobilenetv3": dict(enc_out=960, aspp=128, feat=[16, 24, 40, 128], dec=[80, 40, 32, 16]),
    "resnet50": dict(enc_out=2048, aspp=256, feat=[64, 256, 512, 256], dec=[128, 64, 32, 16]),
}


def make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def se_squeeze(expanded):
    return make_divisible(expanded // 4, 8)


def _bn(d, prefix, c):
    d[prefix + ".weight"] = (c,)
    d[prefix + ".bias"] = (c,)
    d[prefix + ".running_mean"] = (c,)
    d[prefix + ".running_var"] = (c,)
    d[prefix + ".num_batches_tracked"] = ()


def mbv3_block_layout(row):
    """Sub-module kinds inside ``features[i].block`` in index order."""
    cin, k, exp, cout, use_se, act, stride, dil = row
    kinds = []
    if exp != cin:
        kinds.append("expand")
    kinds.append("dw")
    if use_se:
        kinds.append("se")
    kinds.append("project")
    return kinds


def param_shapes(variant="mobilenetv3", refiner="deep_guided_filter"):
    """OrderedDict name -> shape with exactly the reference's state_dict keys."""
    v = VARIANTS[variant]
    d = OrderedDict()
    if variant == "mobilenetv3":
        d["backbone.features.0.0.weight"] = (MBV3_STEM, 3, 3, 3)
        _bn(d, "backbone.features.0.1", MBV3_STEM)
        for i, row in enumerate(MBV3_ROWS):
            cin, k, exp, cout, use_se, act, stride, dil = row
            p = "backbone.features.%d.block" % (i + 1)
            for j, kind in enumerate(mbv3_block_layout(row)):
                q = "%s.%d" % (p, j)
                if kind == "expand":
                    d[q + ".0.weight"] = (exp, cin, 1, 1)
                    _bn(d, q + ".1", exp)
                elif kind == "dw":
                    d[q + ".0.weight"] = (exp, 1, k, k)
                    _bn(d, q + ".1", exp)
                elif kind == "se":
                    sq = se_squeeze(exp)
                    d[q + ".fc1.weight"] = (sq, exp, 1, 1)
                    d[q + ".fc1.bias"] = (sq,)
                    d[q + ".fc2.weight"] = (exp, sq, 1, 1)
                    d[q + ".fc2.bias"] = (exp,)
                else:
                    d[q + ".0.weight"] = (cout, exp, 1, 1)
                    _bn(d, q + ".1", cout)
        d["backbone.features.16.0.weight"] = (MBV3_LAST, MBV3_ROWS[-1][3], 1, 1)
        _bn(d, "backbone.features.16.1", MBV3_LAST)
    else:
        d["backbone.conv1.weight"] = (64, 3, 7, 7)
        _bn(d, "backbone.bn1", 64)
        inplanes = 64
        for li, (nblk, planes) in enumerate(zip(RESNET_LAYERS, RESNET_PLANES)):
            for b in range(nblk):
                p = "backbone.layer%d.%d" % (li + 1, b)
                d[p + ".conv1.weight"] = (planes, inplanes, 1, 1)
                _bn(d, p + ".bn1", planes)
                d[p + ".conv2.weight"] = (planes, planes, 3, 3)
                _bn(d, p + ".bn2", planes)
                d[p + ".conv3.weight"] = (planes * 4, planes, 1, 1)
                _bn(d, p + ".bn3", planes * 4)
                if b == 0:
                    d[p + ".downsample.0.weight"] = (planes * 4, inplanes, 1, 1)
                    _bn(d, p + ".downsample.1", planes * 4)
                inplanes = planes * 4
    # LR-ASPP
    d["aspp.aspp1.0.weight"] = (v["aspp"], v["enc_out"], 1, 1)
    _bn(d, "aspp.aspp1.1", v["aspp"])
    d["aspp.aspp2.1.weight"] = (v["aspp"], v["enc_out"], 1, 1)
    # decoder
    feat, dec = v["feat"], v["dec"]

    def gru(prefix, c):
        d[prefix + ".ih.0.weight"] = (2 * c, 2 * c, 3, 3)
        d[prefix + ".ih.0.bias"] = (2 * c,)
        d[prefix + ".hh.0.weight"] = (c, 2 * c, 3, 3)
        d[prefix + ".hh.0.bias"] = (c,)

    gru("decoder.decode4.gru", feat[3] // 2)
    ups = [("decode3", feat[3], feat[2], dec[0]),
           ("decode2", dec[0], feat[1], dec[1]),
           ("decode1", dec[1], feat[0], dec[2])]
    for name, cin, cskip, cout in ups:
        d["decoder.%s.conv.0.weight" % name] = (cout, cin + cskip + 3, 3, 3)
        _bn(d, "decoder.%s.conv.1" % name, cout)
        gru("decoder.%s.gru" % name, cout // 2)
    d["decoder.decode0.conv.0.weight"] = (dec[3], dec[2] + 3, 3, 3)
    _bn(d, "decoder.decode0.conv.1", dec[3])
    d["decoder.decode0.conv.3.weight"] = (dec[3], dec[3], 3, 3)
    _bn(d, "decoder.decode0.conv.4", dec[3])
    d["project_mat.conv.weight"] = (4, 16, 1, 1)
    d["project_mat.conv.bias"] = (4,)
    d["project_seg.conv.weight"] = (1, 16, 1, 1)
    d["project_seg.conv.bias"] = (1,)
    if refiner == "deep_guided_filter":
        d["refiner.box_filter.weight"] = (4, 1, 3, 3)
        d["refiner.conv.0.weight"] = (16, 24, 1, 1)
        _bn(d, "refiner.conv.1", 16)
        d["refiner.conv.3.weight"] = (16, 16, 1, 1)
        _bn(d, "refiner.conv.4", 16)
        d["refiner.conv.6.weight"] = (4, 16, 1, 1)
        d["refiner.conv.6.bias"] = (4,)
    return d


def state_channels(variant="mobilenetv3"):
    """Channels of r1..r4 (decoder.py:80,61: half of each block's width)."""
    v = VARIANTS[variant]
    return [v["dec"][2] // 2, v["dec"][1] // 2, v["dec"][0] // 2, v["feat"][3] // 2]


def lowres_size(H, W, ds):
    """Output size of F.interpolate(scale_factor=ds) (model.py:80-81): floor(in*ds)."""
    if ds == 1:
        return H, W
    import math
    return int(math.floor(float(H) * ds)), int(math.floor(float(W) * ds))


def pyramid_sizes(h, w):
    """Sizes after 1..4 ceil-halvings (decoder.py:33, stride-2 stages of the encoders)."""
    out = []
    for _ in range(4):
        h, w = (h + 1) // 2, (w + 1) // 2
        out.append((h, w))
    return out
