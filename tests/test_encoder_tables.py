"""CPU: the encoder / glue tables of the oracle, the Python spec and the packed-weight layout against the literals the
reference itself holds (tests/golden/ref_encoder_tables.json, read out of /root/reference/model/{mobilenetv3,resnet,model}.py
with `ast` by oracle/make_golden.py - torchvision, which those files import, is absent, so nothing of them can be run).
This is the only reference-held pin the encoder rows have ("parity unpinned" beyond it)."""
import json
import os

import torch

from oracle import rvm_oracle as O
from robustvideomatting_amd import spec, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_encoder_tables.json")))


def test_mobilenetv3_rows_match_the_reference_literals():
    rows = REF["mobilenetv3"]["inverted_residual_setting"]
    assert len(rows) == 15 and all(r[8] == 1 for r in rows)                   # width_mult 1: channels are used as written
    want = [tuple(r[:8]) for r in rows]
    assert [tuple(r) for r in O.MBV3_ROWS] == want
    assert [tuple(r) for r in spec.MBV3_ROWS] == want
    assert list(spec.IMAGENET_MEAN) == REF["mobilenetv3"]["normalize_mean"]
    assert list(spec.IMAGENET_STD) == REF["mobilenetv3"]["normalize_std"]
    assert spec.MBV3_STEM == rows[0][0]
    assert spec.MBV3_LAST == 6 * rows[-1][3]                                  # torchvision: lastconv = 6 x the last block's width
    assert REF["mobilenetv3"]["num_features"] == len(rows) + 2                # stem + 15 blocks + last conv
    assert REF["mobilenetv3"]["taps_after_feature"] == {"f1": 1, "f2": 3, "f3": 6, "f4": 16}


def test_oracle_encoder_taps_have_the_shapes_the_reference_tables_imply():
    rows = REF["mobilenetv3"]["inverted_residual_setting"]
    taps = REF["mobilenetv3"]["taps_after_feature"]
    p = synthetic.make_params("mobilenetv3", "deep_guided_filter", seed=1)
    x = torch.rand(1, 3, 64, 96)
    feats = O.mobilenetv3_encoder(p, x)
    for name, f in zip(("f1", "f2", "f3", "f4"), feats):
        i = taps[name]
        ch = 6 * rows[-1][3] if i == 16 else rows[i - 1][3]
        scale = 2                                                              # the stem's stride
        for r in rows[:min(i, 15)]:
            if r[6] == 2 and r[7] == 1:                                       # a dilated block keeps its resolution (stride -> 1)
                scale *= 2
        assert f.shape == (1, ch, 64 // scale, 96 // scale), (name, f.shape, ch, scale)
    # the block whose config asks for stride 2 AND dilation 2 (row 13) must not downsample: f4 stays at 1/16
    assert rows[12][6] == 2 and rows[12][7] == 2 and feats[3].shape[-2:] == (4, 6)


def test_resnet50_layout_matches_the_reference_literals():
    r = REF["resnet50"]
    assert r["block"] == "Bottleneck" and r["layers"] == spec.RESNET_LAYERS == [3, 4, 6, 3]
    assert r["replace_stride_with_dilation"] == [False, False, True]
    assert r["module_order"] == ["conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4"]
    assert r["taps_after_module"] == {"f1": "relu", "f2": "layer1", "f3": "layer2", "f4": "layer4"}
    p = synthetic.make_params("resnet50", "deep_guided_filter", seed=1)
    f1, f2, f3, f4 = O.resnet50_encoder(p, torch.rand(1, 3, 64, 96))
    assert f1.shape == (1, 64, 32, 48) and f2.shape == (1, 256, 16, 24) and f3.shape == (1, 512, 8, 12)
    assert f4.shape == (1, 2048, 4, 6)                                        # layer4 dilated, not strided: 1/16
    names = [k for k in spec.param_shapes("resnet50", "deep_guided_filter") if k.startswith("backbone.layer") and k.endswith("conv1.weight")]
    assert [sum(1 for k in names if k.startswith("backbone.layer%d." % (l + 1))) for l in range(4)] == r["layers"]


def test_glue_widths_match_model_py():
    g = REF["model"]
    assert g["LRASPP"] == [[960, 128], [2048, 256]]
    assert g["RecurrentDecoder"] == [[[16, 24, 40, 128], [80, 40, 32, 16]], [[64, 256, 512, 256], [128, 64, 32, 16]]]
    assert g["Projection"] == [[16, 4], [16, 1]]
    for k, variant in enumerate(("mobilenetv3", "resnet50")):
        v = spec.VARIANTS[variant]
        assert [v["enc_out"], v["aspp"]] == g["LRASPP"][k]
        assert [v["feat"], v["dec"]] == g["RecurrentDecoder"][k]


def test_state_dict_shapes_follow_the_reference_rows():
    """The packed-weight layout starts from spec.param_shapes (and its native twin, checked against it on the GPU box in
    tests/test_c_abi_gpu.py): every conv of every block has the shape the reference row implies."""
    rows = REF["mobilenetv3"]["inverted_residual_setting"]
    shapes = spec.param_shapes("mobilenetv3", "deep_guided_filter")
    for i, (cin, k, exp, cout, se, act, stride, dil, _) in enumerate(rows):
        pre = "backbone.features.%d.block." % (i + 1)
        j = 0
        if exp != cin:
            assert shapes[pre + "%d.0.weight" % j] == (exp, cin, 1, 1)
            j += 1
        assert shapes[pre + "%d.0.weight" % j] == (exp, 1, k, k)
        j += 1
        if se:
            sq = spec.se_squeeze(exp)
            assert shapes[pre + "%d.fc1.weight" % j] == (sq, exp, 1, 1) and shapes[pre + "%d.fc2.weight" % j] == (exp, sq, 1, 1)
            j += 1
        assert shapes[pre + "%d.0.weight" % j] == (cout, exp, 1, 1)
    assert shapes["backbone.features.16.0.weight"] == (960, 160, 1, 1)
