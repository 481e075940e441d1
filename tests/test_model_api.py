"""CPU: the drop-in surface of MattingNetwork (ctor, state_dict keys, forward signature)."""
import inspect

import pytest
import torch

from robustvideomatting_amd import MattingNetwork, spec, synthetic


@pytest.mark.parametrize("variant,refiner", [("mobilenetv3", "deep_guided_filter"),
                                             ("mobilenetv3", "fast_guided_filter"),
                                             ("resnet50", "deep_guided_filter")])
def test_state_dict_contract(variant, refiner):
    m = MattingNetwork(variant, refiner)
    sd = m.state_dict()
    shapes = spec.param_shapes(variant, refiner)
    assert set(sd) == set(shapes)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    # strict load of a foreign state_dict (inference.py:165 of the reference)
    p = synthetic.make_params(variant, refiner, seed=2)
    res = m.load_state_dict(p)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(m.state_dict()["project_mat.conv.bias"], p["project_mat.conv.bias"])
    with pytest.raises(RuntimeError):
        bad = dict(p)
        bad.pop("aspp.aspp2.1.weight")
        m.load_state_dict(bad)


def test_known_keys_from_survey():
    sd = MattingNetwork().state_dict()
    assert tuple(sd["decoder.decode3.conv.0.weight"].shape) == (80, 171, 3, 3)
    assert tuple(sd["decoder.decode4.gru.ih.0.weight"].shape) == (128, 128, 3, 3)
    assert tuple(sd["refiner.box_filter.weight"].shape) == (4, 1, 3, 3)
    assert tuple(sd["backbone.features.4.block.2.fc1.weight"].shape) == (24, 72, 1, 1)
    assert tuple(sd["backbone.features.16.0.weight"].shape) == (960, 160, 1, 1)
    assert torch.allclose(sd["refiner.box_filter.weight"], torch.full((4, 1, 3, 3), 1 / 9))


def test_constructor_contract():
    with pytest.raises(AssertionError):
        MattingNetwork("vgg")
    with pytest.raises(AssertionError):
        MattingNetwork("mobilenetv3", "bilateral")
    with pytest.raises(RuntimeError):
        MattingNetwork(pretrained_backbone=True)


def test_forward_signature_matches_reference():
    # reference model/model.py:40-48 (this fork: trimap slot after src)
    names = list(inspect.signature(MattingNetwork.forward).parameters)
    assert names == ["self", "src", "trimap", "r1", "r2", "r3", "r4", "downsample_ratio", "segmentation_pass"]


def test_no_cpu_fallback():
    m = MattingNetwork().eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 3, 32, 32))


def test_call_forms(monkeypatch):
    """Fork form (src, trimap, r1..r4, ds), upstream form (src, r1..r4, ds) and keywords."""
    m = MattingNetwork()
    seen = {}

    def fake(src, rec, ds, seg):
        seen.update(rec=rec, ds=ds, seg=seg)
        return []
    monkeypatch.setattr(m, "_run", fake)
    src = torch.rand(1, 3, 64, 64)
    r = [torch.rand(1, c, 4, 4) for c in (16, 20, 40, 64)]
    m(src, src, *r, 0.25)                                   # inference.py:129 of the reference
    assert seen["ds"] == 0.25 and all(a is b for a, b in zip(seen["rec"], r))
    m(src, *r, 0.5)                                         # upstream README form
    assert seen["ds"] == 0.5 and all(a is b for a, b in zip(seen["rec"], r))
    m(src, *r, downsample_ratio=0.5)                        # upstream form with keyword ratio
    assert seen["ds"] == 0.5 and all(a is b for a, b in zip(seen["rec"], r))
    m(src, r1=r[0], r2=r[1], r3=r[2], r4=r[3], downsample_ratio=0.125, segmentation_pass=True)
    assert seen["ds"] == 0.125 and seen["seg"] and seen["rec"][3] is r[3]
    m(src, None, None, None, None, None, 1)
    assert seen["rec"] == [None] * 4 and seen["ds"] == 1.0
    m.forward_upstream(src, *r, 0.25)
    assert seen["ds"] == 0.25 and seen["rec"][0] is r[0]


def test_half_and_to_keep_parameters():
    m = MattingNetwork().half()
    assert m.project_mat.conv.weight.dtype == torch.float16
    assert m._dirty
