"""Generate tests/golden/*.npz.  Runs in the BUILD CONTAINER only (needs /root/reference).

Two kinds of fixture:

1. ``ref_blocks_<variant>.npz`` - outputs of the reference's OWN classes
   (model/decoder.py, model/lraspp.py, model/deep_guided_filter.py,
   model/fast_guided_filter.py, imported by file path; they only need torch) on
   seeded inputs and seeded weights.  They pin oracle/rvm_oracle.py.
   The encoders (model/mobilenetv3.py, model/resnet.py) and model/model.py import
   torchvision, which is absent -> ordinary ModuleNotFoundError -> not run;
   those parts of the oracle stay "parity unpinned" (see rvm_oracle.py header).

2. ``e2e_<variant>_<tag>.npz`` - end-to-end vectors produced by the (pinned
   where possible) oracle itself: seed, clip spec, per-frame fgr/pha at a tiny
   size, final r1..r4.  They freeze the oracle against accidental edits and
   give the GPU tests a fixture that does not need the oracle to run.

Usage:  python oracle/make_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rvm_oracle as O                      # noqa: E402
from robustvideomatting_amd import spec, synthetic      # noqa: E402

REF = "/root/reference/model"
OUT = os.path.join(ROOT, "tests", "golden")


def _load(name):
    sp = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    m = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(m)
    return m


def _sub(params, prefix):
    return {k[len(prefix):]: v for k, v in params.items() if k.startswith(prefix)}


@torch.no_grad()
def ref_blocks(variant):
    torch.manual_seed(0)
    dec_m, asp_m = _load("decoder"), _load("lraspp")
    dgf_m, fgf_m = _load("deep_guided_filter"), _load("fast_guided_filter")
    v = spec.VARIANTS[variant]
    p = synthetic.make_params(variant, "deep_guided_filter", seed=3, perturb_box_filter=True)
    g = torch.Generator().manual_seed(11)
    h, w = 30, 54                                   # odd pyramid: 15x27, 8x14, 4x7, 2x4
    (h1, w1), (h2, w2), (h3, w3), (h4, w4) = spec.pyramid_sizes(h, w)
    T = 3
    feat = v["feat"]
    fix = {}

    def rnd(*s):
        return torch.randn(*s, generator=g)

    # ---- LR-ASPP
    aspp = asp_m.LRASPP(v["enc_out"], v["aspp"]).eval()
    aspp.load_state_dict(_sub(p, "aspp."))
    x = rnd(2, v["enc_out"], h4, w4)
    y_ref = aspp(x)
    y_or = O.lraspp(p, x)
    fix["aspp_in"], fix["aspp_out"] = x, y_ref
    print("lraspp     max|ref-oracle| = %.3g" % (y_ref - y_or).abs().max())
    assert (y_ref - y_or).abs().max() < 1e-6

    # ---- decoder: frame-by-frame with state carry, then the 5-D time-series path
    dec = dec_m.RecurrentDecoder(feat, v["dec"]).eval()
    dec.load_state_dict(_sub(p, "decoder."))
    s0 = torch.rand(1, T, 3, h, w, generator=g)
    f1, f2 = rnd(1, T, feat[0], h1, w1) * 0.7, rnd(1, T, feat[1], h2, w2) * 0.7
    f3, f4 = rnd(1, T, feat[2], h3, w3) * 0.7, rnd(1, T, feat[3], h4, w4).abs() * 0.7
    rec_ref = [None] * 4
    rec_or = [None] * 4
    hid_ref = []
    for t in range(T):
        o_ref = dec(s0[:, t], f1[:, t], f2[:, t], f3[:, t], f4[:, t], *rec_ref)
        o_or = O.decoder(p, s0[:, t], f1[:, t], f2[:, t], f3[:, t], f4[:, t], *rec_or)
        for a, b in zip(o_ref, o_or):
            assert (a - b).abs().max() < 1e-6, (a - b).abs().max()
        rec_ref, rec_or = list(o_ref[1:]), list(o_or[1:])
        hid_ref.append(o_ref[0])
    o_ts = dec(s0, f1, f2, f3, f4, None, None, None, None)       # time-series path
    d_ts = (o_ts[0] - torch.stack(hid_ref, 1)).abs().max()
    print("decoder    frame-by-frame == reference time-series path: %.3g" % d_ts)
    assert d_ts < 1e-5
    for a, b in zip(o_ts[1:], rec_ref):
        assert (a - b).abs().max() < 1e-5
    fix.update(dec_s0=s0, dec_f1=f1, dec_f2=f2, dec_f3=f3, dec_f4=f4,
               dec_hid=torch.stack(hid_ref, 1), dec_r1=rec_ref[0], dec_r2=rec_ref[1],
               dec_r3=rec_ref[2], dec_r4=rec_ref[3])
    print("decoder    oracle == reference classes over T=%d frames" % T)

    # ---- projection (decoder.py:193-199)
    proj = dec_m.Projection(16, 4).eval()
    proj.load_state_dict(_sub(p, "project_mat."))
    y = proj(hid_ref[0])
    fix["proj_out"] = y

    # ---- refiners
    H, W = 4 * h + 1, 4 * w + 2                      # floor(H*0.25)=h, floor(W*0.25)=w, not exact multiples
    fine = torch.rand(1, 3, H, W, generator=g)
    base = torch.rand(1, 3, h, w, generator=g)
    fgr, pha, hid = rnd(1, 3, h, w) * 0.2, torch.rand(1, 1, h, w, generator=g), hid_ref[0]
    dgf = dgf_m.DeepGuidedFilterRefiner().eval()
    dgf.load_state_dict(_sub(p, "refiner."))
    a_ref = dgf(fine, base, fgr, pha, hid)
    a_or = O.deep_guided_filter(p, fine, base, fgr, pha, hid)
    fgf = fgf_m.FastGuidedFilterRefiner().eval()
    b_ref = fgf(fine, base, fgr, pha, hid)
    b_or = O.fast_guided_filter(fine, base, fgr, pha)
    for a, b in zip(a_ref + b_ref, a_or + b_or):
        assert (a - b).abs().max() < 1e-5, (a - b).abs().max()
    print("refiners   oracle == reference classes (deep + fast)")
    fix.update(ref_fine=fine, ref_base=base, ref_fgr=fgr, ref_pha=pha,
               dgf_fgr=a_ref[0], dgf_pha=a_ref[1], fgf_fgr=b_ref[0], fgf_pha=b_ref[1])
    np.savez_compressed(os.path.join(OUT, "ref_blocks_%s.npz" % variant),
                        **{k: v_.numpy().astype(np.float32) for k, v_ in fix.items()})


@torch.no_grad()
def ref_decoder_chunk(variant="mobilenetv3"):
    """SURVEY 8(f-2) pin: the reference decoder's TIME-SERIES path (decoder.py:92-105,134-143,174-180) on a
    [B=2, T=4] chunk that starts from a non-zero state; the oracle walks the same chunk frame by frame."""
    dec_m = _load("decoder")
    v = spec.VARIANTS[variant]
    p = synthetic.make_params(variant, "deep_guided_filter", seed=4)
    g = torch.Generator().manual_seed(21)
    B, T, h, w = 2, 4, 22, 38                       # pyramid 11x19, 6x10, 3x5, 2x3
    sizes = spec.pyramid_sizes(h, w)
    feat = v["feat"]
    dec = dec_m.RecurrentDecoder(feat, v["dec"]).eval()
    dec.load_state_dict(_sub(p, "decoder."))
    s0 = torch.rand(B, T, 3, h, w, generator=g)
    fs = [torch.randn(B, T, feat[i], *sizes[i], generator=g) * 0.7 for i in range(4)]
    fs[3] = fs[3].abs()
    rc = spec.state_channels(variant)
    r0 = [torch.tanh(torch.randn(B, rc[i], *sizes[i], generator=g)) for i in range(4)]
    o_ref = dec(s0, *fs, *r0)
    rec = list(r0)
    worst = 0.0
    for t in range(T):
        o = O.decoder(p, s0[:, t], *[f[:, t] for f in fs], *rec)
        rec = list(o[1:])
        worst = max(worst, (o[0] - o_ref[0][:, t]).abs().max().item())
    for a, b in zip(rec, o_ref[1:]):
        worst = max(worst, (a - b).abs().max().item())
    print("decoder    chunk [B=2,T=4] reference time-series path vs oracle frame by frame: %.3g" % worst)
    assert worst < 1e-5
    fix = dict(s0=s0, f1=fs[0], f2=fs[1], f3=fs[2], f4=fs[3], r1_in=r0[0], r2_in=r0[1], r3_in=r0[2], r4_in=r0[3],
               hid=o_ref[0], r1=o_ref[1], r2=o_ref[2], r3=o_ref[3], r4=o_ref[4])
    np.savez_compressed(os.path.join(OUT, "ref_decoder_chunk_%s.npz" % variant),
                        **{k: v_.numpy().astype(np.float32) for k, v_ in fix.items()})


@torch.no_grad()
def e2e(variant, refiner, tag, H, W, ds, T, seed):
    p = synthetic.make_params(variant, refiner, seed=seed)
    clip = synthetic.make_clip(T, H, W, seed=seed)
    rec = [None] * 4
    fgrs, phas = [], []
    for t in range(T):
        o = O.forward(p, clip[:, t], *rec, downsample_ratio=ds, variant=variant, refiner=refiner)
        rec = o[2:]
        fgrs.append(o[0])
        phas.append(o[1])
    fgr, pha = torch.stack(fgrs, 1), torch.stack(phas, 1)
    print("e2e %-12s %-9s fgr mean %.4f  pha mean %.4f  pha in (0,1): %.1f%%" % (
        variant, tag, fgr.mean(), pha.mean(), 100 * ((pha > 0) & (pha < 1)).float().mean()))
    np.savez_compressed(
        os.path.join(OUT, "e2e_%s_%s.npz" % (variant, tag)),
        meta=np.array([H, W, T, seed], dtype=np.int64), ds=np.array(ds, dtype=np.float64),
        refiner=np.array(refiner),
        fgr=fgr.numpy().astype(np.float16), pha=pha.numpy().astype(np.float32),
        fgr_last=fgr[:, -1].numpy(), r1=rec[0].numpy(), r2=rec[1].numpy(), r3=rec[2].numpy(), r4=rec[3].numpy())


def encoder_tables():
    """What the reference ITSELF holds about the encoders and the glue, read as literals with `ast` - nothing is imported
    or run (model/mobilenetv3.py, model/resnet.py and model/model.py need torchvision, which is absent): the 15
    InvertedResidualConfig rows and last_channel (mobilenetv3.py:9-27), the normalisation constants (:37), the feature
    taps (:39-59); ResNet50's layers / dilation flags (resnet.py:7-11) and taps (:20-33); the LR-ASPP / decoder widths of
    both variants (model.py:25-30).  Written to tests/golden/ref_encoder_tables.json; tests/test_encoder_tables.py holds
    the oracle's, the Python spec's and the native library's tables against it."""
    import ast
    import json

    def lit(node):
        return ast.literal_eval(node)

    out = {}
    tree = ast.parse(open(os.path.join(REF, "mobilenetv3.py")).read())
    rows, taps, last, mean, std = [], {}, None, None, None
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "id", None) == "InvertedResidualConfig":
            rows.append([lit(a) for a in node.args])
        if isinstance(node, ast.keyword) and node.arg == "last_channel":
            last = lit(node.value)
        if isinstance(node, ast.Call) and getattr(node.func, "id", None) == "normalize":
            mean, std = lit(node.args[1]), lit(node.args[2])
    fwd = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "forward_single_frame"][0]
    cur = None
    for st in fwd.body:
        if isinstance(st, ast.Assign) and isinstance(st.value, ast.Call) and isinstance(st.value.func, ast.Subscript):
            cur = lit(st.value.func.slice)                      # x = self.features[i](x)
        elif isinstance(st, ast.Assign) and isinstance(st.value, ast.Name) and st.targets[0].id.startswith("f"):
            taps[st.targets[0].id] = cur                        # fK = x
    out["mobilenetv3"] = dict(inverted_residual_setting=rows, last_channel=last, normalize_mean=mean, normalize_std=std,
                              taps_after_feature=taps, num_features=cur + 1)
    tree = ast.parse(open(os.path.join(REF, "resnet.py")).read())
    kw = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.keyword) and node.arg in ("layers", "replace_stride_with_dilation"):
            kw[node.arg] = lit(node.value)
        if isinstance(node, ast.keyword) and node.arg == "block":
            kw["block"] = node.value.id
    fwd = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "forward_single_frame"][0]
    order, rtaps, cur = [], {}, None
    for st in fwd.body:
        if isinstance(st, ast.Assign) and isinstance(st.value, ast.Call) and isinstance(st.value.func, ast.Attribute):
            cur = st.value.func.attr                            # x = self.<module>(x)
            order.append(cur)
        elif isinstance(st, ast.Assign) and isinstance(st.value, ast.Name) and st.targets[0].id.startswith("f"):
            rtaps[st.targets[0].id] = cur
    out["resnet50"] = dict(kw, module_order=order, taps_after_module=rtaps)
    tree = ast.parse(open(os.path.join(REF, "model.py")).read())
    glue = {"LRASPP": [], "RecurrentDecoder": [], "Projection": []}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "id", None) in glue:
            glue[node.func.id].append([lit(a) for a in node.args])
    out["model"] = glue                                         # [mobilenetv3, resnet50] order of the if / else in model.py:24-30
    with open(os.path.join(OUT, "ref_encoder_tables.json"), "w") as f:
        json.dump(out, f, sort_keys=True)
    print("ref_encoder_tables.json:", len(rows), "rows; taps", taps, rtaps)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if "--chunk-only" in sys.argv:
        ref_decoder_chunk()
        sys.exit(0)
    if "--tables-only" in sys.argv:
        encoder_tables()
        sys.exit(0)
    encoder_tables()
    ref_blocks("mobilenetv3")
    ref_blocks("resnet50")
    ref_decoder_chunk()
    e2e("mobilenetv3", "deep_guided_filter", "ds025", 122, 218, 0.25, 4, 5)
    e2e("mobilenetv3", "deep_guided_filter", "ds1", 36, 62, 1.0, 4, 6)
    e2e("mobilenetv3", "fast_guided_filter", "fast", 122, 218, 0.25, 2, 7)
    e2e("resnet50", "deep_guided_filter", "ds025", 122, 218, 0.25, 2, 8)
