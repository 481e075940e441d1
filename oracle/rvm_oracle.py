"""CPU fp32 oracle for the RVM per-frame inference path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The shipped path (``robustvideomatting_amd``) never calls into it and
fails loudly when the HIP library is missing.

It restates, as plain functional torch ops on a ``{name: tensor}`` parameter
dict, the arithmetic of these reference files (citations are file:line under
/root/reference):

  model/model.py:52-68,73-82          top-level glue, downsample, residual+clamp
  model/mobilenetv3.py:9-27,36-60     MobileNetV3-Large encoder taps
  model/resnet.py:7-11,20-33          ResNet50 encoder taps
  model/lraspp.py:6-18                LR-ASPP
  model/decoder.py:21-27,33-39,63-67,74-90,117-132,159-172,183-185,196-199
  model/deep_guided_filter.py:12-43   DeepGuidedFilterRefiner
  model/fast_guided_filter.py:14-23,50-59,67-76

Pinning status
--------------
* decoder / LR-ASPP / both refiners: PINNED - ``oracle/make_golden.py`` runs the
  reference's own classes (imported by file path in the build container) on
  seeded weights and the restatement below must agree to 1e-6
  (``tests/golden/ref_blocks_*.npz``, checked by ``tests/test_oracle_golden.py``).
* encoders (MobileNetV3 / ResNet50) and the 30 lines of glue in model/model.py:
  the reference delegates their arithmetic to torchvision (pinned
  torchvision==0.10.0, requirements_inference.txt:3), which is NOT in the
  container and not vendored; the reference has no tests or golden vectors.
  Those parts are restated from torchvision's published architecture and are
  "parity unpinned" beyond the state_dict key/shape contract and the channel /
  stride contract the (pinned) decoder consumes.
"""
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

# --- architecture tables (kept local so the oracle does not depend on the product package)
MBV3_ROWS = [
    (16, 3, 16, 16, False, "RE", 1, 1), (16, 3, 64, 24, False, "RE", 2, 1),
    (24, 3, 72, 24, False, "RE", 1, 1), (24, 5, 72, 40, True, "RE", 2, 1),
    (40, 5, 120, 40, True, "RE", 1, 1), (40, 5, 120, 40, True, "RE", 1, 1),
    (40, 3, 240, 80, False, "HS", 2, 1), (80, 3, 200, 80, False, "HS", 1, 1),
    (80, 3, 184, 80, False, "HS", 1, 1), (80, 3, 184, 80, False, "HS", 1, 1),
    (80, 3, 480, 112, True, "HS", 1, 1), (112, 3, 672, 112, True, "HS", 1, 1),
    (112, 5, 672, 160, True, "HS", 2, 2), (160, 5, 960, 160, True, "HS", 1, 2),
    (160, 5, 960, 160, True, "HS", 1, 2),
]

P = Dict[str, torch.Tensor]


def _bn(p: P, prefix: str, x, eps):
    return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                        p[prefix + ".weight"], p[prefix + ".bias"], False, 0.0, eps)


def _act(x, kind):
    if kind == "RE":
        return F.relu(x)
    if kind == "HS":
        return F.hardswish(x)
    return x


# ------------------------------------------------------------------ encoders
def mobilenetv3_encoder(p: P, x):
    """mobilenetv3.py:36-60 (+ torchvision InvertedResidual / SqueezeExcitation)."""
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=x.dtype).view(1, 3, 1, 1)
    x = (x - mean) / std                                                    # :37
    x = F.conv2d(x, p["backbone.features.0.0.weight"], None, 2, 1)
    x = F.hardswish(_bn(p, "backbone.features.0.1", x, 1e-3))
    taps = {}
    for i, (cin, k, exp, cout, use_se, act, stride, dil) in enumerate(MBV3_ROWS):
        pre = "backbone.features.%d.block" % (i + 1)
        j = 0
        y = x
        if exp != cin:
            y = F.conv2d(y, p["%s.%d.0.weight" % (pre, j)])
            y = _act(_bn(p, "%s.%d.1" % (pre, j), y, 1e-3), act)
            j += 1
        s = 1 if dil > 1 else stride
        y = F.conv2d(y, p["%s.%d.0.weight" % (pre, j)], None, s, (k - 1) // 2 * dil, dil, exp)
        y = _act(_bn(p, "%s.%d.1" % (pre, j), y, 1e-3), act)
        j += 1
        if use_se:
            q = "%s.%d" % (pre, j)
            sc = y.mean((2, 3), keepdim=True)
            sc = F.relu(F.conv2d(sc, p[q + ".fc1.weight"], p[q + ".fc1.bias"]))
            sc = F.hardsigmoid(F.conv2d(sc, p[q + ".fc2.weight"], p[q + ".fc2.bias"]))
            y = y * sc
            j += 1
        y = F.conv2d(y, p["%s.%d.0.weight" % (pre, j)])
        y = _bn(p, "%s.%d.1" % (pre, j), y, 1e-3)
        if stride == 1 and cin == cout:
            y = y + x
        x = y
        if i + 1 in (1, 3, 6):
            taps[i + 1] = x
    x = F.conv2d(x, p["backbone.features.16.0.weight"])
    x = F.hardswish(_bn(p, "backbone.features.16.1", x, 1e-3))
    return taps[1], taps[3], taps[6], x


def resnet50_encoder(p: P, x):
    """resnet.py:20-33 (+ torchvision Bottleneck v1.5, replace_stride_with_dilation=[F,F,T])."""
    x = F.conv2d(x, p["backbone.conv1.weight"], None, 2, 3)
    x = F.relu(_bn(p, "backbone.bn1", x, 1e-5))
    f1 = x
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    dilation = 1
    for li, (nblk, planes) in enumerate(zip([3, 4, 6, 3], [64, 128, 256, 512])):
        stride = 1 if li == 0 else 2
        prev_dil = dilation
        if li == 3:                     # dilate instead of stride
            dilation *= stride
            stride = 1
        for b in range(nblk):
            q = "backbone.layer%d.%d" % (li + 1, b)
            s = stride if b == 0 else 1
            d = prev_dil if b == 0 else dilation
            idt = x
            y = F.relu(_bn(p, q + ".bn1", F.conv2d(x, p[q + ".conv1.weight"]), 1e-5))
            y = F.relu(_bn(p, q + ".bn2", F.conv2d(y, p[q + ".conv2.weight"], None, s, d, d), 1e-5))
            y = _bn(p, q + ".bn3", F.conv2d(y, p[q + ".conv3.weight"]), 1e-5)
            if b == 0:
                idt = _bn(p, q + ".downsample.1", F.conv2d(x, p[q + ".downsample.0.weight"], None, s), 1e-5)
            x = F.relu(y + idt)
        feats.append(x)
    return f1, feats[0], feats[1], feats[3]


# ------------------------------------------------------------------ LR-ASPP
def lraspp(p: P, x):
    """lraspp.py:6-18."""
    a = F.relu(_bn(p, "aspp.aspp1.1", F.conv2d(x, p["aspp.aspp1.0.weight"]), 1e-5))
    g = torch.sigmoid(F.conv2d(x.mean((2, 3), keepdim=True), p["aspp.aspp2.1.weight"]))
    return a * g


# ------------------------------------------------------------------ decoder
def conv_gru(p: P, prefix, x, h):
    """decoder.py:168-172,183-185.  h=None -> zeros; gates ordered (r, z)."""
    if h is None:
        h = torch.zeros_like(x)
    c = x.shape[1]
    rz = torch.sigmoid(F.conv2d(torch.cat([x, h], 1), p[prefix + ".ih.0.weight"], p[prefix + ".ih.0.bias"], 1, 1))
    r, z = rz[:, :c], rz[:, c:]
    cand = torch.tanh(F.conv2d(torch.cat([x, r * h], 1), p[prefix + ".hh.0.weight"], p[prefix + ".hh.0.bias"], 1, 1))
    return (1 - z) * h + z * cand


def _up2(x, like):
    """decoder.py:74,83-84: bilinear x2 (align_corners=False) then crop to skip size."""
    x = F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=False)
    return x[:, :, :like.shape[2], :like.shape[3]]


def _avgpool(x):
    return F.avg_pool2d(x, 2, 2, count_include_pad=False, ceil_mode=True)      # decoder.py:33


def decoder(p: P, s0, f1, f2, f3, f4, r1, r2, r3, r4, taps=None):
    """decoder.py:17-27 on a single frame batch [B,C,H,W]."""
    s1 = _avgpool(s0)
    s2 = _avgpool(s1)
    s3 = _avgpool(s2)
    # decode4: BottleneckBlock :63-67
    c = f4.shape[1] // 2
    r4 = conv_gru(p, "decoder.decode4.gru", f4[:, c:], r4)
    x = torch.cat([f4[:, :c], r4], 1)
    if taps is not None:
        taps["x4"] = x
    rs = []
    for name, f, s, r in (("decode3", f3, s3, r3), ("decode2", f2, s2, r2), ("decode1", f1, s1, r1)):
        q = "decoder." + name
        x = torch.cat([_up2(x, s), f, s], 1)                                   # :83-85
        x = F.relu(_bn(p, q + ".conv.1", F.conv2d(x, p[q + ".conv.0.weight"], None, 1, 1), 1e-5))
        c = x.shape[1] // 2
        r = conv_gru(p, q + ".gru", x[:, c:], r)
        x = torch.cat([x[:, :c], r], 1)
        rs.append(r)
        if taps is not None:
            taps["x" + name[-1]] = x
    r3, r2, r1 = rs
    q = "decoder.decode0"
    x = torch.cat([_up2(x, s0), s0], 1)                                        # :128-130
    x = F.relu(_bn(p, q + ".conv.1", F.conv2d(x, p[q + ".conv.0.weight"], None, 1, 1), 1e-5))
    x = F.relu(_bn(p, q + ".conv.4", F.conv2d(x, p[q + ".conv.3.weight"], None, 1, 1), 1e-5))
    return x, r1, r2, r3, r4


# ------------------------------------------------------------------ refiners
def deep_guided_filter(p: P, fine_src, base_src, base_fgr, base_pha, base_hid, taps=None):
    """deep_guided_filter.py:24-43."""
    fine_x = torch.cat([fine_src, fine_src.mean(1, keepdim=True)], 1)
    base_x = torch.cat([base_src, base_src.mean(1, keepdim=True)], 1)
    base_y = torch.cat([base_fgr, base_pha], 1)
    w = p["refiner.box_filter.weight"]

    def box(t):
        return F.conv2d(t, w, None, 1, 1, 1, 4)

    mean_x = box(base_x)
    mean_y = box(base_y)
    cov_xy = box(base_x * base_y) - mean_x * mean_y
    var_x = box(base_x * base_x) - mean_x * mean_x
    a = torch.cat([cov_xy, var_x, base_hid], 1)
    a = F.relu(_bn(p, "refiner.conv.1", F.conv2d(a, p["refiner.conv.0.weight"]), 1e-5))
    a = F.relu(_bn(p, "refiner.conv.4", F.conv2d(a, p["refiner.conv.3.weight"]), 1e-5))
    a = F.conv2d(a, p["refiner.conv.6.weight"], p["refiner.conv.6.bias"])
    b = mean_y - a * mean_x
    if taps is not None:
        taps["A"], taps["b"] = a, b
    H, W = fine_src.shape[2:]
    a = F.interpolate(a, (H, W), mode="bilinear", align_corners=False)
    b = F.interpolate(b, (H, W), mode="bilinear", align_corners=False)
    out = a * fine_x + b
    return out[:, :3], out[:, 3:]


def fast_guided_filter(fine_src, base_src, base_fgr, base_pha, taps=None):
    """fast_guided_filter.py:14-23,50-59,67-76 (r=1, eps=1e-5, separable 1/3 box)."""
    fine_x = torch.cat([fine_src, fine_src.mean(1, keepdim=True)], 1)
    base_x = torch.cat([base_src, base_src.mean(1, keepdim=True)], 1)
    base_y = torch.cat([base_fgr, base_pha], 1)
    kx = torch.full((4, 1, 1, 3), 1.0 / 3.0, dtype=fine_src.dtype)
    ky = torch.full((4, 1, 3, 1), 1.0 / 3.0, dtype=fine_src.dtype)

    def box(t):
        t = F.conv2d(t, kx, None, 1, (0, 1), 1, 4)
        return F.conv2d(t, ky, None, 1, (1, 0), 1, 4)

    mean_x = box(base_x)
    mean_y = box(base_y)
    cov_xy = box(base_x * base_y) - mean_x * mean_y
    var_x = box(base_x * base_x) - mean_x * mean_x
    a = cov_xy / (var_x + 1e-5)
    b = mean_y - a * mean_x
    if taps is not None:
        taps["A"], taps["b"] = a, b
    H, W = fine_src.shape[2:]
    a = F.interpolate(a, (H, W), mode="bilinear", align_corners=False)
    b = F.interpolate(b, (H, W), mode="bilinear", align_corners=False)
    out = a * fine_x + b
    return out[:, :3], out[:, 3:]


# ------------------------------------------------------------------ top level
def forward_frame(p: P, src, rec, downsample_ratio=1.0, variant="mobilenetv3",
                  refiner="deep_guided_filter", segmentation_pass=False, taps=None):
    """model.py:52-71 for one frame batch src[B,3,H,W]; rec = [r1,r2,r3,r4] (None allowed)."""
    if downsample_ratio != 1:
        src_sm = F.interpolate(src, scale_factor=downsample_ratio, mode="bilinear",
                               align_corners=False, recompute_scale_factor=False)   # :80-81
    else:
        src_sm = src
    enc = mobilenetv3_encoder if variant == "mobilenetv3" else resnet50_encoder
    f1, f2, f3, f4 = enc(p, src_sm)
    f4a = lraspp(p, f4)
    hid, r1, r2, r3, r4 = decoder(p, src_sm, f1, f2, f3, f4a, *rec, taps=taps)
    if taps is not None:
        taps.update(src_sm=src_sm, f1=f1, f2=f2, f3=f3, f4=f4, aspp=f4a, hid=hid)
    if segmentation_pass:
        seg = F.conv2d(hid, p["project_seg.conv.weight"], p["project_seg.conv.bias"])
        return [seg, r1, r2, r3, r4]
    y = F.conv2d(hid, p["project_mat.conv.weight"], p["project_mat.conv.bias"])     # :62
    fgr_res, pha = y[:, :3], y[:, 3:]
    if taps is not None:
        taps["proj"] = y
    if downsample_ratio != 1:                                                       # :63-64
        if refiner == "deep_guided_filter":
            fgr_res, pha = deep_guided_filter(p, src, src_sm, fgr_res, pha, hid, taps)
        else:
            fgr_res, pha = fast_guided_filter(src, src_sm, fgr_res, pha, taps)
    if taps is not None:
        taps["fgr_pre"], taps["pha_pre"] = fgr_res + src, pha
    fgr = (fgr_res + src).clamp(0.0, 1.0)                                           # :65-67
    pha = pha.clamp(0.0, 1.0)
    return [fgr, pha, r1, r2, r3, r4]


@torch.no_grad()
def forward(p: P, src, r1=None, r2=None, r3=None, r4=None, downsample_ratio=1.0,
            variant="mobilenetv3", refiner="deep_guided_filter", segmentation_pass=False,
            taps: Optional[List[dict]] = None):
    """Upstream-style call: src [B,3,H,W] or [B,T,3,H,W] -> [fgr, pha, r1..r4].

    A 5-D input is walked frame by frame; this equals the reference's
    time-series path, which batches the feed-forward parts over T and loops
    only the ConvGRU (decoder.py:174-180).
    """
    rec = [r1, r2, r3, r4]
    if src.ndim == 4:
        t = {} if taps is not None else None
        out = forward_frame(p, src, rec, downsample_ratio, variant, refiner, segmentation_pass, t)
        if taps is not None:
            taps.append(t)
        return out
    outs0, outs1 = [], []
    for ti in range(src.shape[1]):
        t = {} if taps is not None else None
        o = forward_frame(p, src[:, ti], rec, downsample_ratio, variant, refiner, segmentation_pass, t)
        if taps is not None:
            taps.append(t)
        rec = o[-4:]
        outs0.append(o[0])
        if not segmentation_pass:
            outs1.append(o[1])
    if segmentation_pass:
        return [torch.stack(outs0, 1), *rec]
    return [torch.stack(outs0, 1), torch.stack(outs1, 1), *rec]


# ------------------------------------------------------------------ driver output step (checker)
def compose_frames(fgr, pha, output_type="video", dtype=None):
    """inference.py:116,137-141 on [T,3,H,W] / [T,1,H,W] tensors of the run dtype: the composition the
    reference driver hands to its writer (green-screen video frame, or RGBA for png_sequence)."""
    dtype = dtype or fgr.dtype
    if output_type == "video":
        bgr = torch.tensor([120, 255, 155], dtype=dtype).div(255).view(1, 3, 1, 1)       # :116
        return fgr * pha + bgr * (1 - pha)                                               # :137
    fgr = fgr * pha.gt(0)                                                                # :139
    return torch.cat([fgr, pha], dim=-3)                                                 # :140


def frames_to_u8(frames):
    """inference_utils.py:42-44 (VideoWriter.write): gray -> RGB repeat, mul(255).byte(), NCHW -> NHWC.
    to_pil_image (ImageSequenceWriter, :82) applies the same mul(255).byte() to float tensors."""
    return frames.mul(255).byte().permute(0, 2, 3, 1).contiguous()
