"""Seeded synthetic weights and clips.

The trained checkpoints are not available offline (``.MISSING_LARGE_BLOBS``),
so parity and timing both use weights drawn here.  The draw is shaped so that
BN folding is actually exercised (non-trivial running stats) and activations
stay O(1) through the whole stack - otherwise saturated gates / clamps would
hide kernel errors.
"""
import math

import torch

from . import spec


def _conv_gain(name):
    """Per-family gains, tuned so every tap of the oracle has std in ~[0.3, 3]
    (tests/test_oracle_golden.py::test_activation_ranges keeps it that way)."""
    if ".gru." in name:
        return 1.0
    if name.startswith("project_"):
        return 1.0
    if name.startswith("refiner.conv.6"):
        return 0.5
    if name.startswith("refiner."):
        return 1.0
    if ".fc1." in name or ".fc2." in name or "aspp2" in name:
        return 1.0
    if name.startswith("decoder."):
        return 1.5
    if name.startswith("aspp."):
        return 3.0
    if name.startswith("backbone.layer"):
        if ".conv3." in name:
            return 0.45
        if ".downsample." in name:
            return 0.8
        return 1.3
    return 1.2


def make_params(variant="mobilenetv3", refiner="deep_guided_filter", seed=0,
                perturb_box_filter=False):
    """Return an OrderedDict name -> fp32 CPU tensor with the reference's keys."""
    g = torch.Generator().manual_seed(seed)
    shapes = spec.param_shapes(variant, refiner)
    out = type(shapes)()

    def randn(shape, std=1.0):
        return torch.randn(shape, generator=g) * std

    def rand(shape, lo, hi):
        return torch.rand(shape, generator=g) * (hi - lo) + lo

    bn_prefixes = {n[:-len(".running_mean")] for n in shapes if n.endswith(".running_mean")}

    for name, shape in shapes.items():
        prefix, leaf = name.rsplit(".", 1)
        if leaf == "num_batches_tracked":
            out[name] = torch.tensor(100, dtype=torch.long)
        elif leaf == "running_mean":
            out[name] = randn(shape, 0.1)
        elif leaf == "running_var":
            out[name] = rand(shape, 0.5, 1.5)
        elif leaf == "weight" and prefix in bn_prefixes:
            out[name] = rand(shape, 0.5, 1.5)            # BN gamma
        elif name == "project_mat.conv.bias":
            out[name] = torch.tensor([0.0, 0.0, 0.0, 0.5]) + randn(shape, 0.05)   # centre pha in (0,1)
        elif leaf == "bias":
            out[name] = randn(shape, 0.1)                # BN beta and conv biases
        elif name == "refiner.box_filter.weight":
            w = torch.full(shape, 1.0 / 9.0)
            if perturb_box_filter:
                w = w + randn(shape, 0.01)
            out[name] = w
        else:
            # convolution weights: variance-preserving draw
            fan_in = shape[1] * shape[2] * shape[3]
            gain = _conv_gain(name)
            out[name] = randn(shape, gain / math.sqrt(fan_in))
            if name == "project_mat.conv.weight":
                out[name][:3] *= 0.3                     # keep fgr = src + residual mostly inside (0,1)
                out[name][3] *= 0.5                      # keep pha inside (0,1): the clamp must not hide errors
    return out


def clip_frames(T, H, W, seed=0, B=1, noise=0.04, _gen=None):
    """Generator over the frames [3,H,W] of stream 0 of make_clip(T, H, W, seed) (same values, same RNG order)
    without holding the clip: a 300-frame HD clip is 7.5 GB as one fp32 tensor."""
    g = _gen if _gen is not None else torch.Generator().manual_seed(1000 + seed)
    ys = torch.linspace(0, 1, H).view(H, 1)
    xs = torch.linspace(0, 1, W).view(1, W)
    b = 0 if _gen is None else B
    # low-frequency background: a few random sinusoids per channel
    bg = torch.zeros(3, H, W)
    for c in range(3):
        acc = torch.zeros(H, W)
        for _ in range(4):
            fy, fx = (torch.rand(2, generator=g) * 3.0 + 0.3).tolist()
            ph = (torch.rand(1, generator=g) * 6.28).item()
            amp = (torch.rand(1, generator=g) * 0.15 + 0.05).item()
            acc = acc + amp * torch.sin(6.28 * (fy * ys + fx * xs) + ph)
        bg[c] = 0.5 + acc
    col = torch.rand(3, generator=g) * 0.6 + 0.2
    cx0, cy0 = (torch.rand(2, generator=g) * 0.4 + 0.3).tolist()
    for t in range(T):
        cx = cx0 + 0.15 * math.sin(0.07 * t + b)
        cy = cy0 + 0.10 * math.cos(0.05 * t + 2 * b)
        d2 = ((xs - cx) / 0.18) ** 2 + ((ys - cy) / 0.30) ** 2
        m = torch.sigmoid((1.0 - d2) * 6.0)
        fr = bg * (1 - m) + col.view(3, 1, 1) * m
        fr = fr + (torch.rand(3, H, W, generator=g) - 0.5) * noise
        yield fr.clamp(0, 1)


def make_clip(T, H, W, seed=0, B=1, dtype=torch.float32, noise=0.04):
    """[B,T,3,H,W] clip in [0,1]: smooth static background + moving soft ellipse + noise."""
    g = torch.Generator().manual_seed(1000 + seed)
    clips = [torch.stack(list(clip_frames(T, H, W, seed, b, noise, _gen=g))) for b in range(B)]
    return torch.stack(clips).to(dtype)
