"""Drop-in ``MattingNetwork`` for MI355X.

Same constructor, ``forward`` signature, ``state_dict`` keys and recurrent-state
contract as the reference class (/root/reference/model/model.py:14-71), so

    from robustvideomatting_amd import MattingNetwork      # instead of: from model import MattingNetwork

is the only edit ``inference.py`` / ``hubconf.py`` style callers make.  The
module only *holds* the parameters; all arithmetic happens in
librvm_mi355x.so (hand-written gfx950 kernels) reached through the C ABI in
include/rvm_mi355x.h.  There is no PyTorch/CPU fallback: calling it with CPU
tensors, or without the native library, raises.
"""
import math
from typing import Optional

import torch
from torch import Tensor, nn

from . import native, spec

_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked")


class _Node(nn.Module):
    """Anonymous container; lets ``model.backbone.features[3].block[0][1].weight`` work."""

    def __getitem__(self, idx):
        return getattr(self, str(idx))

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container - compute happens in MattingNetwork.forward")


def _default_init(name, shape, gen, bn_prefixes):
    """Random init in the spirit of the reference's (torchvision kaiming_normal_ fan_out for
    the encoder, nn.Conv2d defaults elsewhere, identity BatchNorm)."""
    prefix, leaf = name.rsplit(".", 1)
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "running_mean":
        return torch.zeros(shape)
    if leaf == "running_var":
        return torch.ones(shape)
    if name == "refiner.box_filter.weight":
        return torch.full(shape, 1.0 / 9.0)                       # deep_guided_filter.py:13
    if prefix in bn_prefixes:
        return torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
    if len(shape) == 1:                                           # conv bias
        return (torch.rand(shape, generator=gen) * 2 - 1) * 0.05
    fan_in = shape[1] * shape[2] * shape[3]
    if name.startswith("backbone."):
        fan_out = shape[0] * shape[2] * shape[3]
        return torch.randn(shape, generator=gen) * math.sqrt(2.0 / fan_out)
    bound = 1.0 / math.sqrt(fan_in)
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


class MattingNetwork(nn.Module):
    def __init__(self,
                 variant: str = 'mobilenetv3',
                 refiner: str = 'deep_guided_filter',
                 pretrained_backbone: bool = False):
        super().__init__()
        assert variant in ['mobilenetv3', 'resnet50']
        assert refiner in ['fast_guided_filter', 'deep_guided_filter']
        if pretrained_backbone:
            raise RuntimeError("pretrained_backbone=True needs a network fetch "
                               "(reference mobilenetv3.py:29-31); load a state_dict instead")
        self.variant = variant
        self.refiner_kind = refiner
        shapes = spec.param_shapes(variant, refiner)
        bn_prefixes = {n[:-len(".running_mean")] for n in shapes if n.endswith(".running_mean")}
        gen = torch.Generator().manual_seed(0)
        self._names = list(shapes.keys())
        for name, shape in shapes.items():
            value = _default_init(name, shape, gen, bn_prefixes)
            node = self
            parts = name.split(".")
            for part in parts[:-1]:
                if not hasattr(node, part):
                    node.add_module(part, _Node())
                node = getattr(node, part)
            if parts[-1] in _BUFFER_LEAVES:
                node.register_buffer(parts[-1], value)
            else:
                node.register_parameter(parts[-1], nn.Parameter(value, requires_grad=False))
        self._master = None         # fp32 master copies (see _snapshot_masters)
        self._ctx = None            # native.Context
        self._ctx_key = None        # (device index, dtype)
        self._dirty = True          # parameters changed since the last upload
        self._graph = False
        self._pipe = False
        self.src_ready = None                       # frame pipeline: a torch.cuda.Event the next src is ordered behind (one-shot)
        self.fp16_arithmetic = "fp32"   # half() module: "fp32" = fp16 tensors, fp32 math inside (meets 2e-3);
        #                                 "fp16" = fp16 storage + fp16 MFMA inside (fastest, fp16 accuracy)
        self.chunk_batching = True  # [B,T,...] input: one native call per chunk (rvm_plan_seq) instead of T calls
        self._ring = {}             # plan key -> ring of static state buffers (graph mode)
        self._ring_pos = 0
        self._live = None           # live floating-point parameter / buffer tensors, for the in-place edit check
        self._live_sig = None       # (sum of their _version counters, sum of data_ptr) at the last upload
        self._calls = 0
        self.edit_check_interval = 64   # under no_grad, look for in-place parameter edits every N-th call (see _check_in_place_edits)

    # ---- bookkeeping: any .to()/.half()/.cuda()/load_state_dict invalidates the packed weights.
    # fp32 MASTER copies of the parameters are kept across .half(): the native packer folds
    # BatchNorm and rounds ONCE from them, instead of inheriting the rounding `.half()` applied to
    # every raw tensor (which alone moves pha by >2e-3 on the synthetic stress weights).
    def _snapshot_masters(self):
        w = self.project_mat.conv.weight
        if w.dtype == torch.float32:
            self._master = {k: v.detach().to("cpu", torch.float32).clone()
                            for k, v in self.state_dict().items() if v.is_floating_point()}

    def _apply(self, fn, *args, **kwargs):
        if getattr(self, "_names", None) is not None:
            self._snapshot_masters()
            self._dirty = True
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        res = super().load_state_dict(state_dict, *args, **kwargs)
        self._master = {k: v.detach().to("cpu", torch.float32).clone()
                        for k, v in state_dict.items() if torch.is_tensor(v) and v.is_floating_point()}
        self._dirty = True
        return res

    def refresh(self):
        """Call after modifying parameters in place: drops the fp32 masters and re-uploads."""
        self._master = None
        self._dirty = True

    def _signature(self):
        # the LIVE parameter / buffer objects (not state_dict aliases: those keep pointing at the old storage when a
        # parameter's .data is rebound): version counters catch in-place ops, data_ptr catches `p.data = ...`
        if self._live is None:
            self._live = [p for p in self.parameters()] + [b for b in self.buffers() if b.is_floating_point()]
        return (sum(v._version for v in self._live), sum(v.data_ptr() for v in self._live))

    def _check_in_place_edits(self):
        """In-place edits of the live parameters (an optimiser step, ``p.mul_()``, an EMA ``copy_``, ``p.data = ...``) make
        the packed weights stale: compare the tensors' version counters / storage with what was uploaded and, on a change,
        drop the fp32 masters and re-pack.  Checked on every call while autograd is enabled and on every
        ``edit_check_interval``-th call (default 64; 1 = every call, ~40 us of host time) under ``torch.no_grad()``, the
        inference loop - so up to 63 frames may still use the old weights; call ``refresh()`` right after an edit to
        re-pack at once.  Writes through ``p.data.copy_()`` bump no counter at all - ``refresh()`` is the only way."""
        object.__setattr__(self, "_calls", self._calls + 1)      # (nn.Module.__setattr__ costs ~2.5 us a write: this is the frame path)
        if self._dirty or self._live_sig is None:
            return
        if not torch.is_grad_enabled() and (self._calls % max(1, int(self.edit_check_interval))):
            return
        if self._signature() != self._live_sig:
            self._master = None
            self._dirty = True

    def enable_graph(self, on: bool = True):
        """Replay the per-frame schedule as a HIP graph (default: off, eager launches - equally fast for one
        stream).  The recurrent states then live in a ring of two static buffer sets per shape:
        a returned r1..r4 tensor is overwritten by the second-next call (inference.py hands them straight back,
        reference inference.py:129).  src may be a new tensor every call and fgr / pha are fresh tensors: the
        launches that touch them run outside the graph."""
        self._graph = bool(on)
        if self._ctx is not None:
            self._ctx.set_graph_mode(self._graph)
        return self

    def pipeline_frames(self, on: bool = True):
        """Frame pipeline (default: off).  The encoder phase of a frame (downsample, backbone, LR-ASPP:
        model/model.py:52-58) does not depend on the recurrent state; with the pipeline on it runs on a stream of the
        native context's own, beside the decoder / refiner phase (model/model.py:59-67) of the frame before on the
        current stream.  Same bits; fgr / pha / r1..r4 are ordered on the current stream as always.  For callers that
        send frames without waiting for each result (reference inference.py:122-141).  The contract for `src` changes:
        it is read on the context's stream, so it must be COMPLETE when forward is called (e.g. a frame uploaded and
        synchronised earlier, or produced on another stream) - or set `model.src_ready = event` (a recorded
        torch.cuda.Event, one-shot) before the call.  A src this wrapper has to copy first (non-contiguous) is ordered
        automatically.  With enable_graph() the decoder phase replays as a HIP graph as well (fewest host calls per frame)."""
        self._pipe = bool(on)
        if self._ctx is not None:
            self._ctx.set_frame_pipeline(self._pipe)
        return self

    def _order_src(self, ctx, dev, copied):
        if not self._pipe:
            return
        ev = self.src_ready
        if ev is not None:
            object.__setattr__(self, "src_ready", None)
        if copied:                                  # produced on the current stream a moment ago
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        object.__setattr__(self, "_src_ev_hold", ev)    # the native side takes the raw handle: keep the event alive through the call
        if ev is not None:
            ctx.src_event(ev.cuda_event)

    def _native(self, device, dtype):
        if dtype == torch.float32:
            code = native.DTYPE_F32
        elif dtype == torch.float16:
            if self.fp16_arithmetic not in ("fp32", "fp16"):
                raise RuntimeError("fp16_arithmetic must be 'fp32' or 'fp16'")
            code = native.DTYPE_F16IO if self.fp16_arithmetic == "fp32" else native.DTYPE_F16
        elif dtype == torch.bfloat16:
            code = native.DTYPE_BF16IO      # bf16 tensors at the boundary, fp32-accurate arithmetic inside
        else:
            raise RuntimeError("MattingNetwork (MI355X): unsupported dtype %s (float32 / float16 / bfloat16)" % dtype)
        key = (device.index if device.index is not None else torch.cuda.current_device(), code)
        if self._ctx is None or self._ctx_key != key:
            if self._ctx is not None:
                self._ctx.close()
            self._ctx = native.Context(self.variant, self.refiner_kind, code, key[0])
            self._ctx.set_graph_mode(self._graph)
            self._ctx.set_frame_pipeline(self._pipe)
            self._ctx_key = key
            self._dirty = True
            self._ring = {}
        if self._dirty:
            sd = self.state_dict()
            for name, shape in self._ctx.param_specs():
                if len(shape) == 0:
                    continue
                t = self._master.get(name) if self._master else None
                if t is None or tuple(t.shape) != tuple(sd[name].shape):
                    t = sd[name]
                t = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
                if tuple(t.shape) != shape:
                    raise RuntimeError("parameter %s has shape %s, expected %s" % (name, tuple(t.shape), shape))
                self._ctx.set_param(name, t)
            self._ctx.finalize()
            self._dirty = False
            self._ring = {}
            self._live = None
            self._live_sig = self._signature()
        return self._ctx

    # ---- the hot path
    def forward(self,
                src: Tensor,
                trimap: Optional[Tensor] = None,
                r1: Optional[Tensor] = None,
                r2: Optional[Tensor] = None,
                r3: Optional[Tensor] = None,
                r4: Optional[Tensor] = None,
                downsample_ratio: float = 1,
                segmentation_pass: bool = False):
        # Upstream RVM callers use forward(src, r1, r2, r3, r4, downsample_ratio); this fork
        # inserted `trimap` (reference model/model.py:42, never read).  Recognise the upstream
        # positional form so both keep working.
        if isinstance(r4, (int, float)) and not isinstance(r4, bool):
            trimap, r1, r2, r3, r4, downsample_ratio = None, trimap, r1, r2, r3, float(r4)
        elif (r4 is None and isinstance(trimap, Tensor) and trimap.ndim == 4 and src.ndim >= 4
              and trimap.shape[1] == spec.state_channels(self.variant)[0] and trimap.shape[1] != src.shape[-3]):
            trimap, r1, r2, r3, r4 = None, trimap, r1, r2, r3
        return self._run(src, [r1, r2, r3, r4], float(downsample_ratio), bool(segmentation_pass))

    def forward_upstream(self, src, r1=None, r2=None, r3=None, r4=None, downsample_ratio: float = 1,
                         segmentation_pass: bool = False):
        """Explicit upstream-RVM signature (README.md:184 of the reference)."""
        return self._run(src, [r1, r2, r3, r4], float(downsample_ratio), bool(segmentation_pass))

    def forward_u8(self, src, r1=None, r2=None, r3=None, r4=None, downsample_ratio: float = 1,
                   mode: int = native.COMPOSE_GREEN, bgr=None):
        """forward + the driver's output step in one pass (rvm_forward_u8): returns ``[frames_u8, r1, r2, r3, r4]`` with
        ``frames_u8`` uint8 ``[B,H,W,C]`` (``[B,T,H,W,C]`` for a 5-D src) - what the reference's writers get after
        convert_video's composition, ``.mul(255).byte()`` and the NHWC permute (inference.py:137-141, inference_utils.py:44).
        ``mode``: native.COMPOSE_GREEN (needs ``bgr``), COMPOSE_RGBA, COMPOSE_ALPHA, COMPOSE_FOREGROUND.  Needs
        ``downsample_ratio != 1`` (the composition rides on the refiner's full-resolution kernel)."""
        return self._run(src, [r1, r2, r3, r4], float(downsample_ratio), False, u8=(int(mode), bgr))

    def _run(self, src, rec, ds, seg, u8=None):
        if not isinstance(src, Tensor) or src.ndim not in (4, 5) or src.shape[-3] != 3:
            raise RuntimeError("src must be [B,3,H,W] or [B,T,3,H,W]")
        if not src.is_cuda:
            raise RuntimeError("MattingNetwork (MI355X) runs on HIP device tensors only; got a %s tensor. "
                               "There is no CPU fallback by design." % src.device.type)
        wdtype = self.project_mat.conv.weight.dtype
        if src.dtype != wdtype:
            raise RuntimeError("Input type (%s) and weight type (%s) should be the same" % (src.dtype, wdtype))
        self._check_in_place_edits()
        ctx = self._native(src.device, src.dtype)
        five = src.ndim == 5
        if five:
            B, T = src.shape[:2]
            frames = src.transpose(0, 1)            # [T,B,3,H,W]
        else:
            B, T = src.shape[0], 1
            frames = src.unsqueeze(0)
        H, W = src.shape[-2:]
        chunk = five and T > 1 and self.chunk_batching
        ctx.plan(B, H, W, ds, T if chunk else 1)
        shapes = ctx.state_shapes()
        dev, dt = src.device, src.dtype

        rec = list(rec)
        for i, r in enumerate(rec):
            if r is None:
                continue
            if not isinstance(r, Tensor):
                raise RuntimeError("r%d must be a Tensor or None" % (i + 1))
            if tuple(r.shape) == (1, 1, 1, 1) and float(r.abs().sum()) == 0.0:
                rec[i] = None                       # ONNX-style explicit zero state (documentation/inference.md:212)
                continue
            if tuple(r.shape) != shapes[i]:
                raise RuntimeError("Sizes of tensors must match: r%d is %s but this input needs %s "
                                   "(the recurrent state cannot follow a resolution change)"
                                   % (i + 1, tuple(r.shape), shapes[i]))
            if r.device != dev or r.dtype != dt:
                raise RuntimeError("r%d must be on %s with dtype %s" % (i + 1, dev, dt))
            rec[i] = r.contiguous(memory_format=torch.channels_last)

        h0, w0 = spec.lowres_size(H, W, ds)
        if u8 is not None:
            return self._run_u8(ctx, src, rec, shapes, B, T, H, W, five, chunk, u8)
        if chunk:
            return self._run_chunk(ctx, src, rec, shapes, B, T, H, W, h0, w0, seg)
        oshape0 = (B, 1, h0, w0) if seg else (B, 3, H, W)
        oshape1 = None if seg else (B, 1, H, W)
        out0 = torch.empty((T,) + oshape0, device=dev, dtype=dt)
        out1 = None if seg else torch.empty((T,) + oshape1, device=dev, dtype=dt)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ring = self._static_ring(ctx, shapes, dev, dt) if self._graph else None
        for t in range(T):
            f = frames[t]
            copied = not f.is_contiguous()
            if copied:
                f = f.contiguous()
            if ring is not None:
                # static state buffers => stable pointers => the captured HIP graph can be replayed
                new = self._ring_next(ring, rec)
            else:
                new = [torch.empty(s, device=dev, dtype=dt, memory_format=torch.channels_last) for s in shapes]
            o0, o1 = out0[t], (None if seg else out1[t])
            rin = [r.data_ptr() if r is not None else 0 for r in rec]
            self._order_src(ctx, dev, copied)
            if seg:
                ctx.forward(f.data_ptr(), rin, 0, o0.data_ptr(), [n.data_ptr() for n in new], True, stream)
            else:
                ctx.forward(f.data_ptr(), rin, o0.data_ptr(), o1.data_ptr(), [n.data_ptr() for n in new], False, stream)
            rec = new
        if five:
            outs = [out0.transpose(0, 1)] if seg else [out0.transpose(0, 1), out1.transpose(0, 1)]
        else:
            outs = [out0[0]] if seg else [out0[0], out1[0]]
        return [*outs, *rec]

    def _run_chunk(self, ctx, src, rec, shapes, B, T, H, W, h0, w0, seg):
        """[B,T,3,H,W] in one native call: like the reference's time-series path (decoder.py:92-105,174-180)
        every stage outside the ConvGRUs runs on all B*T frames at once; the native side wants time-major frames."""
        dev, dt = src.device, src.dtype
        tb = src.transpose(0, 1)                    # [T,B,3,H,W]; already contiguous when B == 1
        copied = not tb.is_contiguous()
        if copied:
            tb = tb.contiguous()
        self._order_src(ctx, dev, copied)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rin = [r.data_ptr() if r is not None else 0 for r in rec]
        if self._graph:
            ring = self._static_ring(ctx, shapes, dev, dt)
            new = self._ring_next(ring, rec)
        else:
            new = [torch.empty(s, device=dev, dtype=dt, memory_format=torch.channels_last) for s in shapes]
        o0 = torch.empty((T * B, 1, h0, w0) if seg else (T * B, 3, H, W), device=dev, dtype=dt)
        o1 = None if seg else torch.empty((T * B, 1, H, W), device=dev, dtype=dt)
        if seg:
            ctx.forward(tb.data_ptr(), rin, 0, o0.data_ptr(), [n.data_ptr() for n in new], True, stream)
            outs = [o0.view(T, B, 1, h0, w0).transpose(0, 1)]
        else:
            ctx.forward(tb.data_ptr(), rin, o0.data_ptr(), o1.data_ptr(), [n.data_ptr() for n in new], False, stream)
            outs = [o0.view(T, B, 3, H, W).transpose(0, 1), o1.view(T, B, 1, H, W).transpose(0, 1)]
        return [*outs, *new]

    def _run_u8(self, ctx, src, rec, shapes, B, T, H, W, five, chunk, u8):
        mode, bgr = u8
        if mode == native.COMPOSE_GREEN and bgr is None:
            raise RuntimeError("forward_u8: COMPOSE_GREEN needs the background colour `bgr`")
        dev, dt = src.device, src.dtype
        C = 4 if mode == native.COMPOSE_RGBA else 3
        stream = torch.cuda.current_stream(dev).cuda_stream
        tb = src.transpose(0, 1) if five else src.unsqueeze(0)         # [T,B,3,H,W]
        copied = not tb.is_contiguous()
        if copied:
            tb = tb.contiguous()
        out = torch.empty((T, B, H, W, C), device=dev, dtype=torch.uint8)
        steps = [(0, T)] if (chunk or T == 1) else [(t, 1) for t in range(T)]
        for t0, _ in steps:
            self._order_src(ctx, dev, copied)
            new = self._ring_next(self._static_ring(ctx, shapes, dev, dt), rec) if self._graph else \
                [torch.empty(s, device=dev, dtype=dt, memory_format=torch.channels_last) for s in shapes]
            ctx.forward_u8(tb[t0].data_ptr(), [r.data_ptr() if r is not None else 0 for r in rec], out[t0].data_ptr(),
                           [n.data_ptr() for n in new], mode, bgr, stream)
            rec = new
        frames = out.transpose(0, 1) if five else out[0]
        return [frames, *rec]

    def _static_ring(self, ctx, shapes, dev, dt):
        key = (ctx.plan_key, dev, dt)
        ring = self._ring.get(key)
        if ring is None:
            ring = [[torch.zeros(s, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last) for s in shapes]
                    for _ in range(2)]
            self._ring = {key: ring}
            self._ring_pos = 0
        return ring

    def _ring_next(self, ring, rec):
        """Next state buffer set of the ring that is not the one the incoming states live in."""
        ins = {r.data_ptr() for r in rec if r is not None}
        for _ in range(len(ring)):
            new = ring[self._ring_pos]
            self._ring_pos = (self._ring_pos + 1) % len(ring)
            if not any(n.data_ptr() in ins for n in new):
                return new
        raise RuntimeError("graph mode: the incoming states occupy every buffer set of the ring")

    def debug_tap(self, name):
        """fp32 NCHW copy of an intermediate activation of the last frame (tests only)."""
        return self._ctx.debug_tap(name)

    def launch_count(self):
        return self._ctx.launch_count()
