"""Driver counterpart of the reference's ``inference.py`` (convert_video / Converter).

Same call surface as /root/reference/inference.py:25-38,155-171, with the steps that touch
tensors moved onto the device:

* the per-frame ``model(src, src, *rec, downsample_ratio)`` call is the HIP path
  (``MattingNetwork`` of this package, reference inference.py:129);
* composition, ``.mul(255).byte()`` and the NCHW->NHWC permute of the writers
  (inference.py:116,137-141, inference_utils.py:44) run as ONE kernel (``rvm_compose_u8``) that
  emits uint8 frames - a quarter of the bytes that cross PCIe compared with fp16 NCHW x 4 channels.

Host <-> device traffic is pipelined like the reference's (pinned source tensors, ``non_blocking`` copies, inference.py:79,127):
the next chunk is uploaded on a side stream while the current one is computed, the finished uint8 frames come back on a third
stream into pinned buffers and are handed to the writer one chunk later.  Sources that yield uint8 frames (``yields_uint8``)
are uploaded as bytes and divided by 255 on the device - a quarter of the fp32 upload, same values bit for bit.

Codecs are out of scope (SURVEY.md component 9): ``av`` / ``pims`` are not in this image, so a video
FILE source or sink raises a clear error; image-sequence directories (PIL) and in-memory sources /
sinks work.  ``torch.jit.script`` + ``freeze`` of the reference's ``Converter`` is skipped: BatchNorm
folding and launch trimming are done by the native packer and the HIP graph.
"""
import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import native
from .model import MattingNetwork


def auto_downsample_ratio(h, w):
    """Largest side becomes 512 px (reference inference.py:155-159)."""
    return min(512 / max(h, w), 1)


# ------------------------------------------------------------------------------------------ sources / sinks
class ImageSequenceReader:
    """Sorted png/jpg files of a directory -> float tensors [3,H,W] in 0..1 (inference_utils.py:55-69 +
    torchvision ToTensor: uint8 / 255)."""

    frame_rate = 30

    def __init__(self, path, resize: Optional[Tuple[int, int]] = None, as_uint8: bool = False):
        self.path = path
        self.files = sorted(os.listdir(path))
        self.resize = resize
        self.yields_uint8 = as_uint8       # frames as uint8 [3,H,W]; convert_video divides by 255 on the device

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        from PIL import Image
        with Image.open(os.path.join(self.path, self.files[idx])) as img:
            img.load()
            img = img.convert("RGB")
        if self.resize is not None:
            img = img.resize(self.resize, Image.BILINEAR)        # transforms.Resize default interpolation
        arr = np.asarray(img, dtype=np.uint8)
        t = torch.from_numpy(arr.copy()).permute(2, 0, 1)
        return t if self.yields_uint8 else t.float().div(255)


class TensorSource:
    """In-memory clip [T,3,H,W] (float 0..1, or uint8 0..255) with the reader interface."""

    def __init__(self, clip, frame_rate=30):
        self.clip = clip
        self.frame_rate = frame_rate
        self.yields_uint8 = clip.dtype == torch.uint8

    def __len__(self):
        return self.clip.shape[0]

    def __getitem__(self, idx):
        return self.clip[idx]

    def pin(self):
        """Move the clip to page-locked host memory: convert_video then uploads its chunks straight from it, without the
        staging copy (which at ~15 GB/s of host memcpy is what bounds a 4K stream otherwise)."""
        self.clip = self.clip.contiguous().pin_memory()
        return self

    def chunk(self, start, stop):
        """Frames [start, stop) as ONE tensor (a view): lets the pipeline skip the per-frame staging copy when it is pinned."""
        return self.clip[start:stop]


class ImageSequenceWriter:
    """uint8 frames [T,H,W,C] -> numbered image files (inference_utils.py:72-87)."""

    def __init__(self, path, extension="png"):
        self.path = path
        self.extension = extension
        self.counter = 0
        os.makedirs(path, exist_ok=True)

    def write(self, frames_u8):
        from PIL import Image
        for t in range(frames_u8.shape[0]):
            a = frames_u8[t]
            img = Image.fromarray(a[..., 0] if a.shape[-1] == 1 else a)
            img.save(os.path.join(self.path, str(self.counter).zfill(4) + "." + self.extension))
            self.counter += 1

    def close(self):
        pass


class ListWriter:
    """Collects uint8 frames in memory (tests, callers with their own encoder)."""

    def __init__(self):
        self.frames = []

    def write(self, frames_u8):
        self.frames.extend(np.array(f, copy=True) for f in frames_u8)      # the array handed in is a staging buffer

    def close(self):
        pass


def _video_unavailable(what):
    raise RuntimeError("%s needs PyAV/pims (reference inference_utils.py:10-52), which this image does not have; "
                       "use an image-sequence directory, or pass a reader / writer object" % what)


# ------------------------------------------------------------------------------------------ device output step
def compose_u8(fgr, pha, mode, bgr=None):
    """fgr [T,3,H,W], pha [T,1,H,W] device tensors -> uint8 [T,H,W,C] device tensor (rvm_compose_u8)."""
    assert pha.is_cuda and pha.is_contiguous() and pha.dtype in (torch.float32, torch.float16, torch.bfloat16)
    T, _, H, W = pha.shape
    C = 4 if mode == native.COMPOSE_RGBA else 3
    out = torch.empty((T, H, W, C), device=pha.device, dtype=torch.uint8)
    code = {torch.float32: native.DTYPE_F32, torch.float16: native.DTYPE_F16IO, torch.bfloat16: native.DTYPE_BF16IO}[pha.dtype]
    fptr = 0
    if mode != native.COMPOSE_ALPHA:
        assert fgr.is_contiguous() and fgr.dtype == pha.dtype and fgr.shape == (T, 3, H, W)
        fptr = fgr.data_ptr()
    stream = torch.cuda.current_stream(pha.device).cuda_stream
    native.compose_u8(code, mode, fptr, pha.data_ptr(), out.data_ptr(), T, H, W, bgr, stream)
    return out


def convert_video(model,
                  input_source,
                  input_resize: Optional[Tuple[int, int]] = None,
                  downsample_ratio: Optional[float] = None,
                  output_type: str = 'video',
                  output_composition=None,
                  output_alpha=None,
                  output_foreground=None,
                  output_video_mbps: Optional[float] = None,
                  seq_chunk: int = 1,
                  num_workers: int = 0,
                  progress: bool = True,
                  device: Optional[str] = None,
                  dtype: Optional[torch.dtype] = None):
    """Reference signature (inference.py:25-38).  ``input_source``: image-sequence directory or a reader
    object (``__len__`` / ``__getitem__`` -> [3,H,W] float 0..1); outputs: directory paths
    (``output_type='png_sequence'``) or writer objects with ``write(uint8[T,H,W,C])`` / ``close()``."""
    assert downsample_ratio is None or (downsample_ratio > 0 and downsample_ratio <= 1), \
        'Downsample ratio must be between 0 (exclusive) and 1 (inclusive).'
    assert any([output_composition, output_alpha, output_foreground]), 'Must provide at least one output.'
    assert output_type in ['video', 'png_sequence'], 'Only support "video" and "png_sequence" output modes.'
    assert seq_chunk >= 1, 'Sequence chunk must be >= 1'
    assert num_workers >= 0, 'Number of workers must be >= 0'

    if isinstance(input_source, str):
        if os.path.isfile(input_source):
            _video_unavailable("reading the video file %r" % input_source)
        source = ImageSequenceReader(input_source, input_resize)
    else:
        source = input_source

    def writer_for(target):
        if target is None or not isinstance(target, str):
            return target
        if output_type == 'video':
            _video_unavailable("writing the video file %r" % target)
        return ImageSequenceWriter(target, 'png')

    writer_com, writer_pha, writer_fgr = writer_for(output_composition), writer_for(output_alpha), writer_for(output_foreground)

    model = model.eval()
    if device is None or dtype is None:
        param = next(model.parameters())
        dtype = param.dtype
        device = param.device
    bgr = None
    if writer_com is not None and output_type == 'video':
        # the reference's constant, rounded like its tensor (inference.py:116)
        bgr = [float(v) for v in torch.tensor([120, 255, 155], dtype=dtype).div(255)]
    dev = torch.device(device)
    pipe = _pipeline_for(dev) if dev.type == "cuda" else None
    u8src = bool(getattr(source, "yields_uint8", False))
    try:
        with torch.no_grad():
            rec = [None] * 4
            n = len(source)
            pending = []                                     # earlier chunks' outputs, still on their way to the host (oldest first)
            for k, start in enumerate(range(0, n, seq_chunk)):
                stop = min(start + seq_chunk, n)
                whole = source.chunk(start, stop) if hasattr(source, "chunk") else None
                frames_in = whole if (whole is not None and pipe is not None and whole.is_pinned() and whole.is_contiguous()) \
                    else [source[i] for i in range(start, stop)]
                if downsample_ratio is None:
                    downsample_ratio = auto_downsample_ratio(*frames_in[0].shape[1:])
                if pipe is not None:
                    src = pipe.upload(k, frames_in, dtype, u8src).unsqueeze(0)                     # [1,T,3,H,W]
                else:
                    chunk = torch.stack(frames_in)
                    chunk = chunk.float().div(255) if u8src else chunk
                    src = chunk.to(device, dtype, non_blocking=True).unsqueeze(0)
                # One output only and a refining ratio: the composition / byte conversion rides on the refiner's
                # full-resolution kernel (rvm_forward_u8) - half the bytes written, no second pass over fgr / pha.
                outs = []                                     # (writer, uint8 device frames [T,H,W,C])
                wanted = [w for w in (writer_com, writer_pha, writer_fgr) if w is not None]
                if len(wanted) == 1 and downsample_ratio != 1 and hasattr(model, "forward_u8"):
                    if writer_com is not None:
                        mode = native.COMPOSE_GREEN if output_type == 'video' else native.COMPOSE_RGBA
                    else:
                        mode = native.COMPOSE_ALPHA if writer_pha is not None else native.COMPOSE_FOREGROUND
                    frames, *rec = model.forward_u8(src, *rec, downsample_ratio=downsample_ratio, mode=mode, bgr=bgr)
                    frames = frames[0]
                    if writer_pha is not None and output_type != 'video':
                        frames = frames[..., :1]                        # to_pil_image of a 1-channel tensor: mode L
                    outs.append((wanted[0], frames))
                else:
                    # the second src is the placeholder for the trimap (inference.py:128-129)
                    fgr, pha, *rec = model(src, src, *rec, downsample_ratio)
                    fgr, pha = fgr[0].contiguous(), pha[0].contiguous()
                    if writer_fgr is not None:
                        outs.append((writer_fgr, compose_u8(fgr, pha, native.COMPOSE_FOREGROUND)))
                    if writer_pha is not None:
                        a = compose_u8(None, pha, native.COMPOSE_ALPHA)
                        if output_type != 'video':
                            a = a[..., :1]                              # to_pil_image of a 1-channel tensor: mode L
                        outs.append((writer_pha, a))
                    if writer_com is not None:
                        mode = native.COMPOSE_GREEN if output_type == 'video' else native.COMPOSE_RGBA
                        outs.append((writer_com, compose_u8(fgr, pha, mode, bgr)))
                # the frames start their way back now
                if pipe is not None:
                    pending.append(pipe.download(k, outs))
                    while len(pending) >= pipe.DEPTH:           # chunk k - DEPTH + 1 (long arrived) goes to the writers
                        pipe.deliver(pending.pop(0))
                else:
                    for w, fr in outs:
                        w.write(fr.cpu().numpy())
                if progress:
                    print("\r%d / %d frames" % (min(start + seq_chunk, n), n), end="", flush=True)
            for p in pending:
                pipe.deliver(p)
            if progress:
                print()
    finally:
        for w in (writer_com, writer_pha, writer_fgr):
            if w is not None:
                w.close()


class _Pipeline:
    """Host <-> device staging for convert_video: pinned buffers, an upload stream and a download stream beside the compute
    stream (the reference pins and copies ``non_blocking``, inference.py:79,127).  DEPTH buffer sets: set k % DEPTH serves
    chunk k and is reused only after the copies that touched it have completed (events); the frames of chunk k go to the
    writers while chunks k + 1 and k + 2 are on their way, so uploads and downloads run back to back and the loop's period
    is the slowest of upload, compute and download rather than their partial sums."""

    DEPTH = 3

    def __init__(self, device):
        self.device = device
        self.up = torch.cuda.Stream(device)
        self.down = torch.cuda.Stream(device)
        self.host_in = [None] * self.DEPTH
        self.up_done = [None] * self.DEPTH
        self.host_out = [{} for _ in range(self.DEPTH)]
        self.c255 = torch.full((), 255.0, device=device)

    def upload(self, k, frames, dtype, u8):
        """frames: list of [3,H,W] host tensors (staged through a pinned buffer), or ONE pinned [T,3,H,W] tensor (copied from
        where it lies; the caller keeps it alive and unchanged until the chunk has been delivered)."""
        j = k % self.DEPTH
        cur = torch.cuda.current_stream(self.device)
        if torch.is_tensor(frames):
            with torch.cuda.stream(self.up):
                d = frames.to(self.device, non_blocking=True)
                d = torch.div(d.float(), self.c255).to(dtype) if u8 else d.to(dtype)
                ev = self.up.record_event()
            cur.wait_event(ev)
            d.record_stream(cur)
            return d
        T, shape = len(frames), tuple(frames[0].shape)
        buf = self.host_in[j]
        if buf is None or buf.shape[1:] != shape or buf.shape[0] < T or buf.dtype != frames[0].dtype:
            buf = self.host_in[j] = torch.empty((T,) + shape, dtype=frames[0].dtype).pin_memory()
        if self.up_done[j] is not None:
            self.up_done[j].synchronize()                   # the copy that last read this buffer
        for t, f in enumerate(frames):
            buf[t].copy_(f)
        with torch.cuda.stream(self.up):
            d = buf[:T].to(self.device, non_blocking=True)
            # uint8 frames: ToTensor's `/ 255` in fp32, then the model dtype (reference inference.py:66-70,127) - on the device
            # (a tensor divisor: with a Python scalar torch multiplies by the rounded reciprocal on the GPU, one ulp off the CPU's division)
            d = torch.div(d.float(), self.c255).to(dtype) if u8 else d.to(dtype)
            self.up_done[j] = self.up.record_event()
        cur.wait_event(self.up_done[j])
        d.record_stream(cur)
        return d

    def download(self, k, outs):
        j = k % self.DEPTH
        cur = torch.cuda.current_stream(self.device)
        done = cur.record_event()
        ticket = []
        with torch.cuda.stream(self.down):
            self.down.wait_event(done)
            for idx, (w, fr) in enumerate(outs):
                key = (idx, tuple(fr.shape[1:]))
                host = self.host_out[j].get(key)
                if host is None or host.shape[0] < fr.shape[0]:
                    host = self.host_out[j][key] = torch.empty(fr.shape, dtype=torch.uint8).pin_memory()
                host[:fr.shape[0]].copy_(fr, non_blocking=True)
                fr.record_stream(self.down)
                ticket.append((w, host, fr.shape[0]))
            ev = self.down.record_event()
        return ev, ticket

    def deliver(self, pending):
        ev, ticket = pending
        ev.synchronize()
        for w, host, T in ticket:
            w.write(host[:T].numpy())


_PIPELINES = {}


def _pipeline_for(device):
    """One staging pipeline per device for the life of the process: page-locking its buffers costs ~10 ms per 100 MB, more
    than a short clip's whole conversion (release_staging() drops them)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _PIPELINES:
        _PIPELINES[key] = _Pipeline(device)
    return _PIPELINES[key]


def release_staging():
    """Free the pinned staging buffers convert_video keeps between calls."""
    _PIPELINES.clear()


class Converter:
    """Reference inference.py:162-171 without the TorchScript step (see the module docstring)."""

    def __init__(self, variant: str, checkpoint: str, device: str):
        self.model = MattingNetwork(variant).eval().to(device)
        self.model.load_state_dict(torch.load(checkpoint, map_location=device, weights_only=True))
        self.device = device

    def convert(self, *args, **kwargs):
        convert_video(self.model, *args, device=self.device, dtype=torch.float32, **kwargs)
