"""Driver counterpart of the reference's ``inference.py`` (convert_video / Converter).

Same call surface as /root/reference/inference.py:25-38,155-171, with the steps that touch
tensors moved onto the device:

* the per-frame ``model(src, src, *rec, downsample_ratio)`` call is the HIP path
  (``MattingNetwork`` of this package, reference inference.py:129);
* composition, ``.mul(255).byte()`` and the NCHW->NHWC permute of the writers
  (inference.py:116,137-141, inference_utils.py:44) run as ONE kernel (``rvm_compose_u8``) that
  emits uint8 frames - a quarter of the bytes that cross PCIe compared with fp16 NCHW x 4 channels.

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

    def __init__(self, path, resize: Optional[Tuple[int, int]] = None):
        self.path = path
        self.files = sorted(os.listdir(path))
        self.resize = resize

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
        return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255)


class TensorSource:
    """In-memory clip [T,3,H,W] (float 0..1) with the reader interface."""

    def __init__(self, clip, frame_rate=30):
        self.clip = clip
        self.frame_rate = frame_rate

    def __len__(self):
        return self.clip.shape[0]

    def __getitem__(self, idx):
        return self.clip[idx]


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
        self.frames.extend(np.ascontiguousarray(f) for f in frames_u8)

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
    try:
        with torch.no_grad():
            rec = [None] * 4
            n = len(source)
            for start in range(0, n, seq_chunk):
                chunk = torch.stack([source[i] for i in range(start, min(start + seq_chunk, n))])     # [T,3,H,W]
                if downsample_ratio is None:
                    downsample_ratio = auto_downsample_ratio(*chunk.shape[2:])
                src = chunk.to(device, dtype, non_blocking=True).unsqueeze(0)                          # [1,T,3,H,W]
                # One output only and a refining ratio: the composition / byte conversion rides on the refiner's
                # full-resolution kernel (rvm_forward_u8) - half the bytes written, no second pass over fgr / pha.
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
                    wanted[0].write(frames.cpu().numpy())
                    if progress:
                        print("\r%d / %d frames" % (min(start + seq_chunk, n), n), end="", flush=True)
                    continue
                # the second src is the placeholder for the trimap (inference.py:128-129)
                fgr, pha, *rec = model(src, src, *rec, downsample_ratio)
                fgr, pha = fgr[0].contiguous(), pha[0].contiguous()
                if writer_fgr is not None:
                    writer_fgr.write(compose_u8(fgr, pha, native.COMPOSE_FOREGROUND).cpu().numpy())
                if writer_pha is not None:
                    a = compose_u8(None, pha, native.COMPOSE_ALPHA)
                    if output_type != 'video':
                        a = a[..., :1]                                  # to_pil_image of a 1-channel tensor: mode L
                    writer_pha.write(a.cpu().numpy())
                if writer_com is not None:
                    mode = native.COMPOSE_GREEN if output_type == 'video' else native.COMPOSE_RGBA
                    writer_com.write(compose_u8(fgr, pha, mode, bgr).cpu().numpy())
                if progress:
                    print("\r%d / %d frames" % (min(start + seq_chunk, n), n), end="", flush=True)
            if progress:
                print()
    finally:
        for w in (writer_com, writer_pha, writer_fgr):
            if w is not None:
                w.close()


class Converter:
    """Reference inference.py:162-171 without the TorchScript step (see the module docstring)."""

    def __init__(self, variant: str, checkpoint: str, device: str):
        self.model = MattingNetwork(variant).eval().to(device)
        self.model.load_state_dict(torch.load(checkpoint, map_location=device, weights_only=True))
        self.device = device

    def convert(self, *args, **kwargs):
        convert_video(self.model, *args, device=self.device, dtype=torch.float32, **kwargs)
