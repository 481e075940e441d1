"""Driver counterpart (robustvideomatting_amd.inference) of the reference's inference.py.

CPU: argument contract, readers / writers.  GPU: the fused composition + uint8 kernel must give the
SAME BYTES as the reference's torch ops (oracle.compose_frames / frames_to_u8 restate
inference.py:116,137-141 and inference_utils.py:44), and convert_video end to end on a PNG sequence
must match the oracle pipeline to 1 LSB.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rvm_oracle as O                                     # noqa: E402  (checker)
from robustvideomatting_amd import inference as I, synthetic          # noqa: E402


def test_auto_downsample_ratio():
    assert I.auto_downsample_ratio(1080, 1920) == 512 / 1920           # inference.py:155-159
    assert I.auto_downsample_ratio(288, 512) == 1
    assert I.auto_downsample_ratio(2160, 3840) == 512 / 3840


def test_convert_video_argument_contract(tmp_path):
    src = I.TensorSource(torch.rand(2, 3, 32, 48))
    with pytest.raises(AssertionError, match="Downsample ratio"):
        I.convert_video(None, src, downsample_ratio=1.5, output_alpha=I.ListWriter())
    with pytest.raises(AssertionError, match="at least one output"):
        I.convert_video(None, src)
    with pytest.raises(AssertionError, match="output modes"):
        I.convert_video(None, src, output_type="gif", output_alpha=I.ListWriter())
    with pytest.raises(AssertionError, match="Sequence chunk"):
        I.convert_video(None, src, seq_chunk=0, output_alpha=I.ListWriter())
    f = tmp_path / "clip.mp4"
    f.write_bytes(b"x")
    with pytest.raises(RuntimeError, match="PyAV"):
        I.convert_video(None, str(f), output_alpha=I.ListWriter())


def test_image_sequence_reader_writer_roundtrip(tmp_path):
    frames = (np.random.RandomState(0).rand(3, 20, 30, 3) * 255).astype(np.uint8)
    w = I.ImageSequenceWriter(str(tmp_path / "seq"), "png")
    w.write(frames)
    w.close()
    r = I.ImageSequenceReader(str(tmp_path / "seq"))
    assert len(r) == 3 and sorted(os.listdir(tmp_path / "seq"))[0] == "0000.png"
    for t in range(3):
        x = r[t]
        assert x.shape == (3, 20, 30) and x.dtype == torch.float32
        assert torch.equal(x.mul(255).round().byte(), torch.from_numpy(frames[t]).permute(2, 0, 1))


def test_oracle_frames_to_u8_truncates():
    x = torch.tensor([0.0, 0.5, 0.999, 1.0]).view(1, 1, 1, 4)
    assert O.frames_to_u8(x).flatten().tolist() == [0, 127, 254, 255]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_compose_kernel_bytes_equal_reference_ops(dtype):
    from robustvideomatting_amd import native
    g = torch.Generator().manual_seed(3)
    fgr = torch.rand(2, 3, 37, 53, generator=g).to(dtype)
    pha = torch.rand(2, 1, 37, 53, generator=g)
    pha[pha < 0.2] = 0.0                                                # exercise pha.gt(0)
    pha[pha > 0.9] = 1.0
    pha = pha.to(dtype)
    fd, pd = fgr.cuda(), pha.cuda()
    bgr = [float(v) for v in torch.tensor([120, 255, 155], dtype=dtype).div(255)]
    got = I.compose_u8(fd, pd, native.COMPOSE_GREEN, bgr).cpu()
    assert torch.equal(got, O.frames_to_u8(O.compose_frames(fgr, pha, "video")))
    got = I.compose_u8(fd, pd, native.COMPOSE_RGBA).cpu()
    assert torch.equal(got, O.frames_to_u8(O.compose_frames(fgr, pha, "png_sequence")))
    got = I.compose_u8(None, pd, native.COMPOSE_ALPHA).cpu()
    assert torch.equal(got, O.frames_to_u8(pha.repeat(1, 3, 1, 1)))    # inference_utils.py:42-43
    got = I.compose_u8(fd, pd, native.COMPOSE_FOREGROUND).cpu()
    assert torch.equal(got, O.frames_to_u8(fgr))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,W", [(72, 104), (61, 99)])          # W % 8 == 0 (vector stores) and an odd width (tail path)
def test_forward_u8_bytes_equal_forward_then_compose(dtype, H, W):
    """rvm_forward_u8 (composition + byte conversion inside the refiner's full-resolution kernel): the bytes
    equal the reference's output ops applied to what forward returns, in every mode, frame by frame and as a chunk."""
    from robustvideomatting_amd import MattingNetwork, native
    params = synthetic.make_params("mobilenetv3", "deep_guided_filter", seed=2)
    m = MattingNetwork().eval()
    m.load_state_dict(params)
    m = m.to("cuda", dtype)
    clip = synthetic.make_clip(3, H, W, seed=8, B=2).to("cuda", dtype)
    bgr = [float(v) for v in torch.tensor([120, 255, 155], dtype=dtype).div(255)]
    with torch.no_grad():
        fgr, pha, *rec = m(clip, downsample_ratio=0.5)
        for mode, kind in ((native.COMPOSE_GREEN, "video"), (native.COMPOSE_RGBA, "png_sequence"),
                           (native.COMPOSE_ALPHA, None), (native.COMPOSE_FOREGROUND, None)):
            got, *rec8 = m.forward_u8(clip, downsample_ratio=0.5, mode=mode, bgr=bgr)
            assert got.dtype == torch.uint8 and got.shape[:4] == (2, 3, H, W)
            f, p = fgr.cpu(), pha.cpu()
            if kind is not None:
                want = O.frames_to_u8(O.compose_frames(f.flatten(0, 1), p.flatten(0, 1), kind))
            elif mode == native.COMPOSE_ALPHA:
                want = O.frames_to_u8(p.flatten(0, 1).repeat(1, 3, 1, 1))
            else:
                want = O.frames_to_u8(f.flatten(0, 1))
            assert torch.equal(got.cpu().flatten(0, 1), want), mode
            for a, b in zip(rec, rec8):
                assert torch.equal(a, b)
        # frame by frame with carried states
        r = [None] * 4
        for t in range(3):
            got, *r = m.forward_u8(clip[:, t], *r, downsample_ratio=0.5, mode=native.COMPOSE_GREEN, bgr=bgr)
            want = O.frames_to_u8(O.compose_frames(fgr[:, t].cpu(), pha[:, t].cpu(), "video"))
            assert got.shape == (2, H, W, 3)
            assert (got.cpu().int() - want.int()).abs().max() <= 1           # chunked vs framewise tiles: rounding-level
        with pytest.raises(RuntimeError, match="downsample_ratio == 1"):
            m.forward_u8(clip[:, 0], downsample_ratio=1.0, mode=native.COMPOSE_ALPHA)


@pytest.mark.gpu
def test_convert_video_png_sequence_matches_oracle_pipeline(tmp_path):
    from robustvideomatting_amd import MattingNetwork
    T, H, W = 4, 64, 96
    clip = synthetic.make_clip(T, H, W, seed=5)[0]                       # [T,3,H,W]
    src_dir = tmp_path / "in"
    I.ImageSequenceWriter(str(src_dir), "png").write(O.frames_to_u8(clip).numpy())
    params = synthetic.make_params("mobilenetv3", "deep_guided_filter", seed=1)
    model = MattingNetwork("mobilenetv3").eval()
    model.load_state_dict(params)
    model = model.cuda()
    com, alpha, fg = I.ListWriter(), I.ListWriter(), I.ListWriter()
    I.convert_video(model, str(src_dir), downsample_ratio=0.5, output_type="png_sequence",
                    output_composition=str(tmp_path / "com"), output_alpha=alpha, output_foreground=fg,
                    seq_chunk=3, progress=False)
    I.convert_video(model, str(src_dir), downsample_ratio=0.5, output_type="video", output_composition=com,
                    seq_chunk=2, progress=False)
    assert sorted(os.listdir(tmp_path / "com")) == ["%04d.png" % i for i in range(T)]
    # oracle pipeline on what the reader delivers
    rd = I.ImageSequenceReader(str(src_dir))
    frames = torch.stack([rd[i] for i in range(T)]).unsqueeze(0)
    out = O.forward(params, frames, downsample_ratio=0.5)
    fgr, pha = out[0][0], out[1][0]
    ref_com = O.frames_to_u8(O.compose_frames(fgr, pha, "video")).numpy().astype(int)
    ref_pha = O.frames_to_u8(pha).numpy().astype(int)
    ref_fgr = O.frames_to_u8(fgr).numpy().astype(int)
    assert len(com.frames) == T and com.frames[0].shape == (H, W, 3)
    assert np.abs(np.stack(com.frames).astype(int) - ref_com).max() <= 1
    assert np.abs(np.stack(alpha.frames).astype(int) - ref_pha).max() <= 1
    assert np.abs(np.stack(fg.frames).astype(int) - ref_fgr).max() <= 1
    from PIL import Image
    rgba = np.asarray(Image.open(tmp_path / "com" / "0000.png"))
    assert rgba.shape == (H, W, 4)
    assert np.abs(rgba[..., 3].astype(int) - ref_pha[0, ..., 0]).max() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_convert_video_pipelined_uint8_source_equals_float_source(dtype):
    """The double-buffered upload / download path of convert_video (pinned staging, side streams): 7 frames in chunks of 3
    (two full chunks and a short last one, so both buffer sets are reused), uint8 frames uploaded as bytes and divided by
    255 on the device against the same frames as float tensors (ToTensor's values): identical bytes out; and chunked output
    equals the oracle pipeline to one count."""
    from robustvideomatting_amd import MattingNetwork
    T, H, W = 7, 64, 96
    clip = synthetic.make_clip(T, H, W, seed=6)[0]
    u8 = O.frames_to_u8(clip).permute(0, 3, 1, 2).contiguous()            # [T,3,H,W] uint8, what a decoder yields
    f32 = u8.float().div(255)                                             # ToTensor
    params = synthetic.make_params("mobilenetv3", "deep_guided_filter", seed=1)
    model = MattingNetwork("mobilenetv3").eval()
    model.load_state_dict(params)
    model = model.to("cuda", dtype)
    outs = []
    for src in (I.TensorSource(u8), I.TensorSource(f32), I.TensorSource(u8.clone()).pin()):
        com, alpha = I.ListWriter(), I.ListWriter()
        I.convert_video(model, src, downsample_ratio=0.5, output_type="video", output_composition=com, output_alpha=alpha,
                        seq_chunk=3, progress=False)
        one = I.ListWriter()
        I.convert_video(model, src, downsample_ratio=0.5, output_type="video", output_composition=one, seq_chunk=3, progress=False)
        assert len(com.frames) == T and len(alpha.frames) == T and len(one.frames) == T
        outs.append((np.stack(com.frames), np.stack(alpha.frames), np.stack(one.frames)))
    # what reaches the model is identical bit for bit (bytes divided by 255 on the device == ToTensor on the host) ...
    pipe = I._Pipeline(torch.device("cuda", torch.cuda.current_device()))
    for k in range(3):                                                    # both buffer sets, one reused
        lo = 2 * k
        a = pipe.upload(k, [u8[i] for i in range(lo, lo + 3)], dtype, True)
        b = pipe.upload(k, [f32[i] for i in range(lo, lo + 3)], dtype, False)
        torch.cuda.synchronize()
        assert a.dtype == dtype and torch.equal(a, b) and torch.equal(b.cpu(), f32[lo:lo + 3].to(dtype))
    # ... so the frames out are the same (fp32: bit for bit; 16-bit: a context that switches between its chunk plans takes a
    # carried state at its 16-bit value or at the fp32 value it kept, depending on history - rounding-level, see rvm_mi355x.h)
    for a, b in list(zip(outs[0], outs[1])) + list(zip(outs[0], outs[2])):      # float source, pinned source (no staging copy)
        if dtype == torch.float32:
            assert np.array_equal(a, b)
        else:
            assert np.abs(a.astype(int) - b.astype(int)).max() <= 1
    assert np.abs(outs[0][0].astype(int) - outs[0][2].astype(int)).max() <= 1   # two-output path vs the fused uint8 path
    if dtype == torch.float32:
        out = O.forward(params, f32.unsqueeze(0), downsample_ratio=0.5)
        ref = O.frames_to_u8(O.compose_frames(out[0][0], out[1][0], "video")).numpy().astype(int)
        assert np.abs(outs[0][0].astype(int) - ref).max() <= 1
