"""GPU: parity of the HIP path (through the C ABI, via MattingNetwork) against the CPU oracle.

Tolerances are the ones BASELINE.json's north_star states: max-abs 1e-4 (fp32 kernel mode),
2e-3 (fp16 mode) on fgr / pha of identical synthetic clips; no drift over T=300.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

TOL32 = 1e-4
TOL16 = 2e-3


@pytest.fixture(scope="module")
def gc():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gpu_check
    return gpu_check


def _setup(dtype, variant="mobilenetv3", refiner="deep_guided_filter", seed=0, graph=False):
    from robustvideomatting_amd import MattingNetwork, synthetic
    p = synthetic.make_params(variant, refiner, seed=seed, perturb_box_filter=True)
    m = MattingNetwork(variant, refiner).eval()
    m.load_state_dict(p)
    m = m.to("cuda", dtype)
    if graph:
        m.enable_graph(True)
    return m, p


# ------------------------------------------------------------------ fp32 kernel mode
def test_fp32_ds025_odd_sizes_all_taps(gc):
    """122x218 -> 30x54 low-res: every pyramid level is odd somewhere (ceil pool, crop-after-upsample)."""
    w = gc.run("fp32", H=122, W=218, ds=0.25, T=3, verbose=False)
    for k, v in w.items():
        assert v < TOL32, (k, v, w)


def test_fp32_ds1_batch2(gc):
    """downsample_ratio=1 skips the refiner (model.py:52-55,63-64); B=2 exercises the batch index."""
    w = gc.run("fp32", H=70, W=90, ds=1.0, T=2, B=2, verbose=False)
    for k, v in w.items():
        assert v < TOL32, (k, v, w)


def test_fp32_other_ratio(gc):
    """A ratio that is not 1/2^k: generic scale-factor-driven bilinear coordinates."""
    w = gc.run("fp32", H=150, W=201, ds=0.4, T=2, verbose=False)
    for k, v in w.items():
        assert v < TOL32, (k, v, w)


def test_fp32_fast_guided_filter(gc):
    """cov/(var+1e-5) amplifies fp32 rounding-order differences where var ~ eps, so the
    coefficient tap `ab` is checked on a textured clip through the final outputs only."""
    w = gc.run("fp32", H=122, W=218, ds=0.25, T=2, refiner="fast_guided_filter", verbose=False, noise=0.3)
    for k in ("hid", "r1", "r2", "r3", "r4"):
        assert w[k] < TOL32
    # The division is ill-conditioned: the fp32 reference path itself sits a few 1e-5..1e-4 away from
    # the same arithmetic in fp64.  Bound = 1e-4, or 3x the fp32 oracle's own distance to fp64 if larger.
    from robustvideomatting_amd import synthetic
    p = synthetic.make_params("mobilenetv3", "fast_guided_filter", seed=0, perturb_box_filter=True)
    clip = synthetic.make_clip(2, 122, 218, seed=0, B=1, noise=0.3)
    bound, own = gc.fast_gf_bound(p, clip, 0.25, TOL32)
    assert w["fgr"] < bound and w["pha"] < bound, (w, own)


def test_fp32_hd_frame(gc):
    """BASELINE configs[1] shape at full size, two frames (second one consumes the state)."""
    w = gc.run("fp32", H=1080, W=1920, ds=0.25, T=2, verbose=False)
    for k, v in w.items():
        assert v < TOL32, (k, v, w)


# ------------------------------------------------------------------ fp16 mode
# "fp16" = fp16 tensors at the boundary (src, fgr, pha, r1..r4), fp32 storage + arithmetic inside:
# the parity-gated mode (2e-3).  "fp16fast" = fp16 storage + v_mfma_f32_16x16x32_f16 inside.
def test_fp16_small(gc):
    w = gc.run("fp16", H=122, W=218, ds=0.25, T=3, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w


def test_fp16_hd_frame(gc):
    w = gc.run("fp16", H=1080, W=1920, ds=0.25, T=2, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w


def test_full_size_lowres_batch2_all_modes(gc):
    """The HD frame's 270x480 low-res map with two streams, in every arithmetic mode: the launches that carry a
    neighbour's work (stem in the downsample kernel, block 1's project, SE tails in the project GEMMs, the OutputBlock
    conv in the guided filter) on full-size grids with a batch index."""
    w = gc.run("fp32", H=540, W=960, ds=0.5, T=2, B=2, verbose=False)
    for k, v in w.items():
        assert v < TOL32, (k, v, w)
    w = gc.run("fp16", H=540, W=960, ds=0.5, T=2, B=2, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w
    w = gc.run("fp16fast", H=540, W=960, ds=0.5, T=2, B=2, verbose=False)
    assert w["fgr"] < 2e-2 and w["pha"] < 2e-2, w          # fp16 arithmetic: fp16 accuracy, not the parity-gated mode


def test_fp16_4k_frame(gc):
    """BASELINE configs[2]: 3840x2160, ds=0.125."""
    w = gc.run("fp16", H=2160, W=3840, ds=0.125, T=1, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w


def test_fp16fast_is_fp16_accurate(gc):
    """Pure-fp16 kernels (fp16 activations and weights, fp32 accumulate).  On the synthetic stress
    weights, rounding the WEIGHTS alone to fp16 already moves pha by ~1-2e-3 (DESIGN.md, fp16
    study), so this mode is held to the accuracy of an fp16 pipeline - 1e-2 - not to 2e-3."""
    w = gc.run("fp16fast", H=122, W=218, ds=0.25, T=3, verbose=False)
    assert w["fgr"] < 1e-2 and w["pha"] < 1e-2, w
    w = gc.run("fp16fast", H=1080, W=1920, ds=0.25, T=2, verbose=False)
    assert w["fgr"] < 1e-2 and w["pha"] < 1e-2, w


# ------------------------------------------------------------------ resnet50 variant (BASELINE configs[3])
def test_resnet50_fp32_all_taps(gc):
    """7x7 s2 stem, max-pool, strided / dilated / channel-chunked 3x3 bottleneck convs, strided 1x1."""
    w = gc.run("fp32", H=122, W=218, ds=0.25, T=2, variant="resnet50", verbose=False)
    for k, v in w.items():
        assert v < TOL32, (k, v, w)


def test_resnet50_fp16_hd_frame(gc):
    w = gc.run("fp16", H=1080, W=1920, ds=0.25, T=2, variant="resnet50", verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w


def test_resnet50_fp16fast(gc):
    w = gc.run("fp16fast", H=122, W=218, ds=0.25, T=2, variant="resnet50", verbose=False)
    assert w["fgr"] < 1e-2 and w["pha"] < 1e-2, w


# ------------------------------------------------------------------ shape sweep
@pytest.mark.parametrize("H,W,ds,B", [
    (97, 131, 0.5, 1),        # odd everything; low-res 48x65
    (64, 64, 1.0, 3),         # no refiner, batch 3
    (45, 77, 0.37, 2),        # ratio that is no power of two, tiny pyramid (16x28 -> 1x2 at 1/16)
    (200, 360, 0.6, 1),       # larger low-res frame: several bands / tiles per level
    (33, 520, 0.25, 1),       # very wide, 8 rows low-res
])
def test_fp32_shape_sweep_framewise_and_chunked(gc, H, W, ds, B):
    """Band / tile / slice edges of every kernel on shapes unlike the benchmark's, frame by frame (all taps) and as
    one chunk (plan-time tile measurement is active whenever B*T > 1)."""
    w = gc.run("fp32", H=H, W=W, ds=ds, T=2, B=B, verbose=False)
    for k, v in w.items():
        assert v < TOL32, (k, v, w)
    from oracle import rvm_oracle as O
    from robustvideomatting_amd import synthetic
    m, p = _setup(torch.float32)
    clip = synthetic.make_clip(3, H, W, seed=3, B=B)
    want = O.forward(p, clip, downsample_ratio=ds)
    with torch.no_grad():
        got = m(clip.cuda(), downsample_ratio=ds)
    for a, b in zip(got, want):
        assert a.shape == b.shape and (a.cpu() - b).abs().max() < TOL32


def test_column_tiled_depthwise_forced_on_a_small_frame(gc, monkeypatch):
    """mbconv_ed in column tiles (the form wide frames need) forced on a small odd-sized frame: every tile edge, halo and
    SE partial row against the oracle, all taps."""
    monkeypatch.setenv("RVM_ED_CWT", "13")
    w = gc.run("fp32", H=150, W=201, ds=0.8, T=2, B=2, verbose=False)
    for k, v in w.items():
        assert v < TOL32, (k, v, w)
    w = gc.run("fp16", H=122, W=218, ds=0.5, T=2, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w


@pytest.mark.parametrize("cfg", ["7", "9", "11", "12"])
def test_lds_tiled_pointwise_gemm_forced(gc, monkeypatch, cfg):
    """pw_gemm_lds (64- or 128-pixel x 64-channel tiles, both operands by LDS-DMA through a six-stage ring; cfg 9 / 12:
    split-K in four / two slices + reduce pass) is normally one of the candidates plan-time measurement chooses from; forced
    here on every 1x1 layer that admits it, both variants (resnet50: strided 1x1, residual + ReLU after the add, pooled
    sums for the LR-ASPP gate; mobilenetv3: channel counts that are no multiple of 32, i.e. a masked last K step).  The
    fp16-storage mode has no such kernel (the hook then selects configuration 0)."""
    monkeypatch.setenv("RVM_PW_CFG", cfg)
    monkeypatch.setenv("RVM_AUTOTUNE", "0")
    w = gc.run("fp32", H=122, W=218, ds=0.5, T=2, B=2, variant="resnet50", verbose=False)
    for k, v in w.items():
        assert v < TOL32, (k, v, w)
    w = gc.run("fp16", H=150, W=201, ds=0.8, T=2, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w
    w = gc.run("fp16fast", H=122, W=218, ds=0.25, T=2, variant="resnet50", verbose=False)
    assert w["fgr"] < 1e-2 and w["pha"] < 1e-2, w


@pytest.mark.parametrize("k", ["0", "1", "2", "3"])
def test_chunked_decoder_convs_forced(gc, monkeypatch, k):
    """decode3.conv / decode2.conv in their second packing (56-channel chunks, one block per 20- or 8-row tile looping over
    the chunks, x2-upsampled sources staged chunk by chunk): normally a plan-time candidate for batched frames only; forced
    here (B = 2 and a chunk of T = 2) against the oracle, all taps."""
    monkeypatch.setenv("RVM_CONV_ALT", k)
    monkeypatch.setenv("RVM_AUTOTUNE", "0")
    w = gc.run("fp32", H=150, W=201, ds=0.8, T=2, B=2, verbose=False)
    for key, v in w.items():
        assert v < TOL32, (key, v, w)
    w = gc.run("fp16", H=540, W=960, ds=0.5, T=2, B=2, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w


@pytest.mark.parametrize("tile", ["off", "nofold", "4,4", "2,16", "8,4", "2,8", "4,8"])
def test_fused_inverted_residual_blocks_forced_tiles(gc, monkeypatch, tile):
    """mbconv_fused (features[2], [3], [7]..[10] as ONE launch each: expand -> depthwise -> project per tile of output
    pixels) with every tile shape it offers forced in turn, and switched off (the two-launch form mbconv_ed + pw_gemm it
    replaced, which stays the fallback for shapes the fused kernel does not fit): odd-sized maps (tiles hanging over every
    border, stride-2 halos), a batch index, all three arithmetic modes, all taps.  features[1] (depthwise + 16 x 16 project on
    the stem output) rides inside features[2]'s launch; "nofold" keeps it a launch of its own."""
    if tile == "off":
        monkeypatch.setenv("RVM_MB_FUSE", "0")
    elif tile == "nofold":
        monkeypatch.setenv("RVM_MB1_FOLD", "0")
    else:
        monkeypatch.setenv("RVM_MBF_TILE", tile)
    monkeypatch.setenv("RVM_AUTOTUNE", "0")
    w = gc.run("fp32", H=150, W=201, ds=0.8, T=2, B=2, verbose=False)
    for k, v in w.items():
        assert v < TOL32, (k, v, w)
    w = gc.run("fp16", H=540, W=960, ds=0.5, T=2, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w
    for k in ("f2", "f3", "f4"):
        assert w[k] < 1e-4, w                       # the encoder taps behind the fused blocks stay fp32-accurate
    w = gc.run("fp16fast", H=122, W=218, ds=0.5, T=2, verbose=False)
    assert w["fgr"] < 1e-2 and w["pha"] < 1e-2, w


@pytest.mark.parametrize("variant", ["off"] + [str(v) for v in range(12)])
def test_decoder_convs_with_weights_in_registers_forced(gc, monkeypatch, variant):
    """conv_cat (the decoder's upsample + concat + conv + BN + ReLU layers with the weight fragments in registers, K split over
    the waves) with each of its 12 variants (4 tiles x 3 wave layouts) forced where the layer admits it, and switched off
    (conv_kxk): odd-sized maps, a batch index, all arithmetic modes, every tap."""
    if variant == "off":
        monkeypatch.setenv("RVM_CONV_REG", "0")
    else:
        monkeypatch.setenv("RVM_CONV_REG_VAR", variant)
    monkeypatch.setenv("RVM_AUTOTUNE", "0")
    w = gc.run("fp32", H=150, W=201, ds=0.8, T=2, B=2, verbose=False)
    for k, v in w.items():
        assert v < TOL32, (k, v, w)
    w = gc.run("fp16", H=540, W=960, ds=0.5, T=2, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w
    for k in ("x1", "x2", "x3", "hid"):
        assert w[k] < 1e-4, w
    w = gc.run("fp16fast", H=122, W=218, ds=0.5, T=2, verbose=False)
    assert w["fgr"] < 1e-2 and w["pha"] < 1e-2, w


@pytest.mark.parametrize("tile", ["off", "4,8", "8,8", "4,16", "8,16"])
def test_fused_convgru_forced_tiles(gc, monkeypatch, tile):
    """gru_fused (decode1 / decode2: gate conv, r * h, candidate conv and the state blend as ONE launch per time step) with
    each of its tile shapes forced, and switched off (the conv_kxk ih + hh pair it replaced, which the two small-map GRUs
    still use): odd-sized maps (tiles hanging over every border), a batch index, chunks of T > 1 (the state of step t - 1
    read back from the arena), zero and given initial states, all arithmetic modes, every tap including r1..r4."""
    if tile == "off":
        monkeypatch.setenv("RVM_GRU_FUSE", "0")
    else:
        monkeypatch.setenv("RVM_GRU_TILE", tile)
    monkeypatch.setenv("RVM_AUTOTUNE", "0")
    w = gc.run("fp32", H=150, W=201, ds=0.8, T=3, B=2, verbose=False)
    for k, v in w.items():
        assert v < TOL32, (k, v, w)
    w = gc.run("fp16", H=540, W=960, ds=0.5, T=2, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w
    for k in ("x1", "x2"):
        assert w[k] < 1e-4, w                       # the decoder features behind the fused GRUs stay fp32-accurate
    w = gc.run("bf16", H=122, W=218, ds=0.5, T=2, verbose=False)
    assert w["fgr"] < 8e-3 and w["pha"] < 8e-3, w
    w = gc.run("fp16fast", H=122, W=218, ds=0.5, T=2, verbose=False)
    assert w["fgr"] < 1e-2 and w["pha"] < 1e-2, w
    # a chunk in one native call (states of step t - 1 read back from the arena, the caller's copy written by the last
    # step) against the same frames fed one by one
    from robustvideomatting_amd import synthetic
    for tdt, tol in ((torch.float32, 1e-5), (torch.float16, 2e-3)):
        m, _ = _setup(tdt)
        clip = synthetic.make_clip(3, 90, 134, seed=5, B=2).cuda().to(tdt)
        with torch.no_grad():
            a = m(clip, None, None, None, None, None, 0.6)
            rec = [None] * 4
            for s in range(3):
                o = m(clip[:, s], None, *rec, 0.6)
                rec = o[2:]
                assert (o[0].float() - a[0][:, s].float()).abs().max() < tol and (o[1].float() - a[1][:, s].float()).abs().max() < tol
        for x, y in zip(rec, a[2:]):
            assert (x.float() - y.float()).abs().max() < tol


@pytest.mark.parametrize("tdt", [torch.float16, torch.float32, torch.bfloat16])
def test_dgf_apply_power_of_two_kernel_is_bit_identical_to_the_general_one(monkeypatch, tdt):
    """dgf_apply_pow2 (ratios 1/4 and 1/8 with W % 8 == 0: a thread's 8 pixels on 4 / 3 fixed low-res columns with compile-time
    fractions, 1 or 2 rows per block) against the general kernel (RVM_DGF_POW2=0) on the same plan: every fgr / pha bit equal -
    HD and 4K, two segments per row (W > 2048), the smallest map, a batch index; a ratio the special kernel does not take
    (W % 4 != 0) runs the general kernel either way."""
    from robustvideomatting_amd import synthetic
    monkeypatch.setenv("RVM_AUTOTUNE", "0")

    def run(H, W, ds, B, pow2, rows=None):
        monkeypatch.setenv("RVM_DGF_POW2", "1" if pow2 else "0")
        if rows is None:
            monkeypatch.delenv("RVM_DGF_ROWS", raising=False)
        else:
            monkeypatch.setenv("RVM_DGF_ROWS", str(rows))
        m, _ = _setup(tdt)
        src = synthetic.make_clip(1, H, W, seed=1, B=B)[:, 0].to("cuda", tdt)
        with torch.no_grad():
            o = m(src, downsample_ratio=ds)
        return o[0].clone(), o[1].clone()

    for H, W, ds, B in ((1080, 1920, 0.25, 2), (2160, 3840, 0.125, 1), (64, 4096, 0.25, 1), (48, 4104, 0.125, 1), (8, 8, 0.25, 3),
                        (270, 479, 0.25, 1)):
        base = run(H, W, ds, B, False)
        for rows in (None, 1):
            got = run(H, W, ds, B, True, rows)
            assert torch.equal(base[0], got[0]) and torch.equal(base[1], got[1]), (H, W, ds, B, rows)


def test_fused_blocks_launch_count():
    """45 launches per mobilenetv3 frame with the fused blocks and GRUs, 54 without."""
    from robustvideomatting_amd import synthetic
    m, _ = _setup(torch.float16)
    src = synthetic.make_clip(1, 1080, 1920, seed=0)[:, 0].cuda().half()
    with torch.no_grad():
        m(src, downsample_ratio=0.25)
    assert m.launch_count() <= 45


def test_in_place_parameter_edits_are_picked_up():
    """p.add_() bumps the version counter, p.data = ... changes the storage pointer: both re-pack the weights (checked on
    every call here; every 64th under no_grad by default, documented in INTEGRATION.md)."""
    m, _ = _setup(torch.float32)
    m.edit_check_interval = 1
    src = torch.rand(1, 3, 64, 96, device="cuda")
    with torch.no_grad():
        a = m(src, downsample_ratio=0.5)
        m.project_mat.conv.bias.add_(0.05)
        b = m(src, downsample_ratio=0.5)
        assert (b[1] - a[1]).abs().max() > 1e-3
        m.project_mat.conv.bias.data = m.project_mat.conv.bias.data - 0.05
        c = m(src, downsample_ratio=0.5)
    assert (c[1] - a[1]).abs().max() < 1e-6


def test_plans_are_kept_reported_and_trimmed():
    """rvm_memory_bytes / rvm_trim: the context reports what it holds (weights, the active plan's arena, parked plans) and
    frees the parked plans on request; results are unchanged afterwards."""
    m, _ = _setup(torch.float16)
    a = torch.rand(1, 3, 128, 160, device="cuda").half()
    b = torch.rand(1, 3, 96, 128, device="cuda").half()
    with torch.no_grad():
        ra = m(a, downsample_ratio=0.5)
        mem1 = m._ctx.memory_bytes()
        assert mem1["weights"] > 0 and mem1["active_plan"] > 0 and mem1["parked_plans"] == 0
        m(b, downsample_ratio=0.5)
        mem2 = m._ctx.memory_bytes()
        assert mem2["parked_plans"] == mem1["active_plan"] and mem2["total"] == mem2["weights"] + mem2["active_plan"] + mem2["parked_plans"]
        m._ctx.trim()
        assert m._ctx.memory_bytes()["parked_plans"] == 0
        ra2 = m(a, downsample_ratio=0.5)                  # re-planned from scratch
    for x, y in zip(ra, ra2):
        assert torch.equal(x, y)


def test_fp32_1080p_at_ratio_one():
    """1920x1080 at downsample_ratio=1: the low-res map is the full frame, wider than any full-width LDS tile of the
    depthwise kernel (column tiles) - the largest shape the reference runs on common inputs."""
    from oracle import rvm_oracle as O
    from robustvideomatting_amd import synthetic
    m, p = _setup(torch.float32)
    clip = synthetic.make_clip(1, 1080, 1920, seed=2)
    with torch.no_grad():
        got = m(clip[:, 0].cuda(), downsample_ratio=1.0)
    want = O.forward(p, clip[:, 0], downsample_ratio=1.0)
    for a, b in zip(got, want):
        assert a.shape == b.shape and (a.cpu() - b).abs().max() < TOL32


# ------------------------------------------------------------------ golden fixture (no oracle involved)
@pytest.mark.parametrize("name", ["mobilenetv3_ds025", "mobilenetv3_ds1", "mobilenetv3_fast", "resnet50_ds025"])
def test_against_committed_fixture(name):
    from robustvideomatting_amd import MattingNetwork, synthetic
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_%s.npz" % name))
    H, W, T, seed = (int(x) for x in g["meta"])
    ds, refiner = float(g["ds"]), str(g["refiner"])
    variant = name.split("_")[0]
    m = MattingNetwork(variant, refiner).eval()
    m.load_state_dict(synthetic.make_params(variant, refiner, seed=seed))
    m = m.cuda()
    clip = synthetic.make_clip(T, H, W, seed=seed).cuda()
    with torch.no_grad():
        out = m(clip, downsample_ratio=ds)                    # 5-D input, keyword ratio
    tol = TOL32
    if refiner == "fast_guided_filter":                       # self-calibrated bound, see test_fp32_fast_guided_filter
        import gpu_check
        tol, _ = gpu_check.fast_gf_bound(synthetic.make_params(variant, refiner, seed=seed), clip.float().cpu(), ds, TOL32, variant)
    assert out[0].shape == (1, T, 3, H, W) and out[1].shape == (1, T, 1, H, W)
    assert (out[1].cpu() - torch.from_numpy(g["pha"])).abs().max() < tol
    assert (out[0][:, -1].cpu() - torch.from_numpy(g["fgr_last"])).abs().max() < tol
    for i in range(4):
        assert (out[2 + i].cpu() - torch.from_numpy(g["r%d" % (i + 1)])).abs().max() < TOL32


# ------------------------------------------------------------------ contract details
def test_time_series_equals_framewise_and_rank_rules():
    """[B,T,...] input: with chunk_batching off the module feeds frame by frame (bit-identical to T calls);
    with it on (default) the chunk is one native call whose tile shapes differ, so equality is to rounding."""
    m, _ = _setup(torch.float32)
    from robustvideomatting_amd import synthetic
    clip = synthetic.make_clip(3, 64, 96, seed=4, B=2).cuda()
    with torch.no_grad():
        a = m(clip, None, None, None, None, None, 0.5)
        m.chunk_batching = False
        b = m(clip, None, None, None, None, None, 0.5)
        rec = [None] * 4
        for t in range(3):
            o = m(clip[:, t], None, *rec, 0.5)
            rec = o[2:]
            assert o[0].ndim == 4 and torch.equal(o[0], b[0][:, t]) and torch.equal(o[1], b[1][:, t])
            assert (o[0] - a[0][:, t]).abs().max() < 1e-5 and (o[1] - a[1][:, t]).abs().max() < 1e-5
    assert a[0].ndim == 5 and all(r.ndim == 4 for r in a[2:])
    assert a[0].shape == b[0].shape == (2, 3, 3, 64, 96)
    for x, y, z in zip(rec, b[2:], a[2:]):
        assert torch.equal(x, y) and (x - z).abs().max() < 1e-5


@pytest.mark.parametrize("mode", ["fp32", "fp16", "fp32-graph"])
def test_chunked_clip_against_oracle(mode):
    """[B=2,T=4] chunks in one native call each (rvm_plan_seq), two chunks with the state
    carried between them, against the oracle fed frame by frame."""
    from oracle import rvm_oracle as O
    from robustvideomatting_amd import synthetic
    half = mode == "fp16"
    m, p = _setup(torch.float16 if half else torch.float32, graph=mode.endswith("graph"))
    tol = TOL16 if half else TOL32
    B, T, H, W, ds = 2, 4, 122, 218, 0.25
    clip = synthetic.make_clip(2 * T, H, W, seed=9, B=B)
    dev = clip.cuda().half() if half else clip.cuda()
    ref = dev.float().cpu()                                  # the oracle sees the rounded input
    ro = [None] * 4
    want_f, want_p = [], []
    for t in range(2 * T):
        o = O.forward(p, ref[:, t], *ro, downsample_ratio=ds)
        ro = o[2:]
        want_f.append(o[0]); want_p.append(o[1])
    rg = [None] * 4
    with torch.no_grad():
        for k in range(2):
            o = m(dev[:, k * T:(k + 1) * T], None, *rg, ds)
            rg = o[2:]
            assert o[0].shape == (B, T, 3, H, W) and o[1].shape == (B, T, 1, H, W)
            for t in range(T):
                assert (o[0][:, t].float().cpu() - want_f[k * T + t]).abs().max() < tol
                assert (o[1][:, t].float().cpu() - want_p[k * T + t]).abs().max() < tol
    for x, y in zip(rg, ro):
        assert (x.float().cpu() - y).abs().max() < tol
    # segmentation pass on a chunk (model.py:69-71)
    with torch.no_grad():
        s = m(dev[:, :T], None, None, None, None, None, ds, True)
    want = O.forward(p, ref[:, :T], downsample_ratio=ds, segmentation_pass=True)
    assert s[0].shape == want[0].shape and (s[0].float().cpu() - want[0]).abs().max() < tol * 5


def test_alternating_shapes_switch_between_kept_plans(monkeypatch):
    """A driver that alternates call shapes (convert_video's short last chunk; 4-D and 5-D calls in turn) switches between
    plans the context keeps instead of re-planning: results are bit-identical to models that only ever saw one shape."""
    from robustvideomatting_amd import synthetic
    monkeypatch.setenv("RVM_AUTOTUNE", "0")          # model-chosen tiles: two contexts then run the very same kernels
    m, _ = _setup(torch.float16)
    clip = synthetic.make_clip(4, 122, 218, seed=11).cuda().half()
    other = synthetic.make_clip(2, 64, 96, seed=12).cuda().half()
    refs = {}
    with torch.no_grad():
        for key, fn in (("a", lambda mm: mm(clip[:, 0], None, None, None, None, None, 0.25)),
                        ("b", lambda mm: mm(clip[:, :3], None, None, None, None, None, 0.25)),
                        ("c", lambda mm: mm(other[:, 0], None, None, None, None, None, 0.5))):
            fresh, _ = _setup(torch.float16)
            refs[key] = (fn, [t.clone() for t in fn(fresh)])
        for rnd in range(3):
            for key in ("a", "b", "c", "b", "a"):
                fn, want = refs[key]
                got = fn(m)
                for x, y in zip(got, want):
                    assert torch.equal(x, y), (rnd, key)


def test_graph_replay_is_bit_identical_to_eager():
    from robustvideomatting_amd import synthetic
    me, _ = _setup(torch.float16)
    mg, _ = _setup(torch.float16, graph=True)
    clip = synthetic.make_clip(5, 122, 218, seed=2).cuda().half()
    re_, rg = [None] * 4, [None] * 4
    with torch.no_grad():
        for t in range(5):
            oe = me(clip[:, t], None, *re_, 0.25)
            og = mg(clip[:, t], None, *rg, 0.25)
            re_, rg = oe[2:], og[2:]
            for x, y in zip(oe, og):
                assert torch.equal(x, y)


def test_upstream_call_form_and_determinism():
    m, _ = _setup(torch.float32)
    src = torch.rand(1, 3, 64, 80, device="cuda")
    with torch.no_grad():
        a = m(src, None, None, None, None, None, 0.5)
        b = m(src, None, None, None, None, 0.5)               # upstream: (src, r1..r4, ds)
        c = m(src, *a[2:], 0.5)                               # upstream with states
        d = m(src, src, *a[2:], 0.5)                          # this fork: (src, trimap, r1..r4, ds)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    for x, y in zip(c, d):
        assert torch.equal(x, y)
    assert not torch.equal(a[1], c[1])                        # the state matters


def test_segmentation_pass(gc):
    from oracle import rvm_oracle as O
    m, p = _setup(torch.float32)
    src = torch.rand(1, 3, 64, 96, device="cuda")
    with torch.no_grad():
        out = m(src, downsample_ratio=0.5, segmentation_pass=True)
    ref = O.forward(p, src.cpu(), downsample_ratio=0.5, segmentation_pass=True)
    assert len(out) == 5 and out[0].shape == ref[0].shape
    assert (out[0].cpu() - ref[0]).abs().max() < TOL32
    for a, b in zip(out[1:], ref[1:]):
        assert (a.cpu() - b).abs().max() < TOL32


def test_errors():
    m, _ = _setup(torch.float32)
    src = torch.rand(1, 3, 64, 96, device="cuda")
    with torch.no_grad():
        out = m(src, downsample_ratio=0.5)
        with pytest.raises(RuntimeError, match="resolution change"):
            m(torch.rand(1, 3, 80, 96, device="cuda"), None, *out[2:], 0.5)
        with pytest.raises(RuntimeError, match="should be the same"):
            m(src.half(), downsample_ratio=0.5)
        with pytest.raises(RuntimeError):
            m(src, downsample_ratio=1.5)
        zero = torch.zeros(1, 1, 1, 1, device="cuda")         # ONNX-style explicit zero state
        z = m(src, None, zero, zero, zero, zero, 0.5)
    for a, b in zip(out, z):
        assert torch.equal(a, b)


def test_foreign_nchw_state_is_accepted():
    """A state produced elsewhere (plain NCHW-contiguous tensor) works like our own channels-last one."""
    m, _ = _setup(torch.float32)
    src = torch.rand(1, 3, 64, 96, device="cuda")
    with torch.no_grad():
        out = m(src, downsample_ratio=0.5)
        nchw = [r.contiguous() for r in out[2:]]
        a = m(src, None, *out[2:], 0.5)
        b = m(src, None, *nchw, 0.5)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


# ------------------------------------------------------------------ drift
def _drift(dtype, tol, T, H, W, ds, state_tol, variant="mobilenetv3"):
    """T frames with carried state against the oracle, frame by frame (the clip is generated on the fly): the error
    stays inside the bound at every frame and does not grow (last third no worse than 2x the first third)."""
    from oracle import rvm_oracle as O
    from robustvideomatting_amd import synthetic
    tdt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[dtype]
    m, p = _setup(tdt, variant=variant)
    rg, ro = [None] * 4, [None] * 4
    errs = []
    with torch.no_grad():
        for fr in synthetic.clip_frames(T, H, W, seed=9):
            dev = fr[None].cuda().to(tdt)
            og = m(dev, None, *rg, ds)
            oo = O.forward(p, dev.float().cpu(), *ro, downsample_ratio=ds, variant=variant)
            rg, ro = og[2:], oo[2:]
            errs.append(max((og[0].float() - oo[0].cuda()).abs().max().item(),
                            (og[1].float() - oo[1].cuda()).abs().max().item()))
    errs = np.array(errs)
    assert errs.max() < tol, errs.max()
    third = max(T // 3, 1)
    assert errs[-third:].mean() <= 2.0 * errs[:third].mean() + 1e-7, (errs[:third].mean(), errs[-third:].mean())
    state_err = max((a.float().cpu() - b).abs().max().item() for a, b in zip(rg, ro))
    assert state_err < state_tol, state_err
    return errs


@pytest.mark.parametrize("dtype,tol", [("fp32", TOL32), ("fp16", TOL16)])
def test_no_drift_over_300_frames(dtype, tol):
    """T=300 with carried state at a small size; the recurrent state itself stays inside the output bound."""
    _drift(dtype, tol, T=300, H=96, W=160, ds=0.5, state_tol=tol)


def test_config1_hd_fp16_T300_drift():
    """BASELINE configs[1] at its real size and length: 1920x1080, ds=0.25, fp16, T=300 with carried state."""
    _drift("fp16", TOL16, T=300, H=1080, W=1920, ds=0.25, state_tol=TOL16)


def test_config2_4k_fp16_carried_state():
    """BASELINE configs[2] at its real size: 3840x2160, ds=0.125, fp16; frames 2 and 3 consume the recurrent state."""
    _drift("fp16", TOL16, T=3, H=2160, W=3840, ds=0.125, state_tol=TOL16)


def test_config2_4k_fp16_T300_drift():
    """BASELINE configs[2] at its real size AND length: 3840x2160, ds=0.125, fp16, T=300 with carried state, every frame
    against the oracle (frames generated on the fly: the clip would be 30 GB as one tensor)."""
    _drift("fp16", TOL16, T=300, H=2160, W=3840, ds=0.125, state_tol=TOL16)


def test_config0_512x288_ds1_fp32_T30():
    """BASELINE configs[0] at its real size: 512x288, ds=1 (no refiner), fp32, T=30."""
    _drift("fp32", TOL32, T=30, H=288, W=512, ds=1.0, state_tol=TOL32)


# ------------------------------------------------------------------ bf16 tensors (decoder.py:184-185: any src.dtype)
# bf16 has 8 significant bits: rounding fgr / pha in [0.5, 1) to bf16 alone costs up to 2^-9 = 1.953e-3 of the 2e-3
# the north_star allows.  The interior is fp32-accurate (same split-f16 MFMA path as the fp16 mode) and the recurrent
# state is carried in fp32 inside the context (rvm_mi355x.h, RVM_DTYPE_BF16IO), so what is left is output rounding.
def test_bf16_small_and_hd(gc):
    w = gc.run("bf16", H=122, W=218, ds=0.25, T=3, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w
    for k in ("r1", "r2", "r3", "r4"):
        assert w[k] < TOL16, w
    for k in ("f4", "aspp", "x1", "hid"):          # interior taps are fp32-accurate, far inside the output bound
        assert w[k] < 1e-4, w
    w = gc.run("bf16", H=1080, W=1920, ds=0.25, T=2, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w


def test_bf16_outputs_are_the_rounded_fp32_results():
    """Stronger than the 2e-3 bound: fgr / pha equal bf16(oracle) except where the oracle value sits within the
    interior's ~1e-5 of a rounding boundary (then one bf16 ulp off), and the states likewise."""
    from oracle import rvm_oracle as O
    from robustvideomatting_amd import synthetic
    m, p = _setup(torch.bfloat16)
    clip = synthetic.make_clip(3, 122, 218, seed=4).cuda().bfloat16()
    ref = clip.float().cpu()
    rg, ro = [None] * 4, [None] * 4
    with torch.no_grad():
        for t in range(3):
            og = m(clip[:, t], None, *rg, 0.25)
            oo = O.forward(p, ref[:, t], *ro, downsample_ratio=0.25)
            rg, ro = og[2:], oo[2:]
            for a, b in zip(og, oo):
                assert a.dtype == torch.bfloat16
                want = b.bfloat16().float()
                d = (a.float().cpu() - want).abs()
                ulp = want.abs() * 2.0 ** -7 + 5e-5        # one bf16 ulp, or the interior's own error next to 0 (clamp)
                assert (d <= ulp).all(), (d - ulp).max()
                assert (d > 1e-4).float().mean() < 0.02, (d > 1e-4).float().mean()


def test_bf16_no_drift_over_300_frames():
    _drift("bf16", TOL16, T=300, H=96, W=160, ds=0.5, state_tol=TOL16)


def test_bf16_hd_carried_state_60_frames():
    _drift("bf16", TOL16, T=60, H=1080, W=1920, ds=0.25, state_tol=TOL16)


def test_foreign_16bit_state_falls_back_to_its_own_value():
    """16-bit modes keep the fp32 value of the states they hand out; a state that does NOT come from this context
    (here: produced by a second model instance) must be used at its 16-bit value, like the oracle fed that value."""
    from oracle import rvm_oracle as O
    from robustvideomatting_amd import synthetic
    ma, p = _setup(torch.float16)
    mb, _ = _setup(torch.float16)
    clip = synthetic.make_clip(3, 122, 218, seed=6).cuda().half()
    with torch.no_grad():
        o1 = ma(clip[:, 0], None, None, None, None, None, 0.25)
        _ = mb(clip[:, 1], None, None, None, None, None, 0.25)          # mb now keeps states unrelated to o1's
        got = mb(clip[:, 2], None, *[r.clone() for r in o1[2:]], 0.25)
    want = O.forward(p, clip[:, 2].float().cpu(), *[r.float().cpu() for r in o1[2:]], downsample_ratio=0.25)
    for a, b in zip(got, want):
        assert (a.float().cpu() - b).abs().max() < TOL16


def test_config4_hd_fp16_batch8_autotuned_plan(gc):
    """The per-GPU share of BASELINE configs[4]: 8 HD streams batched, fp16, two frames (the second consumes the state),
    with the tiles the plan-time measurement really picks for B=8."""
    w = gc.run("fp16", H=1080, W=1920, ds=0.25, T=2, B=8, verbose=False)
    assert w["fgr"] < TOL16 and w["pha"] < TOL16, w
    for k in ("r1", "r2", "r3", "r4"):
        assert w[k] < TOL16, w
