"""Quality metrics: the numpy restatement against hand-checked cases (CPU), the HIP path against it (GPU)."""
import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO


def test_oracle_filter_matches_the_evaluators_definition():
    fx, fy = MO.gauss_filter(1.4)
    assert fx.shape == (9, 9) and np.allclose(fy, fx.T)
    assert abs((fx ** 2).sum() - 1.0) < 1e-12                       # evaluate_hr.py:215-216
    assert abs(fx.sum()) < 1e-12 and np.allclose(fx, -fx[:, ::-1])  # odd along x: a derivative filter
    # a horizontal ramp: correlation with an x-derivative filter gives a constant, the y filter gives 0
    ramp = np.tile(np.arange(32, dtype=np.float64) / 31, (24, 1))
    gx = MO.filter2d_replicate(ramp, fx)[8:-8, 8:-8]
    gy = MO.filter2d_replicate(ramp, fy)[8:-8, 8:-8]
    want = (fx * (np.arange(9) - 4)[None, :]).sum() / 31
    assert np.allclose(gx, want) and np.allclose(gy, 0)


def test_oracle_hand_cases():
    a = np.zeros((4, 6)); b = np.full((4, 6), 0.5)
    assert MO.mad(a, b) == 500.0 and MO.mse(a, b) == 250.0
    assert MO.grad(a, b) < 1e-20                                    # constants have no gradient (filter sums to 0 up to rounding)
    assert MO.dtssd(b, a, a, a) == pytest.approx(50.0)              # sqrt(mean(0.25)) * 1e2
    fg_p = np.zeros((3, 4, 6)); fg_t = np.ones((3, 4, 6)); m = np.zeros((4, 6)); m[1, 2] = 0.3
    assert MO.fgr_mse(fg_p, fg_t, m) == 1000.0
    out = MO.evaluate_clip(np.stack([a, b]), np.stack([a, a]))
    assert out["pha_dtssd"][0] == 0.0 and out["pha_mad"] == [0.0, 500.0]


def _clip(T, H, W, seed):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    true = np.stack([1 / (1 + np.exp(-(np.hypot(yy - H / 2 - 2 * t, xx - W / 2 + 3 * t) - min(H, W) / 4) / 2.5)) for t in range(T)])
    true = (np.round(true * 255) / 255).astype(np.float32)          # like an 8-bit matte / 255
    pred = np.clip(true + g.normal(0, 0.05, true.shape) * (true > 0.02) + 0.01, 0, 1).astype(np.float32)
    fg_t = g.random((T, 3, H, W), dtype=np.float32)
    fg_p = np.clip(fg_t + g.normal(0, 0.1, fg_t.shape), 0, 1).astype(np.float32)
    true[:, :3, :] = 0                                              # some pixels outside the foreground mask
    return pred, true, fg_p, fg_t


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 70, 101), (2, 32, 32), (2, 5, 7), (2, 270, 480)])
def test_hip_metrics_against_oracle(shape):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from robustvideomatting_amd import metrics as M
    T, H, W = shape
    pred, true, fg_p, fg_t = _clip(T, H, W, seed=H)
    want = MO.evaluate_clip(pred, true, fg_p, fg_t)
    d = lambda a: torch.from_numpy(a).cuda()
    got = M.evaluate_clip(d(pred), d(true), d(fg_p), d(fg_t))
    for k in want:
        assert np.allclose(got[k], want[k], rtol=2e-5, atol=1e-6), (k, got[k], want[k])   # fp32 sums vs float64
    # single-metric classes (same call surface as the reference's)
    assert float(M.MetricMAD()(d(pred[0]), d(true[0]))) == pytest.approx(want["pha_mad"][0], rel=2e-5)
    assert float(M.MetricMSE()(d(pred[0]), d(true[0]))) == pytest.approx(want["pha_mse"][0], rel=2e-5)
    assert float(M.MetricGRAD()(d(pred[0]), d(true[0]))) == pytest.approx(want["pha_grad"][0], rel=2e-5)
    assert float(M.MetricDTSSD()(d(pred[1]), d(pred[0]), d(true[1]), d(true[0]))) == pytest.approx(want["pha_dtssd"][1], rel=2e-5)
    # determinism
    again = M.evaluate_clip(d(pred), d(true), d(fg_p), d(fg_t))
    assert again == got


def test_metrics_refuse_cpu_tensors():
    from robustvideomatting_amd import metrics as M
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M.clip_sums(torch.zeros(1, 4, 4), torch.zeros(1, 4, 4))
