"""Matting quality metrics on the GPU - counterpart of the reference's evaluator classes
(/root/reference/evaluation/evaluate_hr.py:176-235: MetricMAD, MetricMSE, MetricGRAD, MetricDTSSD, and the
masked foreground MSE of evaluate_worker :159-171).

The reference evaluates frame by frame with a handful of torch ops each (and kornia's filter2d for the gradient);
here one native call (rvm_metrics, hand-written HIP) reads every matte of a clip once and returns all the sums.
Inputs are device tensors in 0..1 (the evaluator's `image / 255`).  No CPU fallback.
"""
import torch

from . import native


def _f32c(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("%s must be a HIP device tensor (no CPU fallback)" % what)
    return t.to(torch.float32).contiguous()


def clip_sums(pred_pha, true_pha, pred_fgr=None, true_fgr=None, prev=None, sigma=1.4):
    """pred_pha / true_pha: [T,H,W] (or [T,1,H,W]); pred_fgr / true_fgr: [T,3,H,W] or None; prev: optional
    (pred [H,W], true [H,W]) frame in front of the clip for dtSSD.  Returns a [T,6] float64 device tensor of sums
    (layout: include/rvm_mi355x.h, rvm_metrics)."""
    pred, true = _f32c(pred_pha, "pred_pha"), _f32c(true_pha, "true_pha")
    if pred.ndim == 4:
        pred, true = pred[:, 0].contiguous(), true[:, 0].contiguous()
    if pred.ndim == 2:
        pred, true = pred[None], true[None]
    if pred.shape != true.shape or pred.ndim != 3:
        raise RuntimeError("pred_pha and true_pha must have the same [T,H,W] shape")
    T, H, W = pred.shape
    pf = tf = pp = tp = None
    if (pred_fgr is None) != (true_fgr is None):
        raise RuntimeError("pred_fgr and true_fgr come in pairs")
    if pred_fgr is not None:
        pf, tf = _f32c(pred_fgr, "pred_fgr"), _f32c(true_fgr, "true_fgr")
        if pf.ndim == 3:
            pf, tf = pf[None], tf[None]
        if tuple(pf.shape) != (T, 3, H, W) or pf.shape != tf.shape:
            raise RuntimeError("foregrounds must be [T,3,H,W]")
    if prev is not None:
        pp, tp = _f32c(prev[0], "prev pred"), _f32c(prev[1], "prev true")
        if tuple(pp.shape) != (H, W) or tuple(tp.shape) != (H, W):
            raise RuntimeError("prev frames must be [H,W]")
    ws = torch.empty(native.metrics_workspace(T, H, W), dtype=torch.uint8, device=pred.device)
    sums = torch.empty((T, 6), dtype=torch.float64, device=pred.device)
    ptr = lambda t: t.data_ptr() if t is not None else 0
    with torch.cuda.device(pred.device):
        native.metrics(ptr(pred), ptr(true), ptr(pp), ptr(tp), ptr(pf), ptr(tf), T, H, W, sigma, ws.data_ptr(),
                       sums.data_ptr(), torch.cuda.current_stream(pred.device).cuda_stream)
    return sums


def evaluate_clip(pred_pha, true_pha, pred_fgr=None, true_fgr=None, sigma=1.4):
    """The per-frame metric lists evaluate_worker builds for one clip (evaluate_hr.py:125-173):
    pha_mad, pha_mse, pha_grad, pha_dtssd (first frame 0) and, with foregrounds, fgr_mse."""
    s = clip_sums(pred_pha, true_pha, pred_fgr, true_fgr, None, sigma).cpu()
    T = s.shape[0]
    hw = float(true_pha.shape[-1] * true_pha.shape[-2])
    out = {
        "pha_mad": (s[:, 0] / hw * 1e3).tolist(),
        "pha_mse": (s[:, 1] / hw * 1e3).tolist(),
        "pha_grad": (s[:, 2] / 1000).tolist(),
        "pha_dtssd": [0.0] + ((s[1:, 3] / hw).sqrt() * 1e2).tolist() if T > 1 else [0.0],
    }
    if pred_fgr is not None:
        out["fgr_mse"] = (s[:, 4] / s[:, 5] * 1e3).tolist()         # nan when true_pha has no positive pixel, like the reference's mean of nothing
    return out


class MetricMAD:
    def __call__(self, pred, true):
        return clip_sums(pred, true)[0, 0] / pred.numel() * 1e3


class MetricMSE:
    def __call__(self, pred, true):
        return clip_sums(pred, true)[0, 1] / pred.numel() * 1e3


class MetricGRAD:
    def __init__(self, sigma=1.4):
        self.sigma = sigma

    def __call__(self, pred, true):
        return clip_sums(pred, true, sigma=self.sigma)[0, 2] / 1000


class MetricDTSSD:
    def __call__(self, pred_t, pred_tm1, true_t, true_tm1):
        s = clip_sums(pred_t, true_t, prev=(pred_tm1, true_tm1))
        return (s[0, 3] / true_t.numel()).sqrt() * 1e2
