"""CPU restatement (numpy, float64) of the reference evaluator's metric classes.  TEST INFRASTRUCTURE ONLY: imported
by tests/ as the checker of robustvideomatting_amd.metrics; never by the product path.

Follows /root/reference/evaluation/evaluate_hr.py:
  MetricMAD :176-178, MetricMSE :181-183, MetricGRAD :186-227, MetricDTSSD :230-235, fgr_mse :159-171.
MetricGRAD calls kornia.filters.filter2d(img, kernel, border_type='replicate') (:198-199).  kornia is a third-party
dependency that is neither under /root/reference nor installed here (requirements: `kornia`, unpinned in
evaluation/); its published definition - cross-correlation (no kernel flip) of the edge-replicated image, kernel not
normalised - is restated below.  The low-resolution twin of the same evaluator does the identical step with
cv2.filter2D(..., borderType=cv2.BORDER_REPLICATE) (evaluate_lr.py:174-177), which is also a correlation, so the two
reference files agree on the semantics.  The reference holds no golden numbers for these metrics: PARITY UNPINNED
beyond this restatement and the hand-checked cases in tests/test_metrics.py.
"""
import numpy as np


def gauss_filter(sigma=1.4, epsilon=1e-2):
    """evaluate_hr.py:203-219."""
    half = int(np.ceil(sigma * np.sqrt(-2 * np.log(np.sqrt(2 * np.pi) * sigma * epsilon))))
    size = 2 * half + 1
    g = lambda x: np.exp(-x ** 2 / (2 * sigma ** 2)) / (sigma * np.sqrt(2 * np.pi))       # :222-223
    dg = lambda x: -x * g(x) / sigma ** 2                                                   # :226-227
    fx = np.zeros((size, size))
    for i in range(size):
        for j in range(size):
            fx[i, j] = g(i - half) * dg(j - half)
    fx = fx / np.sqrt((fx ** 2).sum())
    return fx, fx.T.copy()


def filter2d_replicate(img, k):
    """Cross-correlation with edge replication (kornia.filters.filter2d / cv2.filter2D semantics)."""
    r = k.shape[0] // 2
    p = np.pad(img.astype(np.float64), r, mode="edge")
    out = np.zeros(img.shape, dtype=np.float64)
    for i in range(k.shape[0]):
        for j in range(k.shape[1]):
            out += k[i, j] * p[i:i + img.shape[0], j:j + img.shape[1]]
    return out


def mad(pred, true):
    return np.abs(pred.astype(np.float64) - true).mean() * 1e3


def mse(pred, true):
    return ((pred.astype(np.float64) - true) ** 2).mean() * 1e3


def grad(pred, true, sigma=1.4):
    fx, fy = gauss_filter(sigma)
    gm = lambda im: np.sqrt(filter2d_replicate(im, fx) ** 2 + filter2d_replicate(im, fy) ** 2)   # :197-200
    return ((gm(true) - gm(pred)) ** 2).sum() / 1000                                               # :192-195


def dtssd(pred_t, pred_tm1, true_t, true_tm1):
    d = ((pred_t.astype(np.float64) - pred_tm1) - (true_t.astype(np.float64) - true_tm1)) ** 2
    return np.sqrt(d.sum() / true_t.size) * 1e2


def fgr_mse(pred_fgr, true_fgr, true_pha):
    """pred_fgr / true_fgr [3,H,W]; the reference indexes HWC images with the [H,W] mask (:169-171): same element set."""
    m = true_pha > 0
    return ((pred_fgr.astype(np.float64)[:, m] - true_fgr[:, m]) ** 2).mean() * 1e3


def evaluate_clip(pred_pha, true_pha, pred_fgr=None, true_fgr=None, sigma=1.4):
    """evaluate_worker (:125-173) for arrays [T,H,W] / [T,3,H,W]."""
    out = {k: [] for k in ("pha_mad", "pha_mse", "pha_grad", "pha_dtssd")}
    if pred_fgr is not None:
        out["fgr_mse"] = []
    for t in range(pred_pha.shape[0]):
        out["pha_mad"].append(mad(pred_pha[t], true_pha[t]))
        out["pha_mse"].append(mse(pred_pha[t], true_pha[t]))
        out["pha_grad"].append(grad(pred_pha[t], true_pha[t], sigma))
        out["pha_dtssd"].append(0.0 if t == 0 else dtssd(pred_pha[t], pred_pha[t - 1], true_pha[t], true_pha[t - 1]))
        if pred_fgr is not None:
            out["fgr_mse"].append(fgr_mse(pred_fgr[t], true_fgr[t], true_pha[t]))
    return out
