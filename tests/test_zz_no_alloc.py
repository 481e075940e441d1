"""GPU: the allocation-free entry point rvm_metrics_static against rvm_metrics, through the C ABI.

rvm_metrics_static keeps the per-tile partial sums in the library's static device array instead of a caller
workspace (include/rvm_mi355x.h).  Same kernels, fixed summation order => the sums must be bit-identical.
Two calls in a row, the second with a smaller input (the array keeps the first call's contents), then the
size limit.  Collected last (file name) so a failure here cannot hide another test.
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _clip(N, H, W, seed, fgr):
    g = torch.Generator().manual_seed(seed)
    pred = torch.rand((N, H, W), generator=g)
    true = (pred + 0.1 * torch.randn((N, H, W), generator=g)).clamp(0, 1)
    true[:, : H // 3] = 0.0                                   # a region with true_pha == 0 (masked foreground metric)
    out = [pred.cuda(), true.cuda()]
    if fgr:
        out += [torch.rand((N, 3, H, W), generator=g).cuda(), torch.rand((N, 3, H, W), generator=g).cuda()]
    else:
        out += [None, None]
    return out


def _both(L, N, H, W, seed, fgr, prev):
    pred, true, pf, tf = _clip(N, H, W, seed, fgr)
    pp = tp = None
    if prev:
        pp, tp = torch.rand((H, W), device="cuda"), torch.rand((H, W), device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.empty(L.rvm_metrics_workspace(N, H, W), dtype=torch.uint8, device="cuda")
    want = torch.full((N, 6), -1.0, dtype=torch.float64, device="cuda")
    got = torch.full((N, 6), -2.0, dtype=torch.float64, device="cuda")
    rc = L.rvm_metrics(P(pred), P(true), P(pp), P(tp), P(pf), P(tf), N, H, W, 1.4, P(ws), P(want), stream)
    assert rc == 0, L.rvm_last_error()
    torch.cuda.synchronize()
    # no allocation of ANY kind: torch's counter only sees its own caching allocator, so the device's free memory
    # (hipMemGetInfo) is compared as well - a hipMalloc inside the library would show there
    before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    rc = L.rvm_metrics_static(P(pred), P(true), P(pp), P(tp), P(pf), P(tf), N, H, W, 1.4, P(got), stream)
    assert rc == 0, L.rvm_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before
    return want.cpu(), got.cpu()


def test_static_scratch_matches_workspace_entry_point_twice():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from robustvideomatting_amd import native
    L = native.lib()
    assert L.rvm_metrics_static_capacity() == 4 << 20
    # first call: 5 frames 270x480 with foregrounds and a previous frame (135 tiles per frame)
    want, got = _both(L, 5, 270, 480, seed=1, fgr=True, prev=True)
    assert torch.isfinite(want).all() and (want[:, :4] > 0).all()
    assert torch.equal(want, got), (want, got)
    # second call, smaller input: stale partial sums of the first call stay in the array behind it
    want, got = _both(L, 2, 70, 97, seed=2, fgr=False, prev=False)
    assert torch.equal(want, got), (want, got)
    assert (got[:, 4:] == 0).all() and (got[0, 3] == 0)          # no foregrounds, no frame in front of frame 0
    # and a larger one again
    want, got = _both(L, 3, 540, 960, seed=3, fgr=True, prev=False)
    assert torch.equal(want, got), (want, got)


def test_static_scratch_rejects_oversize_and_bad_arguments():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from robustvideomatting_amd import native
    L = native.lib()
    x = torch.zeros((1, 64, 64), device="cuda")
    sums = torch.zeros((1, 6), dtype=torch.float64, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    # 22 frames of 3840x2160: 22 * 120 * 68 tiles * 24 B = 4.3 MB > 4 MiB - rejected before any pointer is touched
    assert L.rvm_metrics_workspace(22, 2160, 3840) > L.rvm_metrics_static_capacity()
    rc = L.rvm_metrics_static(P(x), P(x), None, None, None, None, 22, 2160, 3840, 1.4, P(sums), None)
    assert rc != 0 and b"scratch" in L.rvm_last_error()
    assert L.rvm_metrics_workspace(21, 2160, 3840) <= L.rvm_metrics_static_capacity()
    rc = L.rvm_metrics_static(P(x), P(x), None, None, None, None, 0, 64, 64, 1.4, P(sums), None)
    assert rc != 0
    rc = L.rvm_metrics_static(None, P(x), None, None, None, None, 1, 64, 64, 1.4, P(sums), None)
    assert rc != 0 and b"null" in L.rvm_last_error()
