import numpy as np
import torch


def T(a, device=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device) if device is not None else t


def maxerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def close(a, b, tol, what=""):
    """max|a-b| <= tol * max(1, max|b|)"""
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.all(np.isfinite(a)), f"{what}: non-finite values"
    err = maxerr(a, b)
    ref = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    assert err <= tol * ref, f"{what}: max|d|={err:.3e} > {tol:.1e}*{ref:.3g}"
    return err
