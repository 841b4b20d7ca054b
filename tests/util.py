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


def close_robust(a, b, tol, what="", outlier_frac=2e-4, cap=20.0):
    """bf16 comparisons: rms error over ALL elements <= tol/2, all but a fraction `outlier_frac`
    of the elements within tol, and no element off by more than cap * tol (all relative to
    max(1, max|b|)).  The outlier allowance covers the few elements whose ReLU pre-activation
    lies within rounding distance of zero, where the (discontinuous) derivative legitimately
    differs between two evaluations; the rms over everything and the cap keep those bounded."""
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.all(np.isfinite(a)), f"{what}: non-finite values"
    ref = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    d = np.abs(a - b) / ref
    rms = float(np.sqrt(np.mean(d ** 2))) if d.size else 0.0
    bad = float(np.mean(d > tol)) if d.size else 0.0
    worst = float(d.max()) if d.size else 0.0
    assert rms <= tol / 2, f"{what}: rms {rms:.3e} > {tol / 2:.1e}"
    assert bad <= outlier_frac, f"{what}: {bad:.2e} of elements off by > {tol:.1e} (max {worst:.2e})"
    assert worst <= cap * tol, f"{what}: max {worst:.2e} > {cap:g} * {tol:.1e}"
    return worst


def same_up_to_ties(sel, ref_sel, vals, what=""):
    """Selections agree except for permutations inside runs of equal values (which the
    reference orders by an unstable sort); a run cut by the K boundary may pick different
    members, so there only the values are compared."""
    sel, ref_sel, vals = np.asarray(sel), np.asarray(ref_sel), np.asarray(vals)
    assert sel.shape == ref_sel.shape == vals.shape, (what, sel.shape, ref_sel.shape)
    K = len(vals)
    q = 0
    while q < K:
        e = q + 1
        while e < K and vals[e] == vals[q]:
            e += 1
        if e < K or e - q == 1:
            if e - q == 1:
                ok = sel[q] == ref_sel[q] or (e == K)     # last single may be a cut tie
            else:
                ok = sorted(sel[q:e]) == sorted(ref_sel[q:e])
            assert ok, f"{what}: rows {q}..{e} differ: {sel[q:e]} vs {ref_sel[q:e]}"
        q = e
