"""Drop-in for ``Code/utils.py``: parameter counting and the per-frame point sub-sampling of
the framewise (2-D) experiments, with the selection done on MI355X.

``pc_maxK`` / ``pc_randK`` (Code/utils.py:25-82) loop over the T frames in Python and
``argsort`` / ``permutation`` each one; here ONE launch (pca_subsample_points) selects the K
points of every frame, and the host only gathers the rows so that the returned arrays keep
the dtypes the reference returns (``x``'s and ``farr``'s own).  The ``*_replace`` variants
(Code/utils.py:86-108) return the float64 [N, T] array the reference builds with
``np.zeros``.  Random selections come from the counter-based device stream
(``seed``, call number) instead of the global numpy RNG: same distribution, reproducible.
"""
import itertools

import numpy as np
import torch

import pca_hip

__all__ = ["count_parameters", "pc_maxK", "pc_randK", "pc_maxK_replace", "pc_randK_replace"]

_draws = itertools.count(1)
FRAMES_PER_LAUNCH = 4096


def count_parameters(model):
    """Print the trainable parameters per tensor and return their total
    (Code/utils.py:7-20; a plain text table instead of prettytable's)."""
    rows = [(n, p.numel()) for n, p in model.named_parameters() if p.requires_grad]
    w = max([len(n) for n, _ in rows] + [7])
    print(f"{'Modules':<{w}}  Parameters")
    for n, c in rows:
        print(f"{n:<{w}}  {c}")
    total = sum(c for _, c in rows)
    print(f"Total Trainable Params: {total}")
    return total


def _select(x: np.ndarray, K: int, mode: int, seed: int, device=None) -> np.ndarray:
    """int64 [T, K]: the selected bin indices of every frame (column) of x [N, T]."""
    x = np.asarray(x)
    N, T = x.shape
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None \
        else torch.device(device)
    spec = torch.as_tensor(x).to(dev, torch.float32).t().contiguous().t()   # [N, T], frame rows
    ramp = torch.zeros(N, dtype=torch.float32, device=dev)                  # coordinates unused
    out = np.empty((T, K), dtype=np.int64)
    draw = next(_draws)
    for s in range(0, T, FRAMES_PER_LAUNCH):
        idx = torch.arange(s, min(s + FRAMES_PER_LAUNCH, T), dtype=torch.int64, device=dev)
        _, _, sel = pca_hip.subsample_points(spec, ramp, None, idx, K, mode, seed, draw,
                                             want_sel=True)
        out[s:s + idx.numel()] = sel.cpu().numpy()
    return out


def pc_maxK(x, farr, Kmax, device=None):
    """(subsampled_x [K, T], subsampled_x_fs [K, T]): the Kmax largest bins of every frame in
    descending order and their frequencies (Code/utils.py:25-53)."""
    x, farr = np.asarray(x), np.asarray(farr)
    sel = _select(x, Kmax, pca_hip.MAXK, 0, device)
    return np.take_along_axis(x, sel.T, axis=0), farr[sel.T]


def pc_randK(x, farr, Kmax, seed: int = 0, device=None):
    """As pc_maxK with Kmax randomly chosen bins per frame (Code/utils.py:56-82)."""
    x, farr = np.asarray(x), np.asarray(farr)
    sel = _select(x, Kmax, pca_hip.RANDK, seed, device)
    return np.take_along_axis(x, sel.T, axis=0), farr[sel.T]


def _replace(x: np.ndarray, sel: np.ndarray) -> np.ndarray:
    out = np.zeros(x.shape, dtype=np.float64)
    np.put_along_axis(out, sel.T, np.take_along_axis(x, sel.T, axis=0), axis=0)
    return out


def pc_maxK_replace(x, Kmax, device=None):
    """[N, T] float64: every frame with all but its Kmax largest bins zeroed
    (Code/utils.py:86-96)."""
    x = np.asarray(x)
    return _replace(x, _select(x, Kmax, pca_hip.MAXK, 0, device))


def pc_randK_replace(x, Kmax, seed: int = 0, device=None):
    """[N, T] float64: every frame with all but Kmax random bins zeroed (Code/utils.py:98-108)."""
    x = np.asarray(x)
    return _replace(x, _select(x, Kmax, pca_hip.RANDK, seed, device))
