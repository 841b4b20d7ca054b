"""Seeded synthetic inputs shared by make_golden.py (which ran the reference on
them) and by the tests (which regenerate them instead of storing them)."""
import numpy as np

# (name, B, nq, nk, dq, dk, d, h) -- MAB level cases
MAB_CASES = [
    ("tiny",     2, 5, 7, 3, 2, 8, 2),
    ("l1mab0",   2, 4, 51, 64, 2, 64, 8),     # shipped arch, layer-1 mab0 (dh=8)
    ("l1mab1",   2, 51, 4, 2, 64, 64, 8),     # shipped arch, layer-1 mab1
    ("dh32mab0", 3, 16, 33, 128, 128, 128, 4),
    ("dh32mab1", 3, 33, 16, 128, 128, 128, 4),
    ("onekey",   2, 3, 1, 16, 16, 16, 4),     # N = 1: softmax over a single key
    ("pma",      2, 1, 40, 128, 128, 128, 4),
]

# (name, B, N, din, d, h, m, C, full_grads) -- whole-ST cases
ST_CASES = [
    ("st_tiny",    4, 7, 2, 16, 4, 4, 10, True),
    ("st_shipped", 3, 51, 3, 64, 8, 64, 10, True),     # 3ST architecture
    ("st_cfg1",    4, 130, 2, 128, 4, 16, 50, True),   # BASELINE cfg1/2 architecture
    ("st_cfg4",    2, 70, 3, 256, 8, 32, 50, False),   # BASELINE cfg4 architecture
    ("st_b1",      1, 9, 2, 16, 2, 4, 10, True),       # B=1 -> squeeze() gives [C]
]

GRAD_SUBSAMPLE = 37     # stride of the stored sub-sample when full_grads is False


def pc_input(seed: int, B: int, N: int, din: int) -> np.ndarray:
    """Spectrogram-shaped point sets: f in [0,0.5], t in [0,0.116],
    logmag ~ clip(N(-9,3^2), -18.4, 0)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = np.empty((B, N, din), dtype=np.float32)
    if din == 2:
        x[:, :, 0] = np.linspace(0.0, 0.5, N, dtype=np.float64)[None, :]
    else:
        nt = 10 if N % 10 == 0 and N >= 10 else 1
        F = N // nt
        f = np.tile(np.linspace(0.0, 0.5, F), nt)
        t = np.repeat(np.linspace(0.0, 0.116, nt) if nt > 1 else np.zeros(1), F)
        x[:, :, 0] = f[None, :]
        x[:, :, 1] = t[None, :]
    mag = np.clip(rng.normal(-9.0, 3.0, size=(B, N)), -18.4, 0.0)
    x[:, :, din - 1] = mag
    return x


def randn(seed: int, *shape) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal(shape).astype(np.float32)


def labels(seed: int, B: int, C: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, C, size=(B,)).astype(np.int64)


CKPT_2D_N = [1, 51, 501, 1025]
CKPT_3D_N = [1, 2551, 5120, 10240]
