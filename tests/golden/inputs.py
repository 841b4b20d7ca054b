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


# --------------------------------------------------------------------------- #
# accuracy-parity corpus (SURVEY.md section 8d "Accuracy parity")               #
# --------------------------------------------------------------------------- #
ACC = dict(C=10, clips_per_class=4, test_clips_per_class=5, seconds=0.5, fs=44100, n_fft=1024, B=128, epochs=46,
           d=128, h=4, m=16, lr=1e-3, wd=1e-3, seed=4242, init_seed=77)


ACC_EVAL_EVERY = 5     # epochs between test evaluations of the accuracy-parity run


def level_clip(clip_id: int, cls: int, seconds: float, fs: int) -> np.ndarray:
    """Class-conditional clip whose class is readable from set-level statistics: white noise
    at -3 dB per class index.  (SURVEY.md 8d's harmonic clips keep the reference ST on the
    ln(C) loss plateau for thousands of steps - measured - which makes an accuracy comparison
    vacuous; the loudness classes are learnt within a few hundred steps.)"""
    rng = np.random.Generator(np.random.PCG64(5000 + clip_id))
    L = int(round(seconds * fs))
    return (0.5 * 10.0 ** (-0.15 * cls) * rng.standard_normal(L)).astype(np.float32)


def accuracy_corpus():
    """ESC-shaped synthetic corpus at the cfg1/2 framing (n_fft 1024, hop 512, Nyquist
    dropped: F = 512 points per set, one set per frame).  Clip j of class c is
    ``level_clip(clip_id=5c+j, cls=c)`` for j < 4 (training); the test clips of class c are
    ``level_clip(1000+5c+j, c)``, j < 5: a split by clip as Code/data_processing.py:40-65
    makes, with a test set large enough (2200 sets) to resolve 0.05 % of accuracy.  Built with the CPU oracle's
    float64 STFT so that the reference (fixture time) and the HIP path (test time) train on
    identical arrays.  Returns dict(x_train [F,T], y_train [T], x_test, y_test, farr [F])."""
    from oracle import st_oracle as so
    a = ACC
    xs = {"train": [], "test": []}
    ys = {"train": [], "test": []}
    for c in range(a["C"]):
        for part, n, first in (("train", a["clips_per_class"], 5 * c),
                               ("test", a["test_clips_per_class"], 1000 + 5 * c)):
            for j in range(n):
                w = level_clip(first + j, c, a["seconds"], a["fs"])
                s = so.stft_logmag(w, a["n_fft"], drop_nyquist=True)
                xs[part].append(s)
                ys[part].append(np.full(s.shape[1], c, dtype=np.int64))
    F = a["n_fft"] // 2
    farr = (np.linspace(0, a["fs"] / 2, F + 1) / a["fs"])[:F]
    return dict(x_train=np.concatenate(xs["train"], axis=1), y_train=np.concatenate(ys["train"]),
                x_test=np.concatenate(xs["test"], axis=1), y_test=np.concatenate(ys["test"]),
                farr=farr)


def agreement_sets(seed: int, n: int, N: int, din: int, chunk: int = 100):
    """Yield (start, X[chunk, N, din]) blocks of the >= 10 000 seeded synthetic sets of the
    weights->predictions agreement test; block i uses seed ``seed + i``."""
    i = 0
    for s in range(0, n, chunk):
        b = min(chunk, n - s)
        yield s, pc_input(seed + i, b, N, din)
        i += 1


# --------------------------------------------------------------------------- #
# sub-sampling datasets at the shipped 3ST framing                             #
# --------------------------------------------------------------------------- #
SS_K = [1, 51, 2551, 5120]
IMP_WINF = [5, 8]


def ss_inputs(F: int = 512, Nt: int = 10, S: int = 3, seed: int = 8080):
    """x [F, Nt, S] float32 log-magnitudes, y [S], farr [F], tarr [Nt] as
    Code/settransformertemp.py:40-41 builds them.  5120 float32 draws contain a few exactly
    equal values; the reference orders such ties by numpy's unstable introsort, so the tests
    compare selections up to permutations inside runs of equal values."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.normal(-9.0, 3.0, size=(F, Nt, S)).astype(np.float32)
    y = rng.integers(0, 10, size=(S,)).astype(np.int64)
    farr = (np.linspace(0, 44100 / 2, F + 1) / 44100)[:F]
    tarr = np.linspace(0, ((0.5 * 1024) / 44100) * Nt, Nt)
    return x, y, farr, tarr


# --------------------------------------------------------------------------- #
# LayerNorm variants (MAB(ln=True)): modules.py:14-16,30,32                      #
# --------------------------------------------------------------------------- #
LN_MAB_CASES = [    # name, B, nq, nk, dq, dk, d, h
    ("ln_tiny", 2, 5, 7, 3, 2, 8, 2),
    ("ln_mab0", 3, 16, 33, 128, 128, 128, 4),
    ("ln_mab1", 2, 51, 4, 2, 64, 64, 8),
]
LN_ST_CASE = ("ln_st", 3, 40, 2, 32, 4, 8, 10)     # name, B, N, din, d, h, m, C


# ---- comparison baselines + metadata helpers (make_golden_base.py / test_boundary_names.py) ----
BASE_FF_DIMS = [33, 17, 8]
BASE_NCLASS = 5
BASE_NT, BASE_NF = 4, 24
BASE_CNN_DIMS = [20, 12, 6]       # Conv2d kernel (Nt, Nf + 1 - 20) = (4, 5)
BASE_K = [1, 7, 40]


def base_ff_input() -> np.ndarray:
    return randn(4101, 6, BASE_FF_DIMS[0])


def base_cnn_input() -> np.ndarray:
    return randn(4102, 3, BASE_NT, BASE_NF)


def base_dataset_inputs():
    """x2 [N=9, T=5], y2 int[5]; x3 [N=8, Nt=5, T=4], y3 int[4] (x3 has repeated values so that
    the top-K boundary meets ties)."""
    x2 = randn(4103, 9, 5)
    y2 = np.array([3, 1, 4, 1, 5], dtype=np.int64)
    x3 = np.round(randn(4104, 8, 5, 4) * 4.0) / 4.0
    y3 = np.array([2, 7, 1, 8], dtype=np.int64)
    return x2, y2, x3.astype(np.float32), y3


def base_csv_text() -> str:
    """A synthetic esc50.csv: 6 categories (4 of them in ESC-10) x 7 files, shuffled rows."""
    cats = ["dog", "rain", "sea_waves", "rooster", "cat", "door_knock"]
    rows = []
    for ci, c in enumerate(cats):
        for k in range(7):
            rows.append((f"{1 + k % 5}-{100000 + 37 * ci + k}-A-{ci}.wav", 1 + k % 5, ci, c,
                         c in ("dog", "rain", "sea_waves", "rooster"), f"{100000 + 37 * ci + k}",
                         "A"))
    order = np.random.Generator(np.random.PCG64(4105)).permutation(len(rows))
    lines = ["filename,fold,target,category,esc10,src_file,take"]
    for i in order:
        r = rows[i]
        lines.append(f"{r[0]},{r[1]},{r[2]},{r[3]},{r[4]},{r[5]},{r[6]}")
    return "\n".join(lines) + "\n"
