"""Scratch: which synthetic-corpus settings does ST learn quickly?  (GPU box, product path)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/point-cloud-audio_amd", ROOT + "/tests/golden"]
import models, dataset
from pca_hip import trainer, _lib
from oracle import st_oracle as so

def corpus(C, step_cls, cpc, seconds, noise):
    xs, ys = [], []
    for c in range(C):
        for j in range(cpc):
            rng = np.random.Generator(np.random.PCG64(5000 + cpc * c + j))
            L = int(seconds * 44100)
            lev, tilt = c % 10, c // 10
            wn = rng.standard_normal(L)
            if tilt:                               # one-pole low-pass: spectral tilt
                a1 = 0.15 * tilt
                for _ in range(1):
                    wn = np.convolve(wn, [1 - a1, a1], mode="same")
            w = (0.5 * 10 ** (-0.15 * lev) * wn).astype(np.float32)
            s = so.stft_logmag(w, 1024, drop_nyquist=True)
            xs.append(s); ys.append(np.full(s.shape[1], c))
    return np.concatenate(xs, 1), np.concatenate(ys)

F = 512; farr = (np.linspace(0, 22050, F + 1) / 44100)[:F]
for (C, step_cls, cpc, seconds, noise, steps) in [(10, 1, 5, 0.5, 0.1, 1500), (50, 1, 5, 0.5, 0.1, 3000)]:
    x, y = corpus(C, step_cls, cpc, seconds, noise)
    ds = dataset.ESC_pc(x, y, farr, device="cuda")
    torch.manual_seed(77)
    net = models.ST(dim_input=2, num_outputs=1, dim_output=C, num_inds=16, dim_hidden=128, num_heads=4).cuda()
    tr = trainer.Trainer(net, ds, 128, mode=_lib.MODE_F32, seed=1)
    print("setting", C, step_cls, cpc, seconds, noise, "sets", x.shape[1], flush=True)
    for s in range(steps):
        tr.step()
        if (s + 1) % 250 == 0:
            l, k = tr.read_stats()
            print(f"  step {s+1} loss {l/(250*128):.4f} acc {k/(250*128):.3f}", flush=True)
