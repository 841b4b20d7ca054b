#!/usr/bin/env python3
"""Forward-only throughput of the ST engine (pca_st_forward) - the eval loop of Code/pceval.py:86-97 -
at a bench.py config, per mode.  usage: infer_bench.py [config] [batch]   (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-audio_amd")]
import torch

import bench
import models
from pca_hip import _lib, trainer

name = sys.argv[1] if len(sys.argv) > 1 else "fst"
cfg = dict(bench.CONFIGS[name])
B = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["B"]
dev = torch.device("cuda", 0)
n_fft, ntemp = cfg["n_fft"], cfg["ntemp"]
F = n_fft // 2 + 1 if (cfg["din"] == 2 and n_fft == 2048) else n_fft // 2
N = F * ntemp
torch.manual_seed(1)
net = models.ST(dim_input=cfg["din"], num_outputs=1, dim_output=cfg["C"], num_inds=cfg["m"],
                dim_hidden=cfg["d"], num_heads=cfg["h"]).to(dev)
g = torch.Generator().manual_seed(0)
X = torch.randn(B, N, cfg["din"], generator=g)
X[..., -1] = X[..., -1] * 3 - 9
X = X.to(dev)
for mode, md in (("f32", _lib.MODE_F32), ("bf16", _lib.MODE_BF16)):
    eng = trainer.STEngine(net, B, N, md, training=False)
    for _ in range(3):
        eng.forward(X)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.forward(X)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    fl = 2.0 * bench.st_fwd_macs(N, cfg["din"], cfg["d"], cfg["m"], 1, cfg["C"]) * B
    print(f"{name} inference B={B} N={N} mode {mode}: {ms:8.3f} ms/forward  {B / ms * 1e3:10.0f} sets/s  "
          f"{fl / ms / 1e9:8.2f} TFLOP/s (reference formulation)")
