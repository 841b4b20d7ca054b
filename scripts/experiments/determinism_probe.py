#!/usr/bin/env python3
"""Which gradient tensors of one STEngine forward + backward differ between two runs on the same
inputs?  (fp32 atomics make a reduction order-dependent.)  usage: determinism_probe.py [cfg4|cfg2]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "point-cloud-audio_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import inputs as gi
import models
from pca_hip import _lib, trainer

which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
B, N, din, d, h, m, C = (16, 4104, 3, 256, 8, 32, 50) if which == "cfg4" else (128, 512, 2, 128, 4, 16, 50)
dev = torch.device("cuda", 0)
torch.manual_seed(1)
net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d, num_heads=h).to(dev)
X = torch.from_numpy(gi.pc_input(2, B, N, din)).to(dev)
y = torch.from_numpy(gi.labels(3, B, C)).to(dev)
eng = trainer.STEngine(net, B, N, _lib.MODE_BF16, training=True)
runs = []
for r in range(3):
    eng.grads.zero_()
    eng.fwd_bwd(X, y, phase=-1)
    torch.cuda.synchronize()
    runs.append((eng.grads.clone(), float(eng.loss)))
off = 0
bad = 0
for k, prm in net.named_parameters():
    n = prm.numel()
    a, b, c = (r[0][off:off + n] for r in runs)
    nd = int(((a != b) | (a != c)).sum())
    if nd:
        bad += 1
        print(f"{k:34s} {nd:7d} / {n} elements differ, max |d| {float((a - b).abs().max()):.3e}")
    off += n
print("losses", [r[1] for r in runs])
print("tensors that differ:", bad)
