#!/usr/bin/env python3
"""Trainer determinism: parameters after 12 steps of two hipGraph trainers, two eager trainers and
graph vs eager (same seed, same batches).  usage (GPU box): graph_vs_eager.py [f32|bf16]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "point-cloud-audio_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
import models
from pca_hip import _lib, trainer

dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["cfg2"])
ds, _ = bench.build_dataset(cfg, 4, dev, seed=0)


MODE = _lib.MODE_F32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else _lib.MODE_BF16


def run(graph, steps=12):
    torch.manual_seed(1)
    net = models.ST(dim_input=2, dim_output=50, num_inds=16, dim_hidden=128, num_heads=4).to(dev)
    tr = trainer.Trainer(net, ds, 128, mode=MODE, use_graph=graph, seed=1, keep_grads=True)
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    return tr.eng.flat.clone(), tr.eng.grads.clone()


g1, g2, e1, e2 = run(True), run(True), run(False), run(False)
for name, a, b in (("graph vs graph", g1, g2), ("eager vs eager", e1, e2), ("graph vs eager", g1, e1)):
    print(f"{name}: max |d params| {float((a[0] - b[0]).abs().max()):.3e}, "
          f"max |d grads| {float((a[1] - b[1]).abs().max()):.3e}")
for s in (1, 2):
    a, b = run(True, s), run(False, s)
    print(f"after {s} step(s) graph vs eager: params {float((a[0] - b[0]).abs().max()):.3e} "
          f"grads {float((a[1] - b[1]).abs().max()):.3e}")
