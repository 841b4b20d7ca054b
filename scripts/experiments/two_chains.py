#!/usr/bin/env python3
"""Experiment (GPU box): the configs[1] training step as ONE chain over B = 128 sets against TWO
independent chains over 64 sets each on two streams inside one captured graph (the sets of a batch
are independent up to the gradient sum: grad_scale 0.5 each, gradients added before Adam).  Asks
whether the latency-bound kernels of the step overlap when a second chain is there to fill in."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "point-cloud-audio_amd"))
import torch
import models
from pca_hip import _lib, trainer

dev = torch.device("cuda", 0)
B, N, din, d, h, m, Cn = 128, 512, 2, 128, 4, 16, 50
if os.environ.get("CFG") == "cfg4":
    B, N, din, d, h, m, Cn = 128, 4096, 3, 256, 8, 32, 10
torch.manual_seed(0)
net = models.ST(dim_input=din, num_outputs=1, dim_output=Cn, num_inds=m, dim_hidden=d, num_heads=h).to(dev)
X = torch.randn(B, N, din, device=dev)
y = torch.randint(0, Cn, (B,), device=dev)
L = _lib.lib()
mode = _lib.MODE_BF16


def adam(eng, mo, vo, sc):
    _lib.check(L.pca_adam_step(eng.flat.data_ptr(), eng.grads.data_ptr(), mo.data_ptr(), vo.data_ptr(),
                               eng.flat.numel(), 1e-3, 0.9, 0.999, 1e-8, 1e-3, 1.0, sc.data_ptr(), 1,
                               eng._stream()))


def timeit(g, reps=200):
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def state(n):
    return (torch.zeros(n, device=dev), torch.zeros(n, device=dev),
            torch.zeros(2, dtype=torch.int32, device=dev))


def one_chain():
    eng = trainer.STEngine(net, B, N, mode, training=True)
    mo, vo, sc = state(eng.flat.numel())
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        eng.fwd_bwd(X, y); adam(eng, mo, vo, sc)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.fwd_bwd(X, y); adam(eng, mo, vo, sc)
    return timeit(g)


def two_chains(nch=2):
    b = B // nch
    engs = [trainer.STEngine(net, b, N, mode, training=True) for _ in range(nch)]
    Xs = [X[i * b:(i + 1) * b].contiguous() for i in range(nch)]
    ys = [y[i * b:(i + 1) * b].contiguous() for i in range(nch)]
    mo, vo, sc = state(engs[0].flat.numel())
    side = [torch.cuda.Stream(dev) for _ in range(nch - 1)]

    def body():
        cur = torch.cuda.current_stream(dev)
        for s in side:
            s.wait_stream(cur)
        engs[0].fwd_bwd(Xs[0], ys[0], grad_scale=1.0 / nch)
        for i, s in enumerate(side):
            with torch.cuda.stream(s):
                engs[i + 1].fwd_bwd(Xs[i + 1], ys[i + 1], grad_scale=1.0 / nch)
        for i, s in enumerate(side):
            cur.wait_stream(s)
            engs[0].grads.add_(engs[i + 1].grads)
            engs[i + 1].grads.zero_()
        adam(engs[0], mo, vo, sc)
    s0 = torch.cuda.Stream(dev)
    s0.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s0):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    return timeit(g)


for rep in range(2):
    print(f"one chain  B={B}: {one_chain():.4f} ms/step")
    print(f"two chains 2x{B // 2}: {two_chains(2):.4f} ms/step")
    print(f"four chains 4x{B // 4}: {two_chains(4):.4f} ms/step")
