#!/usr/bin/env python3
"""Times pca_gemm_bf16 on the shapes the exact chain issues at BASELINE configs[3]
(B=128 sets x N=4096 points, d=256, 8 heads of 32, m=32).  Run on the GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "point-cloud-audio_amd"))
import torch
from pca_hip import _lib

L = _lib.lib()
dev = torch.device("cuda", 0)
R, d, h, m, dh, Bn, Np = 128 * 4096, 256, 8, 32, 32, 128, 4096


def run(name, desc, A, B, bias, Cm, bytes_moved, reps=10):
    for _ in range(2):
        _lib.check(L.pca_gemm_bf16(C.byref(desc), A.data_ptr(), B.data_ptr(),
                                   bias.data_ptr() if bias is not None else None, Cm.data_ptr(), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(L.pca_gemm_bf16(C.byref(desc), A.data_ptr(), B.data_ptr(),
                                   bias.data_ptr() if bias is not None else None, Cm.data_ptr(), None))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * desc.M * desc.N * desc.K * max(desc.nb1, 1) * max(desc.nb2, 1)
    print(f"{name:34s} {us:8.1f} us  {bytes_moved / us / 1e3:7.0f} GB/s  {fl / us / 1e6:7.1f} TFLOP/s")


X = torch.randn(R, d, device=dev)
W = torch.randn(d, d, device=dev)
b = torch.randn(d, device=dev)
Y = torch.empty(R, d, device=dev)
G = _lib.GemmDesc
big = 2 * R * d * 4
run("linear fwd  [R,256]x[256,256]^T", G(R, d, d, d, 1, 1, d, d, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 1.0), X, W, b, Y, big)
run("linear dX   [R,256]x[256,256]", G(R, d, d, d, 1, d, 1, d, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 1.0), X, W, None, Y, big)
run("linear fwd again", G(R, d, d, d, 1, 1, d, d, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 1.0), X, W, b, Y, big)
run("linear dX accumulate", G(R, d, d, d, 1, d, 1, d, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1, 1.0), X, W, None, Y, big * 1.5)
run("linear fwd accumulate", G(R, d, d, d, 1, 1, d, d, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1, 1.0), X, W, None, Y, big * 1.5)
dW = torch.zeros(d, d, device=dev)
run("linear dW   [256,R]x[R,256] acc", G(d, d, R, 1, d, d, 1, d, 1, 1, 0, 0, 0, 0, 0, 0, 1, 0, 1.0), X, Y, None, dW, big)
X3 = torch.randn(R, 3, device=dev); W3 = torch.randn(d, 3, device=dev)
run("layer-1 fwd [R,3]x[256,3]^T", G(R, d, 3, 3, 1, 1, 3, d, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 1.0), X3, W3, b, Y, R * d * 4)
# attention of the many-queries block, per (set, head): S = Q_h K_h^T, O_h = S V_h
Q = X.view(Bn, Np, d); Kp = torch.randn(Bn, m, d, device=dev)
S = torch.empty(Bn, h, Np, m, device=dev)
att = R * dh * h * 4 + Bn * h * Np * m * 4
run("scores 1024x [4096,32]x[32,32]^T", G(Np, m, dh, d, 1, 1, d, m, Bn, h, Np * d, dh, m * d, dh, h * Np * m, Np * m, 0, 1, 1.0), Q, Kp, None, S, att)
O = torch.empty(Bn, Np, d, device=dev)
run("A.V    1024x [4096,32]x[32,32]", G(Np, dh, m, m, 1, d, 1, d, Bn, h, h * Np * m, Np * m, m * d, dh, Np * d, dh, 0, 1, 1.0), S, Kp, None, O, att)
dV = torch.empty(Bn, m, d, device=dev)
run("dV     1024x [32,4096]x[4096,32]", G(m, dh, Np, 1, m, d, 1, d, Bn, h, h * Np * m, Np * m, Np * d, dh, m * d, dh, 0, 1, 1.0), S, Q, None, dV, att)
# few-queries block: S0 = Qp_h X_h^T  [32, 4096] per (set, head)
S0 = torch.empty(Bn, h, m, Np, device=dev)
run("scores0 1024x [32,32]x[4096,32]^T", G(m, Np, dh, d, 1, 1, d, Np, Bn, h, m * d, dh, Np * d, dh, h * m * Np, m * Np, 0, 1, 1.0), Kp, Q, None, S0, att)
