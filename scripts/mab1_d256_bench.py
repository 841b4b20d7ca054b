#!/usr/bin/env python3
"""Forward of the many-queries block (ISAB's mab1) at the north-star shape d=256 / 8 heads /
m=32: fused bf16 kernels (Q phase + O phase) against the bf16 GEMM chain and the exact fp32
chain.  FLOPs in the reference formulation: 2 N B (2 d^2 + 2 m d).  GPU box."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "point-cloud-audio_amd"))
import torch
import pca_hip
from pca_hip import ops

dev = torch.device("cuda", 0)
d, h, m = 256, 8, 32
g = torch.Generator().manual_seed(0)
params = []
for din in (d, d, d, d):
    params += [((torch.rand(d, din, generator=g) * 2 - 1) / din ** 0.5).to(dev),
               ((torch.rand(d, generator=g) * 2 - 1) / din ** 0.5).to(dev)]


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for B, N in ((128, 2048), (128, 4096)):
    X = torch.randn(B, N, d, generator=g).to(dev)
    H = torch.randn(B, m, d, generator=g).to(dev)
    fl = 2.0 * B * N * (2 * d * d + 2 * m * d)
    out = {}
    for mode in ("bf16", "f32"):
        pca_hip.set_mode(mode)
        us = timed(lambda: ops.mab_infer(X, H, params, h))
        out[mode] = us
        print(f"mab1 fwd B={B} N={N} d={d} m={m} mode={mode:5s}: {us:9.1f} us  "
              f"{fl / us / 1e6:8.1f} TFLOP/s  ({100 * fl / us / 1e6 / 2500:.1f} % of 2.5 PF bf16 MFMA)")
    pca_hip.set_mode("f32")
    # the same pair with bf16 activations crossing the ABI (what the fused training path passes
    # between blocks at d = 128): C entry directly
    import ctypes as C
    from pca_hip import _lib
    L = _lib.lib()
    s = _lib.MabShape(B, N, m, d, d, d, h, 0, _lib.MODE_BF16, _lib.PCA_BF16, _lib.PCA_F32,
                      _lib.PCA_BF16, None, 0)
    Xb = X.to(torch.bfloat16)
    Yb = torch.empty(B, N, d, dtype=torch.bfloat16, device=dev)
    ws = torch.empty(L.pca_mab_fwd_ws_bytes(C.byref(s)), dtype=torch.uint8, device=dev)
    pp = _lib.MabParams(*[t.data_ptr() for t in params], None, None, None, None)
    us = timed(lambda: _lib.check(L.pca_mab_fwd(C.byref(s), Xb.data_ptr(), H.data_ptr(),
                                                C.byref(pp), Yb.data_ptr(), None, ws.data_ptr(),
                                                None)))
    print(f"mab1 fwd B={B} N={N} d={d} m={m} bf16 in/out  : {us:9.1f} us  "
          f"{fl / us / 1e6:8.1f} TFLOP/s  ({100 * fl / us / 1e6 / 2500:.1f} % of 2.5 PF bf16 MFMA)")
