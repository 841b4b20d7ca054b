#!/usr/bin/env python3
"""The many-queries block forward at the north-star shape (d=256, 8 heads, m=32 keys), bf16
activations at the ABI, B=128 sets: the single-launch wave-per-head kernel (PCA_D256_FUSED=0: the
Q phase + O phase pair).  FLOPs in the reference formulation.  GPU box; used
under rocprofv3 for the kernel-trace / PMC summaries in profiles/."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "point-cloud-audio_amd"))
import torch
from pca_hip import _lib

dev = torch.device("cuda", 0)
d, h, m, B = 256, 8, 32, 128
g = torch.Generator().manual_seed(0)
params = []
for din in (d, d, d, d):
    params += [((torch.rand(d, din, generator=g) * 2 - 1) / din ** 0.5).to(dev),
               ((torch.rand(d, generator=g) * 2 - 1) / din ** 0.5).to(dev)]
L = _lib.lib()
reps = int(os.environ.get("REPS", "10"))
for N in [int(x) for x in os.environ.get("NS", "2048,4096").split(",")]:
    X = torch.randn(B, N, d, generator=g).to(dev).to(torch.bfloat16)
    H = torch.randn(B, m, d, generator=g).to(dev)
    s = _lib.MabShape(B, N, m, d, d, d, h, 0, _lib.MODE_BF16, _lib.PCA_BF16, _lib.PCA_F32,
                      _lib.PCA_BF16, None, 0)
    Y = torch.empty(B, N, d, dtype=torch.bfloat16, device=dev)
    ws = torch.empty(L.pca_mab_fwd_ws_bytes(C.byref(s)), dtype=torch.uint8, device=dev)
    pp = _lib.MabParams(*[t.data_ptr() for t in params], None, None, None, None)
    call = lambda: _lib.check(L.pca_mab_fwd(C.byref(s), X.data_ptr(), H.data_ptr(), C.byref(pp),
                                            Y.data_ptr(), None, ws.data_ptr(), None))
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * B * N * (2 * d * d + 2 * m * d)
    print(f"mab1 fwd B={B} N={N} d={d} m={m} bf16 in/out (whole call): {us:8.1f} us "
          f"{fl / us / 1e6:7.1f} TFLOP/s = {100 * fl / us / 1e6 / 2500:.1f} % of 2.5 PF")
