#!/usr/bin/env python3
"""The d -> d ISAB forward at the north-star shape (N = 2048, d = 256, 8 heads, m = 32, B = 128, bf16
activations at the ABI): mab0 (few-queries block: fc_k / fc_v over the keys + attention + epilogue,
then mab1's K / V projections of H) followed by mab1 (many-queries block).  Under rocprofv3 the kernel
trace gives the per-kernel times; FLOPs in the reference formulation:
ISAB(d -> d) = 2 * (N (4 d^2 + 4 m d) + 3 m d^2) per set."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "point-cloud-audio_amd"))
import torch
from pca_hip import _lib

dev = torch.device("cuda", 0)
d, h, m, B = 256, 8, 32, 128
N = int(os.environ.get("N", "2048"))
reps = int(os.environ.get("REPS", "10"))
g = torch.Generator().manual_seed(0)


def params():
    out = []
    for din in (d, d, d, d):
        out += [((torch.rand(d, din, generator=g) * 2 - 1) / din ** 0.5).to(dev),
                ((torch.rand(d, generator=g) * 2 - 1) / din ** 0.5).to(dev)]
    return out


L = _lib.lib()
p0, p1 = params(), params()
I = (torch.randn(m, d, generator=g) * 0.5).to(dev)
X = torch.randn(B, N, d, generator=g).to(dev).to(torch.bfloat16)
s0 = _lib.MabShape(B, m, N, d, d, d, h, 1, _lib.MODE_BF16, _lib.PCA_F32, _lib.PCA_BF16, _lib.PCA_F32, None, 0)
s1 = _lib.MabShape(B, N, m, d, d, d, h, 0, _lib.MODE_BF16, _lib.PCA_BF16, _lib.PCA_F32, _lib.PCA_BF16, None, 0)
Hm = torch.empty(B, m, d, dtype=torch.float32, device=dev)
Y = torch.empty(B, N, d, dtype=torch.bfloat16, device=dev)
ws0 = torch.empty(L.pca_mab_fwd_ws_bytes(C.byref(s0)), dtype=torch.uint8, device=dev)
ws1 = torch.empty(L.pca_mab_fwd_ws_bytes(C.byref(s1)), dtype=torch.uint8, device=dev)
pp0 = _lib.MabParams(*[t.data_ptr() for t in p0], None, None, None, None)
pp1 = _lib.MabParams(*[t.data_ptr() for t in p1], None, None, None, None)


def isab():
    _lib.check(L.pca_mab_fwd(C.byref(s0), I.data_ptr(), X.data_ptr(), C.byref(pp0), Hm.data_ptr(),
                             None, ws0.data_ptr(), None))
    _lib.check(L.pca_mab_fwd(C.byref(s1), X.data_ptr(), Hm.data_ptr(), C.byref(pp1), Y.data_ptr(),
                             None, ws1.data_ptr(), None))


for _ in range(3):
    isab()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    isab()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
fl = 2.0 * B * (N * (4 * d * d + 4 * m * d) + 3 * m * d * d)
print(f"ISAB(d->d) fwd B={B} N={N} d={d} m={m} (all launches, HIP events): {us:8.1f} us "
      f"{fl / us / 1e6:7.1f} TFLOP/s = {100 * fl / us / 1e6 / 2500:.1f} % of 2.5 PF")
