#!/usr/bin/env python3
"""The d -> d ISAB at the north-star shape (N = 2048, d = 256, 8 heads, m = 32, B = 128, bf16
activations at the ABI) as a TRAINING unit: forward with the saved-for-backward blocks, then the
backward of mab1 (dX, dH) and of mab0 (dX accumulated, dI), every weight gradient included - the unit
SURVEY.md 8d prices at N (4 d^2 + 4 m d) + 3 m d^2 MACs forward, x 3 for forward + backward:
3.662 GFLOP per set.  Under rocprofv3 --kernel-trace --stats the kernel table gives the per-kernel
times; this script prints the HIP-event time of all launches (forward alone, forward + backward)."""
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
g0 = [torch.zeros_like(t) for t in p0]
g1 = [torch.zeros_like(t) for t in p1]
I = (torch.randn(m, d, generator=g) * 0.5).to(dev)
dI = torch.zeros_like(I)
X = torch.randn(B, N, d, generator=g).to(dev).to(torch.bfloat16)
dY = (torch.randn(B, N, d, generator=g) * 0.1).to(dev).to(torch.bfloat16)
dX = torch.empty_like(X)
s0 = _lib.MabShape(B, m, N, d, d, d, h, 1, _lib.MODE_BF16, _lib.PCA_F32, _lib.PCA_BF16, _lib.PCA_F32, None, 0)
s1 = _lib.MabShape(B, N, m, d, d, d, h, 0, _lib.MODE_BF16, _lib.PCA_BF16, _lib.PCA_F32, _lib.PCA_BF16, None, 0)
Hm = torch.empty(B, m, d, dtype=torch.float32, device=dev)
dH = torch.empty(B, m, d, dtype=torch.float32, device=dev)
Y = torch.empty(B, N, d, dtype=torch.bfloat16, device=dev)


def buf(n):
    return torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)


sv0, sv1 = buf(L.pca_mab_saved_bytes(C.byref(s0))), buf(L.pca_mab_saved_bytes(C.byref(s1)))
wf0, wf1 = buf(L.pca_mab_fwd_ws_bytes(C.byref(s0))), buf(L.pca_mab_fwd_ws_bytes(C.byref(s1)))
wb0, wb1 = buf(L.pca_mab_bwd_ws_bytes(C.byref(s0))), buf(L.pca_mab_bwd_ws_bytes(C.byref(s1)))
pp0 = _lib.MabParams(*[t.data_ptr() for t in p0], None, None, None, None)
pp1 = _lib.MabParams(*[t.data_ptr() for t in p1], None, None, None, None)
gg0 = _lib.MabGrads(*[t.data_ptr() for t in g0], None, None, None, None)
gg1 = _lib.MabGrads(*[t.data_ptr() for t in g1], None, None, None, None)


def fwd(train):
    _lib.check(L.pca_mab_fwd(C.byref(s0), I.data_ptr(), X.data_ptr(), C.byref(pp0), Hm.data_ptr(),
                             sv0.data_ptr() if train else None, wf0.data_ptr(), None))
    _lib.check(L.pca_mab_fwd(C.byref(s1), X.data_ptr(), Hm.data_ptr(), C.byref(pp1), Y.data_ptr(),
                             sv1.data_ptr() if train else None, wf1.data_ptr(), None))


def bwd():
    # modules.py:52-53: X is mab1's query (dX written) and mab0's key (dX accumulated)
    _lib.check(L.pca_mab_bwd(C.byref(s1), X.data_ptr(), Hm.data_ptr(), C.byref(pp1), sv1.data_ptr(),
                             dY.data_ptr(), dX.data_ptr(), dH.data_ptr(), 0, C.byref(gg1), wb1.data_ptr(), None))
    _lib.check(L.pca_mab_bwd(C.byref(s0), I.data_ptr(), X.data_ptr(), C.byref(pp0), sv0.data_ptr(),
                             dH.data_ptr(), dI.data_ptr(), dX.data_ptr(), 1, C.byref(gg0), wb0.data_ptr(), None))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


fl = 2.0 * B * (N * (4 * d * d + 4 * m * d) + 3 * m * d * d)
us_i = timed(lambda: fwd(False))
us_f = timed(lambda: fwd(True))
us_fb = timed(lambda: (fwd(True), bwd()))
assert torch.isfinite(dX.float()).all() and torch.isfinite(g1[0]).all()
for name, us, k in (("fwd (inference)", us_i, 1), ("fwd (training: saves)", us_f, 1), ("fwd + bwd", us_fb, 3)):
    print(f"ISAB(d->d) {name:22s} B={B} N={N} d={d} m={m} (all launches, HIP events): {us:8.1f} us "
          f"{k * fl / us / 1e6:7.1f} TFLOP/s = {100 * k * fl / us / 1e6 / 2500:.1f} % of 2.5 PF "
          f"({k * fl / B / 1e9:.3f} GFLOP per set)")
