"""Micro-benchmark of single MAB entry points with the library's HIP-event hook."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-audio_amd")]
import torch
import modules, pca_hip
from pca_hip import _lib
L = pca_hip.lib()
dev = torch.device("cuda", 0)

def prof(kid, fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    _lib.check(L.pca_prof_start(kid, 100000))
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    _lib.check(L.pca_prof_stop(C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
    us = ms.value * 1e3 / max(n.value, 1)
    return us, fl.value / max(ms.value, 1e-9) / 1e9, by.value / max(ms.value, 1e-9) / 1e6, n.value

for (B, N, m, dq, d, h) in [(128, 512, 16, 128, 128, 4), (128, 512, 16, 2, 128, 4), (32, 2048, 16, 128, 128, 4),
                            (128, 2048, 16, 128, 128, 4), (512, 512, 16, 128, 128, 4)]:
    mab = modules.MAB(dq, d, d, h).to(dev)
    X = torch.randn(B, N, dq, device=dev); H = torch.randn(B, m, d, device=dev)
    pca_hip.set_mode("bf16")
    with torch.no_grad():
        us, tf, gbs, n = prof(_lib.K_MAB1_FWD, lambda: mab(X, H))
    print(f"mab1_fwd bf16 B={B} N={N} m={m} dq={dq} d={d}: {us:8.2f} us  {tf:8.1f} TFLOP/s  {gbs:7.1f} GB/s (alg)  launches {n}")
    pca_hip.set_mode("f32")
