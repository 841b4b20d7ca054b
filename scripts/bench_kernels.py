"""Micro-benchmark of the fused engine at several sizes (HIP-event hook per kernel class)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-audio_amd")]
import torch
import models, pca_hip
from pca_hip import _lib, trainer
L = pca_hip.lib()
dev = torch.device("cuda", 0)
KN = {_lib.K_MAB1_FWD: "mab1_fwd", _lib.K_MAB1_BWD: "mab1_bwd", _lib.K_MAB0_FWD: "mab0_attn",
      _lib.K_MAB0_BWD: "mab0_bwd", _lib.K_WGRAD: "wgrad128"}
for (B, N) in [(128, 512), (512, 512), (128, 2048), (512, 2048)]:
    torch.manual_seed(0)
    net = models.ST(dim_input=2, dim_output=50, num_inds=16, dim_hidden=128, num_heads=4).to(dev)
    eng = trainer.STEngine(net, B, N, mode=_lib.MODE_BF16, training=True)
    X = torch.randn(B, N, 2, device=dev); y = torch.randint(0, 50, (B,), device=dev)
    for _ in range(3): eng.grads.zero_(); eng.fwd_bwd(X, y)
    torch.cuda.synchronize()
    line = f"B={B:4d} N={N:5d}: "
    for kid, nm in KN.items():
        _lib.check(L.pca_prof_start(kid, 10000))
        for _ in range(10): eng.grads.zero_(); eng.fwd_bwd(X, y)
        torch.cuda.synchronize()
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(L.pca_prof_stop(C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        if n.value:
            line += f"{nm} {ms.value*1e3/n.value:7.1f}us {fl.value/ms.value/1e9:6.0f}TF | "
    print(line)
