#!/usr/bin/env python3
"""s_memtime stamps inside k_isab1_fwd256_ab (library built with -DPCA_FWD_STAMPS): per role and wave,
where an iteration's cycles go.  Role A stamps: 0 iteration start, 1 after the input issue, 2 / 4 after
GEMM1 of point block 0 / 1, 3 / 5 after its attention, 6 before the barrier.  Role B: 0 = 1 start,
2 / 3 after GEMM2 of block 0 / 1, 4 / 5 after the epilogues, 6 after the stores (before the barrier)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "point-cloud-audio_amd"))
import numpy as np
import torch
from pca_hip import _lib
dev = torch.device("cuda", 0)
d, h, m, B, N = 256, 8, 32, 128, int(os.environ.get("N", "2048"))
g = torch.Generator().manual_seed(0)
params = []
for din in (d, d, d, d):
    params += [((torch.rand(d, din, generator=g) * 2 - 1) / din ** 0.5).to(dev),
               ((torch.rand(d, generator=g) * 2 - 1) / din ** 0.5).to(dev)]
L = _lib.lib()
X = torch.randn(B, N, d, generator=g).to(dev).to(torch.bfloat16)
H = torch.randn(B, m, d, generator=g).to(dev)
s = _lib.MabShape(B, N, m, d, d, d, h, 0, _lib.MODE_BF16, _lib.PCA_BF16, _lib.PCA_F32, _lib.PCA_BF16, None, 0)
Y = torch.empty(B, N, d, dtype=torch.bfloat16, device=dev)
ws = torch.empty(L.pca_mab_fwd_ws_bytes(C.byref(s)), dtype=torch.uint8, device=dev)
pp = _lib.MabParams(*[t.data_ptr() for t in params], None, None, None, None)
for _ in range(5):
    _lib.check(L.pca_mab_fwd(C.byref(s), X.data_ptr(), H.data_ptr(), C.byref(pp), Y.data_ptr(), None, ws.data_ptr(), None))
torch.cuda.synchronize()
out = np.zeros(16 * 32 * 16, dtype=np.int64)
L.pca_debug_fwd_stamps.argtypes = [C.c_void_p]
rc = L.pca_debug_fwd_stamps(out.ctypes.data)
st = out[:16 * 32 * 8].reshape(16, 32, 8)[:, :, :7].astype(np.float64)
sel = int(os.environ.get("PCA_AB_STAMPSEL", "15"))
if sel != 15:       # one stamp per run: cycles from the iteration start to stamp `sel`, per wave
    it = slice(6, 28)
    per = np.diff(st[:, it, 0], axis=1).mean(axis=1)
    d = ((st[:, it, sel] - st[:, it, 0]) % 2 ** 32).mean(axis=1)
    print(f"stamp {sel}: cycles/iteration A {per[:8].mean():.0f} B {per[8:].mean():.0f}; start -> stamp: " +
          "A " + " ".join(f"{v:.0f}" for v in d[:8]) + " | B " + " ".join(f"{v:.0f}" for v in d[8:]))
    sys.exit(0)
it = slice(6, 28)
names = ["0>1 issue/stores", "1>2 GEMM(0)", "2>3 attn/epi(0)", "3>4 GEMM(1)", "4>5 attn/epi(1)", "5>6 tail", "6>0' barrier"]
for w in range(16):
    seg = np.diff(st[w, it, :], axis=1).mean(axis=0)
    bar = (st[w, 7:29, 0] - st[w, 6:28, 6]).mean()
    per = np.diff(st[w, it, 0]).mean()
    print(f"{'A' if w < 8 else 'B'}{w % 8}: {per:6.0f} cyc/iter | " +
          " ".join(f"{v:5.0f}" for v in seg) + f" | barrier wait {bar:5.0f}")
print("columns:", ", ".join(names))
