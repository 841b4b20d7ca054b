#!/usr/bin/env python3
"""Cycle stamps inside k_isab1_fwd256 (library built with -DPCA_FWD_STAMPS): where a tile's time goes."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "point-cloud-audio_amd"))
import numpy as np
import torch
from pca_hip import _lib
dev = torch.device("cuda", 0)
d, h, m, B, N = 256, 8, 32, 128, 2048
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
for _ in range(3):
    _lib.check(L.pca_mab_fwd(C.byref(s), X.data_ptr(), H.data_ptr(), C.byref(pp), Y.data_ptr(), None, ws.data_ptr(), None))
torch.cuda.synchronize()
out = np.zeros(8 * 64 * 16, dtype=np.int64)
L.pca_debug_fwd_stamps.argtypes = [C.c_void_p]
rc = L.pca_debug_fwd_stamps(out.ctypes.data)
st = out.reshape(8, 64, 16)[:, :, :9].astype(np.float64)
names = ["top->wait", "wait->B0", "B0->stored", "stored->GEMM1", "GEMM1->attn+O", "attn->B1", "B1->GEMM2", "GEMM2->epi"]
tiles = slice(4, 28)
for w in (0, 3, 4, 7):
    d_ = np.diff(st[w, tiles, :], axis=1).mean(axis=0)
    per_tile = np.diff(st[w, tiles, 0]).mean()
    print(f"wave {w}: cycles per tile {per_tile:.0f}; " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, d_)))
