"""Phase stamps of k_set128_fwd (diagnostic build, -DPCA_SET_STAMPS): one cfg2 forward + backward,
then the wall-clock stamps of workgroup 0 (lane 0 of every wave), in microseconds since stamp 0."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-audio_amd"), os.path.join(ROOT, "tests"),
                os.path.join(ROOT, "tests", "golden")]
import inputs as gi
import models
from pca_hip import _lib, trainer

dev = torch.device("cuda", 0)
B, N, din, d, h, m, Cc = 128, 512, 2, 128, 4, 16, 50
torch.manual_seed(0)
net = models.ST(dim_input=din, num_outputs=1, dim_output=Cc, num_inds=m, dim_hidden=d, num_heads=h).to(dev)
X = torch.from_numpy(gi.pc_input(1, B, N, din)).to(dev)
y = torch.from_numpy(gi.labels(2, B, Cc)).to(dev)
eng = trainer.STEngine(net, B, N, _lib.MODE_BF16, training=True)
for _ in range(5):
    eng.fwd_bwd(X, y, phase=-1)
torch.cuda.synchronize()
L = C.CDLL(_lib.LIB_PATH)
out = (C.c_ulonglong * (16 * 32))()
assert L.pca_debug_set_stamps(out) == 0
st = np.array(out, dtype=np.float64).reshape(16, 32)
t0 = st[:, 0].min()
names = {0: "start", 1: "L1 attn", 2: "L1 merged", 3: "L1 mid", 4: "L1 mab1 chain", 5: "L1 O in sY", 6: "L1 gemm2",
         7: "L1 Y in sY", 8: "L1 Y stored", 9: "L2 attn", 10: "L2 merged", 11: "L2 mid", 12: "L2 mab1 chain",
         13: "L2 O in sY", 14: "L2 gemm2", 15: "L2 Y in sY", 16: "L2 Y stored", 17: "PMA scores", 18: "PMA barrier",
         19: "end", 20: "L2m: bar0", 21: "L2m: round A", 22: "L2m: round B", 23: "L2m: published",
         24: "L2m: partner flag", 25: "L2m: merged", 26: "L2m: loads landed"}
print("stamp                 wave0      min      max   (us since first start; 100 MHz clock)")
prev = 0.0
for i in [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 20, 21, 22, 23, 24, 26, 25, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19]:
    col = (st[:, i] - t0) / 100.0
    w0 = col[0]
    print(f"{i:2d} {names.get(i, ''):16s} {w0:8.2f} {col.min():8.2f} {col.max():8.2f}   d(w0) {w0 - prev:6.2f}")
    prev = w0
