"""Which tensor of the training workspace differs between the set-resident forward (PCA_SET128=1) and the
per-block launches (PCA_SET128=0)?  Runs both on the same batch and compares every saved area, tensor
by tensor (offsets from pca_st_ws_layout + the carve order of mab0_carve_saved / mab1_carve_saved)."""
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

B, N, din = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (5, 256, 2))]
d, h, m, Cc = 128, 4, 16, 50
dev = torch.device("cuda", 0)
torch.manual_seed(100 + N + din)
net = models.ST(dim_input=din, num_outputs=1, dim_output=Cc, num_inds=m, dim_hidden=d, num_heads=h).to(dev)
X = torch.from_numpy(gi.pc_input(7000 + N, B, N, din)).to(dev)
y = torch.from_numpy(gi.labels(7001 + N, B, Cc)).to(dev)


def a256(n):
    return (n + 255) & ~255


def run(flag):
    os.environ["PCA_SET128"] = flag
    eng = trainer.STEngine(net, B, N, _lib.MODE_BF16, training=True)
    eng.ws.zero_()
    eng.fwd_bwd(X, y, phase=-1)
    torch.cuda.synchronize()
    return eng


e0, e1 = run("0"), run("1")
lay = (C.c_int64 * 11)()
_lib.check(_lib.lib().pca_st_ws_layout(C.byref(e1.cfg), lay), "layout")
lay = list(lay)


def mab0_fields(dk, S, R=64, nq=16):
    Rp = (R + 31) // 32 * 32
    return [("Qp", nq * d, "f4"), ("Gf", Rp * dk, "f4"), ("Gb", Rp * dk, "bf"), ("GtP", Rp * dk, "bf"),
            ("T", B * R * dk, "f4"), ("LSE", B * R, "f4"), ("O", B * nq * d, "f4"), ("Z", B * nq * d, "f4"),
            ("WvT", dk * d, "f4"), ("WoT", d * d, "f4"), ("Tp", B * S * R * dk, "f4"), ("Mp", B * S * R, "f4"),
            ("Lp", B * S * R, "f4")]


def mab1_fields():
    kv = B * m * d
    return [("KpP", kv, "bf"), ("VpP", kv, "bf"), ("Kt", kv, "bf"), ("Vt", kv, "bf"), ("QpS", B * N * d, "bf"),
            ("OS", B * N * d, "bf"), ("mask", B * (N // 128) * 8 * 64, "u4")]


def view(ws, off, n, kind):
    nb = {"f4": 4, "bf": 2, "u4": 4}[kind]
    raw = ws[off:off + n * nb]
    if kind == "f4":
        return raw.view(torch.float32).double()
    if kind == "bf":
        return raw.view(torch.bfloat16).double()
    return raw.view(torch.int32).double()


def cmp_area(name, base, fields, skip=()):
    off = base
    for fn, n, kind in fields:
        if fn not in skip and n > 0:
            a, b_ = view(e0.ws, off, n, kind), view(e1.ws, off, n, kind)
            dmax = float((a - b_).abs().max())
            nbad = int(((a - b_).abs() > 1e-2 * max(1.0, float(a.abs().max()))).sum())
            first = int(torch.nonzero((a - b_).abs() > 1e-2 * max(1.0, float(a.abs().max())))[0]) if nbad else -1
            print(f"{name:10s}{fn:6s} n={n:9d} max|ref|={float(a.abs().max()):9.3e} max|d|={dmax:9.3e} bad={nbad} first={first}")
        off += a256(n * {"f4": 4, "bf": 2, "u4": 4}[kind])


S0 = 2 if N >= 256 else 1        # mab0_splits of the ISAB few-queries block at these sizes
cmp_area("enc0.mab0", lay[0], mab0_fields(din, 0), skip=("Tp", "Mp", "Lp", "WvT", "WoT", "GtP", "Gb"))
cmp_area("enc0.mab1", lay[1], mab1_fields(), skip=("QpS",))
cmp_area("H0", lay[5], [("H", B * m * d, "f4")])
cmp_area("Y0", lay[7], [("Y", B * N * d, "bf")])
cmp_area("enc1.mab0", lay[2], mab0_fields(d, S0), skip=("Tp", "Mp", "Lp", "WvT", "WoT"))
cmp_area("enc1.mab1", lay[3], mab1_fields())
cmp_area("H1", lay[6], [("H", B * m * d, "f4")])
cmp_area("Y1", lay[8], [("Y", B * N * d, "bf")])
print("logits max|d|", float((e0.logits - e1.logits).abs().max()), " timeouts", int(e1.ws[lay[9]:lay[9] + 4].view(torch.int32)[0]))
