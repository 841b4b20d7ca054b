import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-audio_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import modules, pca_hip
from oracle import st_oracle as orc
from test_gpu_bf16 import _mab_params, MAB1_CASES
dev = torch.device("cuda", 0)
for case in MAB1_CASES:
    B, N, m, dq, d, h = case
    p = _mab_params(dq, d, d, seed=sum(case))
    g = torch.Generator().manual_seed(1 + sum(case))
    X = torch.randn(B, N, dq, generator=g)
    if dq <= 4: X[..., -1] = X[..., -1] * 3 - 9
    H = torch.randn(B, m, d, generator=g)
    G = torch.randn(B, N, d, generator=g)
    ref = orc.mab_backward(G, X, H, p, h)
    out = {}
    for mode in ("f32", "bf16"):
        mab = modules.MAB(dq, d, d, h).to(dev); mab.load_state_dict(p)
        Xd = X.to(dev).requires_grad_(dq > 4); Hd = H.to(dev).requires_grad_(True)
        pca_hip.set_mode(mode)
        Y = mab(Xd, Hd); (Y * G.to(dev)).sum().backward()
        pca_hip.set_mode("f32")
        res = {"dH": Hd.grad.cpu()}
        if dq > 4: res["dX"] = Xd.grad.cpu()
        for k, prm in mab.named_parameters(): res[k] = prm.grad.cpu()
        out[mode] = res
    print(case)
    for k in out["bf16"]:
        r = ref["dQ" if k == "dX" else "dK" if k == "dH" else k]
        sc = max(1.0, float(r.abs().max()))
        e32 = float((out["f32"][k] - r).abs().max()) / sc
        e16 = float((out["bf16"][k] - r).abs().max()) / sc
        rms = float((out["bf16"][k] - r).pow(2).mean().sqrt()) / sc
        print(f"   {k:14s} max|ref| {sc:8.2f}  f32 err {e32:.1e}  bf16 max err {e16:.1e} rms {rms:.1e}")
