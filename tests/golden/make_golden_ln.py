#!/usr/bin/env python3
"""golden_ln.npz: MAB / ST with ln=True, from the REAL reference on CPU.

Run in the build container only: ``python tests/golden/make_golden_ln.py``.
set_transformer-master/modules.py:14-16,30,32 (LayerNorm variants; no caller of the reference
enables them, Code/models.py:31 passes ln=False) - the fixtures pin the optional path."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PCA_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REF, "set_transformer-master"))
sys.path.insert(0, os.path.join(REF, "Code"))
os.chdir(os.path.join(REF, "Code"))

import inputs as gi  # noqa: E402
import modules as ref_modules  # noqa: E402  (reference)
import models as ref_models  # noqa: E402    (reference)

torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def main():
    out = {}
    for ci, (name, B, nq, nk, dq, dk, d, h) in enumerate(gi.LN_MAB_CASES):
        torch.manual_seed(900 + ci)
        mab = ref_modules.MAB(dq, dk, d, h, ln=True)
        with torch.no_grad():          # non-trivial affine parameters
            for ln in (mab.ln0, mab.ln1):
                ln.weight.add_(0.3 * torch.randn_like(ln.weight))
                ln.bias.add_(0.2 * torch.randn_like(ln.bias))
        Q = torch.from_numpy(gi.randn(910 + ci, B, nq, dq)).requires_grad_(True)
        K = torch.from_numpy(gi.randn(920 + ci, B, nk, dk)).requires_grad_(True)
        G = torch.from_numpy(gi.randn(930 + ci, B, nq, d))
        Y = mab(Q, K)
        (Y * G).sum().backward()
        for k, v in mab.state_dict().items():
            out[f"{name}/p/{k}"] = npy(v)
        for k, v in mab.named_parameters():
            out[f"{name}/g/{k}"] = npy(v.grad)
        out[f"{name}/Y"] = npy(Y)
        out[f"{name}/dQ"] = npy(Q.grad)
        out[f"{name}/dK"] = npy(K.grad)
    # whole ST with ln=True
    name, B, N, din, d, h, m, C = gi.LN_ST_CASE
    torch.manual_seed(950)
    net = ref_models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                        num_heads=h, ln=True)
    X = torch.from_numpy(gi.pc_input(951, B, N, din)).requires_grad_(True)
    y = torch.from_numpy(gi.labels(952, B, C))
    logits = net(X)
    loss = torch.nn.CrossEntropyLoss()(logits, y)
    loss.backward()
    out[f"{name}/logits"] = npy(logits)
    out[f"{name}/loss"] = np.float64(loss.item())
    for k, v in net.state_dict().items():
        out[f"{name}/p/{k}"] = npy(v)
    for k, v in net.named_parameters():
        out[f"{name}/g/{k}"] = npy(v.grad)
    np.savez_compressed(os.path.join(HERE, "golden_ln.npz"), **out)
    print("golden_ln.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
