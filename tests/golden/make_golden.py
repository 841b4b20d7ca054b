#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference on CPU.

Run in the build container only (``/root/reference`` does not exist on the GPU
box):  ``python tests/golden/make_golden.py``

It imports the reference's own ``set_transformer-master/modules.py``,
``Code/models.py`` and ``Code/dataset.py`` unchanged, feeds them the seeded
inputs of ``inputs.py`` and stores inputs-that-cannot-be-regenerated, outputs
and gradients.  Only data is written: no reference source text is copied.
The two shipped ST checkpoints are read with ``weights_only=True`` and their
tensors re-saved as plain arrays.
"""
import glob
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
import dataset as ref_dataset  # noqa: E402  (reference)

torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def gen_mab():
    out = {}
    for ci, (name, B, nq, nk, dq, dk, d, h) in enumerate(gi.MAB_CASES):
        torch.manual_seed(100 + ci)
        mab = ref_modules.MAB(dq, dk, d, h)
        Q = torch.from_numpy(gi.randn(200 + ci, B, nq, dq)).requires_grad_(True)
        K = torch.from_numpy(gi.randn(300 + ci, B, nk, dk)).requires_grad_(True)
        G = torch.from_numpy(gi.randn(400 + ci, B, nq, d))
        Y = mab(Q, K)
        (Y * G).sum().backward()
        for k, v in mab.state_dict().items():
            out[f"{name}/p/{k}"] = npy(v)
        for k, v in mab.named_parameters():
            out[f"{name}/g/{k}"] = npy(v.grad)
        out[f"{name}/Y"] = npy(Y)
        out[f"{name}/dQ"] = npy(Q.grad)
        out[f"{name}/dK"] = npy(K.grad)
        # fp64 re-evaluation to calibrate tolerances
        mab64 = ref_modules.MAB(dq, dk, d, h).double()
        mab64.load_state_dict({k: v.double() for k, v in mab.state_dict().items()})
        out[f"{name}/Y64"] = npy(mab64(Q.detach().double(), K.detach().double()))
    np.savez(os.path.join(HERE, "golden_mab.npz"), **out)
    print("golden_mab.npz", len(out), "arrays")


def gen_st():
    out = {}
    for ci, (name, B, N, din, d, h, m, C, full) in enumerate(gi.ST_CASES):
        torch.manual_seed(500 + ci)
        net = ref_models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m,
                            dim_hidden=d, num_heads=h)
        X = torch.from_numpy(gi.pc_input(600 + ci, B, N, din)).requires_grad_(True)
        y = torch.from_numpy(gi.labels(700 + ci, B, C))
        logits = net(X)
        lg2 = logits if logits.dim() == 2 else logits.unsqueeze(0)
        loss = torch.nn.CrossEntropyLoss()(lg2, y)
        loss.backward()
        out[f"{name}/logits"] = npy(logits)
        out[f"{name}/loss"] = np.float64(loss.item())
        out[f"{name}/dX"] = npy(X.grad)
        sd = net.state_dict()
        if full:
            for k, v in sd.items():
                out[f"{name}/p/{k}"] = npy(v)
            for k, v in net.named_parameters():
                out[f"{name}/g/{k}"] = npy(v.grad)
        else:
            # big architecture: parameters are stored (needed to reproduce), grads
            # as norm + strided sub-sample
            for k, v in sd.items():
                out[f"{name}/p/{k}"] = npy(v)
            for k, v in net.named_parameters():
                g = npy(v.grad).reshape(-1)
                out[f"{name}/gnorm/{k}"] = np.float64(np.linalg.norm(g.astype(np.float64)))
                out[f"{name}/gsub/{k}"] = g[::gi.GRAD_SUBSAMPLE].copy()
        net64 = ref_models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m,
                              dim_hidden=d, num_heads=h).double()
        net64.load_state_dict({k: v.double() for k, v in sd.items()})
        out[f"{name}/logits64"] = npy(net64(X.detach().double()))
    np.savez(os.path.join(HERE, "golden_st.npz"), **out)
    print("golden_st.npz", len(out), "arrays")


def gen_ckpt():
    out = {}
    fst = glob.glob(os.path.join(REF, "Code", "model_saves", "FST*_net.pth"))[0]
    tst = glob.glob(os.path.join(REF, "Code", "model_saves", "3ST*_net.pth"))[0]
    for tag, path, din, Ns in (("fst", fst, 2, gi.CKPT_2D_N), ("tst", tst, 3, gi.CKPT_3D_N)):
        sd = torch.load(path, weights_only=True, map_location="cpu")
        net = torch.nn.DataParallel(ref_models.ST(dim_input=din, dim_hidden=64,
                                                  num_heads=8, num_inds=64))
        net.load_state_dict(sd)           # keys carry the 'module.' prefix
        net = net.module
        net64 = ref_models.ST(dim_input=din, dim_hidden=64, num_heads=8,
                              num_inds=64).double()
        net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
        for k, v in sd.items():
            out[f"{tag}/p/{k}"] = npy(v)   # stored WITH the module. prefix, as shipped
        for i, N in enumerate(Ns):
            X = torch.from_numpy(gi.pc_input(900 + 10 * din + i, 8, N, din))
            with torch.no_grad():
                out[f"{tag}/logits/{N}"] = npy(net(X))
                out[f"{tag}/logits64/{N}"] = npy(net64(X.double()))
    np.savez(os.path.join(HERE, "golden_ckpt.npz"), **out)
    print("golden_ckpt.npz", len(out), "arrays")


def gen_dataset():
    out = {}
    rng = np.random.Generator(np.random.PCG64(4242))
    F, T = 9, 5
    x = rng.normal(-9, 3, size=(F, T)).astype(np.float32)
    y = rng.integers(0, 10, size=(T,))
    farr = np.linspace(0, 44100 / 2, F) / 44100
    ds = ref_dataset.ESC_pc(x, y, farr)
    out["pc2d/x"], out["pc2d/y"], out["pc2d/farr"] = x, y, farr
    out["pc2d/len"] = np.int64(len(ds))
    for i in range(T):
        pc, lbl = ds[i]
        out[f"pc2d/item{i}"] = pc.numpy()
        out[f"pc2d/label{i}"] = lbl.numpy()
    F, Nt, S = 6, 4, 3
    x3 = rng.normal(-9, 3, size=(F, Nt, S)).astype(np.float32)
    y3 = rng.integers(0, 10, size=(S,))
    farr3 = np.linspace(0, 44100 / 2, F) / 44100
    tarr3 = np.linspace(0, ((0.5 * 1024) / 44100) * Nt, Nt)
    ds3 = ref_dataset.ESC_pc_temp(x3, y3, farr3, tarr3)
    out["pc3d/x"], out["pc3d/y"], out["pc3d/farr"], out["pc3d/tarr"] = x3, y3, farr3, tarr3
    out["pc3d/len"] = np.int64(len(ds3))
    for i in range(S):
        pc, lbl = ds3[i]
        out[f"pc3d/item{i}"] = pc.numpy()
        out[f"pc3d/label{i}"] = lbl.numpy()
    for K in (1, 7, F * Nt):
        dsk = ref_dataset.ESC_pc_temp_maxKSS(x3, y3, farr3, tarr3, K)
        for i in range(S):
            pc, lbl = dsk[i]
            out[f"pc3d/maxK{K}/item{i}"] = pc.numpy()
    np.savez(os.path.join(HERE, "golden_dataset.npz"), **out)
    print("golden_dataset.npz", len(out), "arrays")


def gen_train():
    """20 training steps of Code/settransformer.py:100-108 at a tiny cfg1-like
    architecture, fixed batches."""
    out = {}
    B, N, din, d, h, m, C, steps = 8, 32, 2, 32, 4, 8, 10, 20
    torch.manual_seed(1)
    net = ref_models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m,
                        dim_hidden=d, num_heads=h)
    for k, v in net.state_dict().items():
        out[f"p0/{k}"] = npy(v).copy()
    opt = torch.optim.Adam(net.parameters(), lr=1.0e-3, weight_decay=1.0e-3)
    crit = torch.nn.CrossEntropyLoss()
    losses = []
    for s in range(steps):
        X = torch.from_numpy(gi.pc_input(5000 + s, B, N, din))
        y = torch.from_numpy(gi.labels(6000 + s, B, C))
        loss = crit(net(X), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    out["losses"] = np.array(losses, dtype=np.float64)
    out["cfg"] = np.array([B, N, din, d, h, m, C, steps], dtype=np.int64)
    for k, v in net.state_dict().items():
        out[f"p20/{k}"] = npy(v)
    np.savez(os.path.join(HERE, "golden_train.npz"), **out)
    print("golden_train.npz losses", losses[0], "->", losses[-1])


if __name__ == "__main__":
    gen_mab()
    gen_st()
    gen_ckpt()
    gen_dataset()
    gen_train()
