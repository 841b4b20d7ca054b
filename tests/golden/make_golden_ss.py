#!/usr/bin/env python3
"""golden_ss.npz: sub-sampling dataset items at full size, from the REAL reference on CPU.

Run in the build container only: ``python tests/golden/make_golden_ss.py``.
ESC_pc_temp_maxKSS (Code/dataset.py:169-199) at F=512, Nt=10 (N=5120, the shipped 3ST
framing) for K in SS_K; ESC_pc_ss (Code/dataset.py:58-80) on per-frame tables.  Inputs are
regenerated from seeds by the tests (inputs.ss_inputs); only outputs are stored (the value
column as float32, the coordinate columns as the selected point indices)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PCA_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "set_transformer-master"))
sys.path.insert(0, os.path.join(REF, "Code"))
os.chdir(os.path.join(REF, "Code"))

import inputs as gi  # noqa: E402
import dataset as ref_dataset  # noqa: E402  (reference)


def main():
    out = {}
    x3, y3, farr, tarr = gi.ss_inputs()
    F = farr.shape[0]
    for K in gi.SS_K:
        ds = ref_dataset.ESC_pc_temp_maxKSS(x3, y3, farr, tarr, K)
        for i in range(x3.shape[2]):
            pc, lbl = ds[i]
            pc = pc.numpy()
            assert pc.dtype == np.float64 and pc.shape == (K, 3)
            # recover the point index p = t*F + f of every row from its coordinates
            f = np.searchsorted(farr, pc[:, 0])
            t = np.searchsorted(tarr, pc[:, 1])
            assert np.array_equal(farr[f], pc[:, 0]) and np.array_equal(tarr[t], pc[:, 1])
            out[f"maxK{K}/sel{i}"] = (t * F + f).astype(np.int32)
            out[f"maxK{K}/val{i}"] = pc[:, 2].astype(np.float32)
            out[f"maxK{K}/label{i}"] = np.int64(lbl.item())
    # ESC_pc_ss on [K, T] tables (made here with numpy exactly as utils.pc_maxK does)
    x2 = x3[:, 0, :]                                        # [F, S] frames
    K = 51
    order = np.stack([(-x2[:, i]).argsort()[:K] for i in range(x2.shape[1])], axis=1)
    xs, fs = np.take_along_axis(x2, order, axis=0), farr[order]
    ds2 = ref_dataset.ESC_pc_ss(xs, y3, fs)
    out["ss2d/len"] = np.int64(len(ds2))
    for i in range(x2.shape[1]):
        pc, lbl = ds2[i]
        out[f"ss2d/item{i}"] = pc.numpy()
        out[f"ss2d/sel{i}"] = order[:, i].astype(np.int32)
    # importance sampling (Code/dataset.py:243-289), deterministic choice = 1; the heat map is
    # recomputed with the reference's own torch expressions to set the test tolerance
    import torch
    for winF in gi.IMP_WINF:
        for K in (51, 2551):
            ds = ref_dataset.ESC_pc_temp_importancerandKSS(x3, y3, farr, tarr, K, 1, winF)
            for i in range(x3.shape[2]):
                pc, lbl = ds[i]
                pc = pc.numpy()
                f = np.searchsorted(farr, pc[:, 0])
                t = np.searchsorted(tarr, pc[:, 1])
                out[f"imp{winF}/K{K}/sel{i}"] = (t * F + f).astype(np.int32)
                out[f"imp{winF}/K{K}/val{i}"] = pc[:, 2].astype(np.float32)
        for i in range(x3.shape[2]):
            g = torch.gradient(torch.tensor(x3[:, :, i]))
            g = g[0].abs() + g[1].abs()
            k = torch.kaiser_window(window_length=2, periodic=True, beta=5.09)[:, None] @ \
                torch.kaiser_window(window_length=winF, periodic=True, beta=5.09)[None, :]
            g = torch.nn.functional.conv2d(g[None, None, ...], k[None, None],
                                           padding='same')[0, 0] + 1.0e-6
            out[f"imp{winF}/heat{i}"] = g.numpy()
    np.savez_compressed(os.path.join(HERE, "golden_ss.npz"), **out)
    print("golden_ss.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
