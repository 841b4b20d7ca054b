#!/usr/bin/env python3
"""golden_acc_spread.npz: run-to-run spread of the REFERENCE's own training on the accuracy corpus.

Run in the build container only:  ``python tests/golden/make_accuracy_spread.py RUN0 RUN1``
(runs RUN0 .. RUN1-1; several invocations in parallel write part files, ``merge`` joins them).

Adam at lr 1e-3 is chaotic on this corpus (see make_accuracy.py): one reference run is one draw
from a distribution.  This script repeats make_accuracy.py's training loop with the REAL reference
model, the same data, batch order and hyper-parameters, from the stored initial weights
multiplied by (1 + 1e-6 * n), n ~ N(0, 1) seeded per run - a perturbation below fp32 reduction-order
noise - and records the ten test evaluations of every run.  tests/test_gpu_accuracy.py compares
the HIP runs with THIS distribution instead of with the single run of golden_acc_train.npz."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PCA_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "point-cloud-audio_amd"))
sys.path.insert(0, os.path.join(REF, "set_transformer-master"))
sys.path.insert(0, os.path.join(REF, "Code"))
os.chdir(os.path.join(REF, "Code"))

import inputs as gi  # noqa: E402
import models as ref_models  # noqa: E402    (reference)
from pca_hip.trainer import ShardedIndexStream  # noqa: E402  (pure index logic)


def pack2d(x, farr32, idx):
    out = np.empty((len(idx), x.shape[0], 2), dtype=np.float32)
    out[:, :, 0] = farr32[None, :]
    out[:, :, 1] = x[:, idx].T
    return torch.from_numpy(out)


def one_run(run, cp, g):
    a = gi.ACC
    farr32 = cp["farr"].astype(np.float32)
    net = ref_models.ST(dim_input=2, num_outputs=1, dim_output=a["C"], num_inds=a["m"],
                        dim_hidden=a["d"], num_heads=a["h"])
    gen = torch.Generator().manual_seed(5000 + run)
    sd = {}
    for k in g.files:
        if k.startswith("init/"):
            w = torch.from_numpy(g[k])
            sd[k[5:]] = w * (1.0 + 1e-6 * torch.randn(w.shape, generator=gen))
    net.load_state_dict(sd)
    crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.Adam(net.parameters(), lr=a["lr"], weight_decay=a["wd"])
    T, Tt = cp["x_train"].shape[1], cp["x_test"].shape[1]
    stream = ShardedIndexStream(T, a["B"], 0, 1, a["seed"], True, "cpu")
    per_epoch = T // a["B"]
    steps = a["epochs"] * per_epoch
    accs = []
    net.train()
    for s in range(steps):
        idx = stream.next().numpy()
        loss = crit(net(pack2d(cp["x_train"], farr32, idx)), torch.from_numpy(cp["y_train"][idx]))
        opt.zero_grad()
        loss.backward()
        opt.step()
        ep_done = (s + 1) // per_epoch
        if (s + 1) % per_epoch == 0 and (ep_done % gi.ACC_EVAL_EVERY == 0 or s + 1 == steps):
            net.eval()
            ok = 0
            with torch.no_grad():
                for s0 in range(0, Tt, 220):
                    ii = np.arange(s0, min(s0 + 220, Tt))
                    lg = net(pack2d(cp["x_test"], farr32, ii)).numpy().reshape(len(ii), -1)
                    ok += int((lg.argmax(1) == cp["y_test"][ii]).sum())
            net.train()
            accs.append(ok / Tt)
            print(f"run {run} epoch {ep_done}: acc {accs[-1]:.4f}", flush=True)
    return np.asarray(accs)


def main():
    if sys.argv[1] == "merge":
        parts = sorted(f for f in os.listdir(HERE) if f.startswith("_spread_part_"))
        runs, accs = [], []
        for f in parts:
            z = np.load(os.path.join(HERE, f))
            runs += list(z["runs"])
            accs += list(z["eval_acc"])
        order = np.argsort(runs)
        np.savez_compressed(os.path.join(HERE, "golden_acc_spread.npz"),
                            runs=np.asarray(runs)[order], eval_acc=np.asarray(accs)[order],
                            perturbation=np.float64(1e-6))
        print("golden_acc_spread.npz:", len(runs), "runs")
        for f in parts:
            os.remove(os.path.join(HERE, f))
        return
    r0, r1 = int(sys.argv[1]), int(sys.argv[2])
    torch.set_num_threads(int(os.environ.get("PCA_THREADS", "2")))
    cp = gi.accuracy_corpus()
    g = np.load(os.path.join(HERE, "golden_acc_train.npz"))
    accs = [one_run(r, cp, g) for r in range(r0, r1)]
    np.savez(os.path.join(HERE, f"_spread_part_{r0:03d}.npz"), runs=np.arange(r0, r1),
             eval_acc=np.stack(accs))


if __name__ == "__main__":
    main()
