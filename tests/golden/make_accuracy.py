#!/usr/bin/env python3
"""Accuracy-parity fixtures, made by running the REAL reference on CPU.

Run in the build container only:  ``python tests/golden/make_accuracy.py [train|agree|all]``

train  -> golden_acc_train.npz : the reference ``models.ST`` (cfg1/2 architecture) trained
          from a stored initialisation on the synthetic corpus of ``inputs.accuracy_corpus``
          with the loop of Code/settransformer.py:89-112 (Adam lr 1e-3, coupled weight decay
          1e-3, CrossEntropyLoss, batch 128), batches in the order of
          ``ShardedIndexStream(seed)``; stores initial and final weights, the per-step loss,
          the test logits and the test accuracy.
agree  -> golden_agree.npz : logits of the shipped FST / 3ST weights on >= 10 000 seeded
          synthetic sets each (``inputs.agreement_sets``), float32.
Only data is written; no reference source text is copied.
"""
import glob
import os
import sys
import time

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

torch.set_num_threads(8)


def pack2d(x, farr32, idx):
    """[B, F, 2] float32 batch of frames ``idx`` (what ESC_pc.__getitem__ + default collate
    give: Code/dataset.py:50-54)."""
    B = len(idx)
    out = np.empty((B, x.shape[0], 2), dtype=np.float32)
    out[:, :, 0] = farr32[None, :]
    out[:, :, 1] = x[:, idx].T
    return torch.from_numpy(out)


def gen_train():
    a = gi.ACC
    cp = gi.accuracy_corpus()
    farr32 = cp["farr"].astype(np.float32)
    torch.manual_seed(a["init_seed"])
    net = ref_models.ST(dim_input=2, num_outputs=1, dim_output=a["C"], num_inds=a["m"],
                        dim_hidden=a["d"], num_heads=a["h"])
    out = {f"init/{k}": v.detach().numpy().copy() for k, v in net.state_dict().items()}
    crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.Adam(net.parameters(), lr=a["lr"], weight_decay=a["wd"])
    T = cp["x_train"].shape[1]
    stream = ShardedIndexStream(T, a["B"], 0, 1, a["seed"], True, "cpu")
    steps = a["epochs"] * (T // a["B"])
    losses = np.zeros(steps, dtype=np.float64)
    per_epoch = T // a["B"]
    Tt = cp["x_test"].shape[1]

    def test_logits():
        net.eval()
        lg = np.zeros((Tt, a["C"]), dtype=np.float32)
        with torch.no_grad():
            for s0 in range(0, Tt, 220):
                idx = np.arange(s0, min(s0 + 220, Tt))
                lg[idx] = net(pack2d(cp["x_test"], farr32, idx)).numpy().reshape(len(idx), -1)
        net.train()
        return lg

    net.train()
    t0 = time.time()
    eval_steps, eval_acc = [], []
    for s in range(steps):
        idx = stream.next().numpy()
        imgs, labels = pack2d(cp["x_train"], farr32, idx), torch.from_numpy(cp["y_train"][idx])
        preds = net(imgs)
        loss = crit(preds, labels)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses[s] = loss.item()
        if s % 20 == 0:
            print(f"step {s}/{steps} loss {losses[s]:.4f}  {time.time() - t0:.0f}s", flush=True)
        # periodic test evaluation (the reference evaluates every 10 epochs,
        # Code/settransformer.py:117-131; every EVAL_EVERY epochs here) and after the last step
        ep_done = (s + 1) // per_epoch
        if (s + 1) % per_epoch == 0 and (ep_done % gi.ACC_EVAL_EVERY == 0 or s + 1 == steps):
            acc_e = float((test_logits().argmax(1) == cp["y_test"]).mean())
            eval_steps.append(s + 1)
            eval_acc.append(acc_e)
            print(f"  epoch {ep_done}: test acc {acc_e:.4f}", flush=True)
    logits = test_logits()
    acc = float((logits.argmax(1) == cp["y_test"]).mean())
    print("test accuracy", acc, "steps", steps)
    out["eval_steps"] = np.asarray(eval_steps, dtype=np.int64)
    out["eval_acc"] = np.asarray(eval_acc, dtype=np.float64)
    for k, v in net.state_dict().items():
        out[f"final/{k}"] = v.detach().numpy().copy()
    out["losses"] = losses
    out["test_logits"] = logits
    out["test_acc"] = np.float64(acc)
    out["steps"] = np.int64(steps)
    # fingerprint of the corpus, to detect drift of the regenerated arrays on another host
    out["corpus_sum"] = np.float64(cp["x_train"].astype(np.float64).sum())
    out["corpus_probe"] = cp["x_train"][::97, ::211].copy()
    np.savez_compressed(os.path.join(HERE, "golden_acc_train.npz"), **out)
    print("golden_acc_train.npz written")


def load_ckpt(pattern):
    (pth,) = glob.glob(os.path.join(REF, "Code", "model_saves", pattern))
    sd = torch.load(pth, map_location="cpu", weights_only=True)
    return {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}


def gen_agree():
    out = {}
    a = gi.ACC
    for tag, pattern, N, din, n, seed in (("cfg1", None, 512, 2, 10000, 9900),
                                          ("fst", "FST*_net.pth", 1025, 2, 10000, 9000),
                                          ("3st", "3ST*_net.pth", 5120, 3, 10000, 9500)):
        if pattern is None:      # the model trained by gen_train()
            g = np.load(os.path.join(HERE, "golden_acc_train.npz"))
            sd = {k[len("final/"):]: torch.from_numpy(g[k]) for k in g.files
                  if k.startswith("final/")}
            net = ref_models.ST(dim_input=2, num_outputs=1, dim_output=a["C"],
                                num_inds=a["m"], dim_hidden=a["d"], num_heads=a["h"])
        else:
            sd = load_ckpt(pattern)
            net = ref_models.ST(dim_input=din, num_outputs=1, dim_output=10, num_inds=64,
                                dim_hidden=64, num_heads=8)
        net.load_state_dict(sd)
        net.eval()
        logits = np.zeros((n, net.dec[1].out_features), dtype=np.float32)
        t0 = time.time()
        with torch.no_grad():
            for s, X in gi.agreement_sets(seed, n, N, din, chunk=100):
                logits[s:s + X.shape[0]] = net(torch.from_numpy(X)).numpy()
                if s % 1000 == 0:
                    print(tag, s, f"{time.time() - t0:.0f}s", flush=True)
        out[f"{tag}/logits"] = logits
    np.savez_compressed(os.path.join(HERE, "golden_agree.npz"), **out)
    print("golden_agree.npz written")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("train", "all"):
        gen_train()
    if what in ("agree", "all"):
        gen_agree()
