"""Accuracy parity of the HIP path against fixtures made by the REAL
reference on CPU (tests/golden/make_accuracy.py):

(i)  weights -> predictions: identical weights, >= 10 000 seeded synthetic sets per model;
     argmax agreement >= 99.8 % and, in fp32 mode, max|dlogit| <= 1e-3 * max|logit|;
(ii) train from scratch: same initial weights, same corpus, same batch order, same number of
     Adam steps; early loss curve and final test accuracy against the reference's.
"""
import os

import numpy as np
import pytest
import torch

import inputs as gi
from util import T

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _npz(name):
    path = os.path.join(HERE, "golden", name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    return np.load(path)


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def _agreement(net, ref_logits, seed, n, N, din, mode, dev, chunk=100):
    from pca_hip import trainer
    eng = trainer.STEngine(net, chunk, N, mode, training=False)
    agree, worst, scale = 0, 0.0, float(np.abs(ref_logits).max())
    for s, X in gi.agreement_sets(seed, n, N, din, chunk):
        lg = eng.forward(T(X, dev)).cpu().numpy()
        ref = ref_logits[s:s + chunk]
        agree += int((lg.argmax(1) == ref.argmax(1)).sum())
        worst = max(worst, float(np.abs(lg - ref).max()))
    return agree / n, worst / scale


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("tag,N,din,seed", [("fst", 1025, 2, 9000), ("3st", 5120, 3, 9500)])
def test_shipped_weights_agreement(tag, N, din, seed, mode, dev, golden_ckpt):
    """Shipped FST / 3ST weights on 10 000 synthetic sets each: the exact fp32 chain, and (mode
    bf16) the fused forward kernels of the shipped architecture d=64, dh=8, m=64
    (csrc/sd64_fwd.hip: fp32 arithmetic on the vector ALU, so the same 1e-3 bound holds)."""
    import models
    from pca_hip import _lib
    ref = _npz("golden_agree.npz")[f"{tag}/logits"]
    net = models.ST(dim_input=din, num_outputs=1, dim_output=10, num_inds=64, dim_hidden=64,
                    num_heads=8).to(dev)
    sd = {k[len("module."):] if k.startswith("module.") else k: T(v)
          for k, v in golden_ckpt.sub(("tst" if tag == "3st" else tag) + "/p/").items()}
    # (saved from a DP-wrapped model, hence the prefix)
    net.load_state_dict(sd)
    md = _lib.MODE_F32 if mode == "f32" else _lib.MODE_BF16
    if mode == "bf16":      # the fused kernels must be what serves the shape
        import ctypes as C
        for shp in (_lib.MabShape(100, N, 64, din, 64, 64, 8, 0, md, 0, 0, 0, None, 0),
                    _lib.MabShape(100, 64, N, 64, din, 64, 8, 1, md, 0, 0, 0, None, 0),
                    _lib.MabShape(100, 1, N, 64, 64, 64, 8, 1, md, 0, 0, 0, None, 0)):
            assert _lib.lib().pca_mab_fwd_ws_bytes(C.byref(shp)) > 0, _lib.lib().pca_last_error()
            assert _lib.lib().pca_mab_saved_bytes(C.byref(shp)) == 0     # (forward only)
    frac, rel = _agreement(net, ref, seed, ref.shape[0], N, din, md, dev)
    print(f"{tag} {mode}: agreement {frac:.5f}, max|dlogit|/max|logit| {rel:.2e}")
    assert ref.shape[0] >= 10000
    assert frac >= 0.998
    assert rel <= 1e-3


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_trained_cfg1_weights_agreement(mode, dev):
    """The reference-trained cfg1/2 model (d=128, h=4, m=16) on 10 000 synthetic sets, in the
    exact fp32 mode and through the fused bf16 MFMA kernels."""
    import models
    from pca_hip import _lib
    g = _npz("golden_acc_train.npz")
    ref = _npz("golden_agree.npz")["cfg1/logits"]
    a = gi.ACC
    net = models.ST(dim_input=2, num_outputs=1, dim_output=a["C"], num_inds=a["m"],
                    dim_hidden=a["d"], num_heads=a["h"]).to(dev)
    net.load_state_dict({k[len("final/"):]: T(g[k]) for k in g.files if k.startswith("final/")})
    m = _lib.MODE_F32 if mode == "f32" else _lib.MODE_BF16
    frac, rel = _agreement(net, ref, 9900, ref.shape[0], 512, 2, m, dev)
    print(f"cfg1 {mode}: agreement {frac:.5f}, max|dlogit|/max|logit| {rel:.2e}")
    assert frac >= 0.998
    if mode == "f32":
        assert rel <= 1e-3


def test_trained_cfg1_weights_agreement_fp8(dev):
    """PCA_MODE_FP8 (fp8 e4m3 operands in fc_o of the many-queries blocks; fc_q stays bf16: with
    it in fp8 as well the agreement measures 99.39 %) on the reference-trained cfg1 model: argmax
    agreement with the reference's fp32 logits on the same 10 000 sets, held to the 99.8 % bar of
    SURVEY.md 8d (measured 99.83 %; bf16 mode 99.97 %)."""
    import models
    from pca_hip import _lib
    g = _npz("golden_acc_train.npz")
    ref = _npz("golden_agree.npz")["cfg1/logits"]
    a = gi.ACC
    net = models.ST(dim_input=2, num_outputs=1, dim_output=a["C"], num_inds=a["m"],
                    dim_hidden=a["d"], num_heads=a["h"]).to(dev)
    net.load_state_dict({k[len("final/"):]: T(g[k]) for k in g.files if k.startswith("final/")})
    frac, rel = _agreement(net, ref, 9900, ref.shape[0], 512, 2, _lib.MODE_FP8, dev)
    fb, _ = _agreement(net, ref, 9900, ref.shape[0], 512, 2, _lib.MODE_BF16, dev)
    print(f"cfg1 fp8: agreement {frac:.5f} (bf16 {fb:.5f}), max|dlogit|/max|logit| {rel:.2e}")
    assert frac >= 0.998


RUNS = int(os.environ.get("PCA_ACC_RUNS", "20"))


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_train_from_scratch_accuracy(mode, dev):
    """Same initial weights, corpus, batch order and 598 Adam steps as the reference run.

    Adam at lr 1e-3 on this data is chaotic: the REFERENCE's own loss spikes above 10 late in
    training and its test accuracy drops from 0.89 to 0.49 and back between evaluations, and
    two runs of the HIP path differ by several tenths of a percent in any late snapshot.  So one
    reference run is one draw from a distribution, and that distribution is what the HIP runs
    are compared with: golden_acc_spread.npz holds 9 runs of the reference itself from the same
    initial weights perturbed by 1e-6 (make_accuracy_spread.py) - converged accuracy 0.9929,
    sigma 0.0013; the single run of golden_acc_train.npz (0.9948) is a +1.5 sigma draw of it.
    What is compared:
      * the first 20 losses, step by step (tight: the trajectories have not separated yet);
      * the converged accuracy = mean of the three best of the ten evaluations (2200 test sets
        each), averaged over RUNS >= 20 repetitions, against the mean of the reference's runs:
        SURVEY.md 8d's +-0.2 %, plus two standard errors of the difference of the two means
        (about 0.15 % at 20 runs).  Every run but the first starts from the initial weights
        perturbed by 1e-6 (relative), exactly as the reference's spread runs do, so that the
        bit-reproducible bf16 step samples a distribution too (the measured means and sigmas
        are printed by the test and recorded in DESIGN.md section 2)."""
    import dataset
    import models
    from pca_hip import _lib, trainer
    g = _npz("golden_acc_train.npz")
    a = gi.ACC
    cp = gi.accuracy_corpus()
    # the regenerated corpus must be the one the reference trained on
    np.testing.assert_allclose(cp["x_train"][::97, ::211], g["corpus_probe"], rtol=0, atol=2e-6)
    ds = dataset.ESC_pc(cp["x_train"], cp["y_train"], cp["farr"], device=dev)
    dt = dataset.ESC_pc(cp["x_test"], cp["y_test"], cp["farr"], device=dev)
    m = _lib.MODE_F32 if mode == "f32" else _lib.MODE_BF16
    steps = int(g["steps"])
    eval_at = {int(v) for v in g["eval_steps"]}
    ref_losses, ref_accs = g["losses"], g["eval_acc"]
    top3 = lambda v: float(np.sort(v)[-3:].mean())                      # noqa: E731
    conv, medians = [], []
    for run in range(RUNS):
        net = models.ST(dim_input=2, num_outputs=1, dim_output=a["C"], num_inds=a["m"],
                        dim_hidden=a["d"], num_heads=a["h"]).to(dev)
        # the reference's own spread (make_accuracy_spread.py) comes from initial weights
        # multiplied by (1 + 1e-6 n), n ~ N(0, 1) seeded per run; the bf16 step is bit-reproducible
        # (no atomics since round 2), so without the same perturbation its RUNS runs would be ONE
        # draw repeated.  Run 0 keeps the unperturbed weights (the early-loss comparison below).
        gen = torch.Generator().manual_seed(7000 + run)
        sd = {}
        for k in g.files:
            if k.startswith("init/"):
                w = torch.from_numpy(g[k])
                sd[k[len("init/"):]] = w if run == 0 else w * (1.0 + 1e-6 * torch.randn(w.shape, generator=gen))
        net.load_state_dict(sd)
        tr = trainer.Trainer(net, ds, a["B"], lr=a["lr"], weight_decay=a["wd"], mode=m,
                             seed=a["seed"], shuffle=True)
        losses, accs = np.zeros(steps), []
        for s in range(steps):
            tr.step()
            losses[s] = float(tr.eng.loss)
            if s + 1 in eval_at:
                acc, n = trainer.evaluate(net, dt, 220, m)
                assert n == cp["x_test"].shape[1]
                accs.append(acc)
        accs = np.asarray(accs)
        # the difference to the reference grows ~3x per step (fp32: 2e-7 at step 1, 1e-4 at
        # step 22, 2e-3 at step 24): compare before the trajectories separate
        early = 20
        tol = 1e-3 if mode == "f32" else 3e-2
        if run == 0:     # (the perturbed runs separate from the reference's trajectory sooner)
            assert np.abs(losses[:early] - ref_losses[:early]).max() < tol, \
                (losses[:early], ref_losses[:early])
        print(f"{mode} run {run}: evaluations {np.round(accs, 4)} -> converged {top3(accs):.4f}; "
              f"largest loss spike {losses[50:].max():.1f}")
        conv.append(top3(accs))
        medians.append(float(np.median(accs)))
    conv = np.asarray(conv)
    sem = conv.std(ddof=1) / np.sqrt(RUNS)
    print(f"{mode}: reference evaluations {np.round(ref_accs, 4)} -> converged "
          f"{top3(ref_accs):.4f} (largest loss spike {ref_losses[50:].max():.1f}); HIP mean "
          f"{conv.mean():.4f}, run-to-run sigma {conv.std(ddof=1):.4f}")
    spread = _npz("golden_acc_spread.npz")["eval_acc"]
    ref_conv = np.asarray([top3(r) for r in spread])
    ref_sem = ref_conv.std(ddof=1) / np.sqrt(len(ref_conv))
    print(f"{mode}: reference distribution ({len(ref_conv)} runs) mean {ref_conv.mean():.4f}, sigma "
          f"{ref_conv.std(ddof=1):.4f}; difference of means {conv.mean() - ref_conv.mean():+.4f}, "
          f"standard error {np.hypot(sem, ref_sem):.4f}")
    tol_acc = float(os.environ.get("PCA_ACC_TOL", "0.002"))
    assert abs(conv.mean() - ref_conv.mean()) <= tol_acc + 2 * np.hypot(sem, ref_sem)
    assert np.median(medians) >= np.median(ref_accs) - 0.20     # coarse guard only: see above


RUNS_SHIPPED = int(os.environ.get("PCA_ACC_RUNS_SHIPPED", "10"))


def test_train_from_scratch_shipped_hyperparameters_fused_vs_exact(dev):
    """The reference's own hyper-parameters (d = 64, 8 heads of dim 8, 64 inducing points:
    Code/settransformer.py:81-83) trained from scratch on the accuracy corpus in the FAST mode - the
    attention core without the matrix A (csrc/attn_core.hip), k_lin64, k_wgrad64 - against the exact fp32
    mode of this library on the same initial weights, corpus, batch order and 598 Adam steps.  The exact
    mode is what (i) / (ii) above pin to the reference; no reference run exists at this width, so the
    bar is the exact mode's own distribution: first 20 losses step by step (3e-2), converged accuracy
    (mean of the three best of ten evaluations) averaged over RUNS_SHIPPED runs per mode from initial
    weights perturbed by 1e-6, within 0.5 % plus two standard errors of the difference.  (Wider than 8d's
    0.2 %: at this width Adam at lr 1e-3 is more chaotic still than in (ii) - single evaluations of either
    mode fall to 0.11-0.3 late in training - and the fused mode's weight gradients are summed with atomics,
    so its runs differ run to run; measured on MI355X, 8 runs each: exact 0.9854 (sigma 0.0026), fused
    0.9826 (sigma 0.0055), difference -0.0028 +- 0.0021.)"""
    import dataset
    import models
    from pca_hip import _lib, trainer
    g = _npz("golden_acc_train.npz")
    a = gi.ACC
    cp = gi.accuracy_corpus()
    ds = dataset.ESC_pc(cp["x_train"], cp["y_train"], cp["farr"], device=dev)
    dt = dataset.ESC_pc(cp["x_test"], cp["y_test"], cp["farr"], device=dev)
    steps = int(g["steps"])
    eval_at = {int(v) for v in g["eval_steps"]}
    top3 = lambda v: float(np.sort(v)[-3:].mean())                      # noqa: E731
    torch.manual_seed(a["init_seed"])
    init = {k: v.detach().cpu().clone() for k, v in
            models.ST(dim_input=2, num_outputs=1, dim_output=a["C"], num_inds=64, dim_hidden=64,
                      num_heads=8).state_dict().items()}
    conv, early = {}, {}
    for mode, m in (("f32", _lib.MODE_F32), ("bf16", _lib.MODE_BF16)):
        conv[mode] = []
        for run in range(RUNS_SHIPPED):
            net = models.ST(dim_input=2, num_outputs=1, dim_output=a["C"], num_inds=64, dim_hidden=64,
                            num_heads=8).to(dev)
            gen = torch.Generator().manual_seed(7100 + run)
            net.load_state_dict({k: (w if run == 0 else w * (1.0 + 1e-6 * torch.randn(w.shape, generator=gen)))
                                 for k, w in init.items()})
            tr = trainer.Trainer(net, ds, a["B"], lr=a["lr"], weight_decay=a["wd"], mode=m,
                                 seed=a["seed"], shuffle=True)
            losses, accs = np.zeros(steps), []
            for s in range(steps):
                tr.step()
                if run == 0 and s < 20:
                    losses[s] = float(tr.eng.loss)
                if s + 1 in eval_at:
                    acc, n = trainer.evaluate(net, dt, 220, m)
                    accs.append(acc)
            if run == 0:
                early[mode] = losses[:20].copy()
            conv[mode].append(top3(np.asarray(accs)))
            print(f"shipped hyper-parameters, {mode} run {run}: evaluations {np.round(accs, 4)}")
        conv[mode] = np.asarray(conv[mode])
    assert np.abs(early["bf16"] - early["f32"]).max() < 3e-2, (early["bf16"], early["f32"])
    sem = np.hypot(conv["f32"].std(ddof=1), conv["bf16"].std(ddof=1)) / np.sqrt(RUNS_SHIPPED)
    diff = conv["bf16"].mean() - conv["f32"].mean()
    print(f"shipped hyper-parameters: exact mode mean {conv['f32'].mean():.4f} (sigma {conv['f32'].std(ddof=1):.4f}), "
          f"fused mode mean {conv['bf16'].mean():.4f} (sigma {conv['bf16'].std(ddof=1):.4f}); difference "
          f"{diff:+.4f}, standard error {sem:.4f}")
    assert conv["f32"].mean() > 0.9, conv["f32"]          # the model did learn the task
    assert abs(diff) <= 0.005 + 2 * sem, (diff, sem)
