"""Pin the CPU oracle (oracle/st_oracle.py) against the golden vectors that the
real reference produced (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import inputs as gi
from oracle import st_oracle as orc

FP32_TOL = 2e-5     # fp32 restatement vs fp32 reference, same op order up to head layout


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, tol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.max(np.abs(a - b)) if a.size else 0.0
    ref = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    assert err <= tol * ref, f"{what}: max|d|={err:.3e} > {tol:.1e}*{ref:.3g}"


@pytest.mark.parametrize("case", gi.MAB_CASES, ids=[c[0] for c in gi.MAB_CASES])
def test_mab_forward_backward(golden_mab, case):
    ci = gi.MAB_CASES.index(case)
    name, B, nq, nk, dq, dk, d, h = case
    p = {k: T(v) for k, v in golden_mab.sub(f"{name}/p/").items()}
    Q = T(gi.randn(200 + ci, B, nq, dq))
    K = T(gi.randn(300 + ci, B, nk, dk))
    G = T(gi.randn(400 + ci, B, nq, d))
    Y = orc.mab_forward(Q, K, p, h)
    close(Y, golden_mab[f"{name}/Y"], FP32_TOL, "Y")
    g = orc.mab_backward(G, Q, K, p, h)
    close(g["dQ"], golden_mab[f"{name}/dQ"], 5e-5, "dQ")
    close(g["dK"], golden_mab[f"{name}/dK"], 5e-5, "dK")
    for k, v in golden_mab.sub(f"{name}/g/").items():
        close(g[k], v, 5e-5, k)
    # fp64 restatement vs fp64 reference: essentially exact
    p64 = {k: v.double() for k, v in p.items()}
    close(orc.mab_forward(Q.double(), K.double(), p64, h), golden_mab[f"{name}/Y64"],
          1e-12, "Y64")


@pytest.mark.parametrize("case", gi.ST_CASES, ids=[c[0] for c in gi.ST_CASES])
def test_st_forward_backward(golden_st, case):
    ci = gi.ST_CASES.index(case)
    name, B, N, din, d, h, m, C, full = case
    p = {k: T(v) for k, v in golden_st.sub(f"{name}/p/").items()}
    assert [k for k, _ in orc.st_param_shapes(din, 1, C, m, d)] == list(p.keys())
    for k, shp in orc.st_param_shapes(din, 1, C, m, d):
        assert tuple(p[k].shape) == shp
    X = T(gi.pc_input(600 + ci, B, N, din))
    y = T(gi.labels(700 + ci, B, C))
    logits = orc.st_forward(X, p, h)
    close(logits, golden_st[f"{name}/logits"], FP32_TOL, "logits")
    assert logits.shape == ((C,) if B == 1 else (B, C))      # .squeeze() semantics
    loss, _, grads = orc.st_grads(X, y, p, h)
    assert abs(loss - float(golden_st[f"{name}/loss"])) < 1e-5
    if full:
        for k, v in golden_st.sub(f"{name}/g/").items():
            close(grads[k], v, 1e-4, k)
    else:
        for k, v in golden_st.sub(f"{name}/gsub/").items():
            close(grads[k].reshape(-1)[::gi.GRAD_SUBSAMPLE], v, 1e-4, k)
            nrm = float(golden_st[f"{name}/gnorm/{k}"])
            assert abs(float(grads[k].double().norm()) - nrm) <= 1e-4 * max(nrm, 1e-3)
    p64 = {k: v.double() for k, v in p.items()}
    close(orc.st_forward(X.double(), p64, h), golden_st[f"{name}/logits64"], 1e-11, "l64")


@pytest.mark.parametrize("tag,din,Ns", [("fst", 2, gi.CKPT_2D_N), ("tst", 3, gi.CKPT_3D_N)])
def test_shipped_checkpoints(golden_ckpt, tag, din, Ns):
    p = {k: T(v) for k, v in golden_ckpt.sub(f"{tag}/p/").items()}
    assert len(p) == 45 and all(k.startswith("module.") for k in p)
    nparam = sum(v.numel() for v in p.values())
    assert nparam == (80202 if tag == "fst" else 80394)     # *_config.json model_params
    for i, N in enumerate(Ns):
        X = T(gi.pc_input(900 + 10 * din + i, 8, N, din))
        lg = orc.st_forward(X, p, 8)
        ref = golden_ckpt[f"{tag}/logits/{N}"]
        close(lg, ref, 1e-4, f"{tag} N={N}")
        assert (lg.argmax(1).numpy() == ref.argmax(1)).all()


def test_pack_points(golden_dataset):
    g = golden_dataset
    x, farr = g["pc2d/x"], g["pc2d/farr"]
    for i in range(x.shape[1]):
        pc = orc.pack_points_2d(x, farr, i)
        assert pc.dtype == np.float32
        np.testing.assert_array_equal(pc, g[f"pc2d/item{i}"])
    x3, farr3, tarr3 = g["pc3d/x"], g["pc3d/farr"], g["pc3d/tarr"]
    for i in range(x3.shape[2]):
        np.testing.assert_array_equal(orc.pack_points_3d(x3, farr3, tarr3, i),
                                      g[f"pc3d/item{i}"])
        for K in (1, 7, x3.shape[0] * x3.shape[1]):
            np.testing.assert_array_equal(orc.pc_maxk_3d(x3, farr3, tarr3, i, K),
                                          g[f"pc3d/maxK{K}/item{i}"])


def test_train_trajectory(golden_train):
    B, N, din, d, h, m, C, steps = [int(v) for v in golden_train["cfg"]]
    p = {k: T(v).clone() for k, v in golden_train.sub("p0/").items()}
    opt = orc.AdamState(p, lr=1e-3, wd=1e-3)
    losses = []
    for s in range(steps):
        X = T(gi.pc_input(5000 + s, B, N, din))
        y = T(gi.labels(6000 + s, B, C))
        loss, _ = orc.train_step(X, y, p, opt, h)
        losses.append(loss)
    np.testing.assert_allclose(losses, golden_train["losses"], rtol=0, atol=2e-4)
    for k, v in golden_train.sub("p20/").items():
        close(p[k], v, 2e-3, k)


def test_stft_against_torch_stft():
    """librosa is absent (parity unpinned against it); the documented librosa-0.8
    semantics are checked against torch.stft here."""
    rng = np.random.Generator(np.random.PCG64(7))
    wave = (rng.standard_normal(6000) * 0.3).astype(np.float32)
    for n_fft, win in ((1024, 1024), (2048, 2048), (256, 200)):
        hop = n_fft // 2
        a = orc.stft_logmag(wave, n_fft, win, hop)
        w = torch.hann_window(win, periodic=True, dtype=torch.float64)
        s = torch.stft(torch.from_numpy(wave).double(), n_fft, hop_length=hop,
                       win_length=win, window=w, center=True, pad_mode="reflect",
                       return_complex=True)
        b = torch.log(1e-8 + s.abs() / n_fft).float().numpy()
        assert a.shape == b.shape == (1 + n_fft // 2, 1 + len(wave) // hop)
        np.testing.assert_allclose(a, b, atol=2e-4, rtol=0)
    c = orc.chunk_frames(a, 3)
    assert c.shape == (a.shape[0], 3, a.shape[1] // 3)
    np.testing.assert_array_equal(c[:, :, 1], a[:, 3:6])


def test_reference_accuracy_spread_fixture():
    """golden_acc_spread.npz (tests/golden/make_accuracy_spread.py): the reference's own
    run-to-run distribution on the accuracy corpus - what tests/test_gpu_accuracy.py compares the
    HIP runs with.  The single reference run of golden_acc_train.npz must be a plausible draw."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    sp = np.load(os.path.join(here, "golden", "golden_acc_spread.npz"))
    one = np.load(os.path.join(here, "golden", "golden_acc_train.npz"))
    ev = sp["eval_acc"]
    assert ev.shape[0] >= 9 and ev.shape[1] == one["eval_acc"].shape[0]
    assert np.all((ev >= 0) & (ev <= 1))
    top3 = lambda v: float(np.sort(v)[-3:].mean())                      # noqa: E731
    conv = np.asarray([top3(r) for r in ev])
    assert 0.99 < conv.mean() < 0.996 and conv.std(ddof=1) < 0.003
    assert abs(top3(one["eval_acc"]) - conv.mean()) < 3 * conv.std(ddof=1)
