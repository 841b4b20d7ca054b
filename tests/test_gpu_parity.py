"""GPU parity: the HIP path (through the C ABI) against the golden vectors produced by the
reference and against the CPU oracle on the same seeded inputs.  Run with ``-m gpu``.

Tolerances (fp32 exact mode, PCA_MODE_F32): forward 1e-4, gradients 2e-4, both relative to
max(1, max|ref|).  The reference's own fp32-vs-fp64 drift is 5.8e-5 at |logit| ~ 15
(SURVEY.md section 4), so these sit just above the noise floor of fp32 summation order.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import inputs as gi
from util import T, close

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4
BWD_TOL = 2e-4


@pytest.fixture(scope="module")
def dev():
    import pca_hip
    pca_hip.lib()                     # fail loudly if the extension is missing
    pca_hip.set_mode("f32")
    return torch.device("cuda", 0)


# ----------------------------------------------------------------------------- #
# building blocks                                                                #
# ----------------------------------------------------------------------------- #
def test_gemm_strided_batched_splitk(dev):
    from pca_hip import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    for (M, N, K, split) in [(70, 33, 19, 1), (5, 130, 300, 1), (128, 128, 5000, 0),
                             (64, 2, 4096, 8), (1, 1, 1, 1)]:
        A = torch.randn(M, K, generator=g).to(dev)
        Bm = torch.randn(N, K, generator=g).to(dev)          # used transposed
        bias = torch.randn(N, generator=g).to(dev)
        Cout = torch.zeros(M, N, device=dev)
        d = _lib.GemmDesc(M, N, K, K, 1, 1, K, N, 1, 1, 0, 0, 0, 0, 0, 0, 0, split, 1.0)
        _lib.check(L.pca_gemm_f32(C.byref(d), A.data_ptr(), Bm.data_ptr(), bias.data_ptr(),
                                  Cout.data_ptr(), None))
        torch.cuda.synchronize()
        ref = A.double().cpu() @ Bm.double().cpu().t() + bias.double().cpu()
        close(Cout, ref, 2e-5 if K < 1000 else 1e-4, f"gemm {M}x{N}x{K}")
    # batched + transposed-A + accumulate: C[z] += A[z]^T B[z]
    nb1, nb2, M, N, K = 3, 2, 17, 9, 40
    A = torch.randn(nb1, nb2, K, M, generator=g).to(dev)
    Bm = torch.randn(nb1, nb2, K, N, generator=g).to(dev)
    C0 = torch.randn(nb1, nb2, M, N, generator=g).to(dev)
    Cout = C0.clone()
    d = _lib.GemmDesc(M, N, K, 1, M, N, 1, N, nb1, nb2, nb2 * K * M, K * M, nb2 * K * N,
                      K * N, nb2 * M * N, M * N, 1, 1, 0.5)
    _lib.check(L.pca_gemm_f32(C.byref(d), A.data_ptr(), Bm.data_ptr(), None,
                              Cout.data_ptr(), None))
    ref = C0.double().cpu() + 0.5 * (A.double().cpu().transpose(-1, -2) @ Bm.double().cpu())
    close(Cout, ref, 2e-5, "batched gemm")


def test_softmax_and_colsum(dev):
    from pca_hip import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    for n in (1, 7, 16, 33, 64, 513, 4, 8, 32, 100, 512, 1028, 4096, 4100):
        X = (torch.randn(37, n, generator=g) * 3).to(dev)
        A = X.clone()
        _lib.check(L.pca_softmax_rows(A.data_ptr(), 37, n, 0.25, None))
        ref = torch.softmax(X.double().cpu() * 0.25, -1)
        close(A, ref, 1e-6, f"softmax n={n}")
        dA = torch.randn(37, n, generator=g).to(dev)
        dS = dA.clone()
        _lib.check(L.pca_softmax_bwd_rows(A.data_ptr(), dS.data_ptr(), 37, n, 0.25, None))
        a = A.double().cpu()
        da = dA.double().cpu()
        close(dS, a * (da - (da * a).sum(-1, keepdim=True)) * 0.25, 1e-6, f"softmax bwd n={n}")
    X = torch.randn(1000, 130, generator=g).to(dev)
    out = torch.ones(130, device=dev)
    _lib.check(L.pca_colsum(X.data_ptr(), 1000, 130, out.data_ptr(), 1, None))
    close(out, 1.0 + X.double().cpu().sum(0), 1e-5, "colsum")
    for rows, cols in ((5000, 256), (4133, 132), (70000, 64)):      # float4 path
        X = torch.randn(rows, cols, generator=g).to(dev)
        for acc in (0, 1):
            out = torch.full((cols,), 2.0, device=dev)
            _lib.check(L.pca_colsum(X.data_ptr(), rows, cols, out.data_ptr(), acc, None))
            close(out, 2.0 * acc + X.double().cpu().sum(0), 2e-5, f"colsum {rows}x{cols}")


# ----------------------------------------------------------------------------- #
# MAB / ST against the reference's golden vectors                                #
# ----------------------------------------------------------------------------- #
@pytest.mark.parametrize("case", gi.MAB_CASES, ids=[c[0] for c in gi.MAB_CASES])
def test_mab_golden(dev, golden_mab, case):
    import modules
    ci = gi.MAB_CASES.index(case)
    name, B, nq, nk, dq, dk, d, h = case
    mab = modules.MAB(dq, dk, d, h).to(dev)
    mab.load_state_dict({k: T(v) for k, v in golden_mab.sub(f"{name}/p/").items()})
    Q = T(gi.randn(200 + ci, B, nq, dq), dev).requires_grad_(True)
    K = T(gi.randn(300 + ci, B, nk, dk), dev).requires_grad_(True)
    G = T(gi.randn(400 + ci, B, nq, d), dev)
    Y = mab(Q, K)
    close(Y, golden_mab[f"{name}/Y"], FWD_TOL, "Y")
    (Y * G).sum().backward()
    close(Q.grad, golden_mab[f"{name}/dQ"], BWD_TOL, "dQ")
    close(K.grad, golden_mab[f"{name}/dK"], BWD_TOL, "dK")
    for k, p in mab.named_parameters():
        close(p.grad, golden_mab[f"{name}/g/{k}"], BWD_TOL, k)
    with torch.no_grad():                                   # inference entry (saved=NULL)
        close(mab(Q, K), golden_mab[f"{name}/Y"], FWD_TOL, "Y(no_grad)")


def _build_st(golden, name, din, d, h, m, C, dev):
    import models
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    net.load_state_dict({k: T(v) for k, v in golden.sub(f"{name}/p/").items()})
    return net


@pytest.mark.parametrize("case", gi.ST_CASES, ids=[c[0] for c in gi.ST_CASES])
def test_st_golden(dev, golden_st, case):
    import pca_hip
    ci = gi.ST_CASES.index(case)
    name, B, N, din, d, h, m, C, full = case
    net = _build_st(golden_st, name, din, d, h, m, C, dev)
    X = T(gi.pc_input(600 + ci, B, N, din), dev).requires_grad_(True)
    y = T(gi.labels(700 + ci, B, C), dev)
    logits = net(X)
    assert tuple(logits.shape) == ((C,) if B == 1 else (B, C))          # .squeeze()
    close(logits, golden_st[f"{name}/logits"], FWD_TOL, "logits")
    lg2 = logits if logits.dim() == 2 else logits.unsqueeze(0)
    loss = pca_hip.cross_entropy(lg2, y)
    assert abs(float(loss) - float(golden_st[f"{name}/loss"])) < 2e-5
    loss.backward()
    close(X.grad, golden_st[f"{name}/dX"], BWD_TOL, "dX")
    for k, p in net.named_parameters():
        if full:
            close(p.grad, golden_st[f"{name}/g/{k}"], BWD_TOL, k)
        else:
            close(p.grad.reshape(-1)[::gi.GRAD_SUBSAMPLE], golden_st[f"{name}/gsub/{k}"],
                  BWD_TOL, k)
            nrm = float(golden_st[f"{name}/gnorm/{k}"])
            assert abs(float(p.grad.double().norm()) - nrm) <= 2e-4 * max(nrm, 1e-3), k


@pytest.mark.parametrize("tag,din,Ns", [("fst", 2, gi.CKPT_2D_N), ("tst", 3, gi.CKPT_3D_N)])
def test_shipped_checkpoints(dev, golden_ckpt, tag, din, Ns):
    """Shipped FST / 3ST weights (d=64, 8 heads -> head dim 8, 64 inducing points) on the
    eval shapes of Code/pceval.py / pc_temp3d_eval.py: [8, N, din], N from 1 to 10240,
    loaded through nn.DataParallel exactly as Code/pceval.py:46-47 does."""
    import models
    net = torch.nn.DataParallel(
        models.ST(dim_input=din, dim_hidden=64, num_heads=8, num_inds=64).to(dev),
        device_ids=[0])
    net.load_state_dict({k: T(v) for k, v in golden_ckpt.sub(f"{tag}/p/").items()})
    net.eval()
    for i, N in enumerate(Ns):
        X = T(gi.pc_input(900 + 10 * din + i, 8, N, din), dev)
        with torch.no_grad():
            lg = net(X)
        ref = golden_ckpt[f"{tag}/logits/{N}"]
        close(lg, ref, FWD_TOL, f"{tag} N={N}")
        assert (lg.argmax(1).cpu().numpy() == ref.argmax(1)).all()


def test_permutation_invariance_and_oracle(dev, golden_st):
    """Property at a larger size than the fixtures: logits do not depend on point order
    (reference: 8.9e-8) and agree with the CPU oracle on the same seeded input."""
    from oracle import st_oracle as orc
    name, B, N, din, d, h, m, C, _ = gi.ST_CASES[2]
    net = _build_st(golden_st, name, din, d, h, m, C, dev)
    X = T(gi.pc_input(4321, 16, 1025, din), dev)
    perm = torch.randperm(1025, generator=torch.Generator().manual_seed(0)).to(dev)
    with torch.no_grad():
        a = net(X)
        b = net(X[:, perm])
    close(a, b, 2e-5, "permutation invariance")
    p = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    close(a, orc.st_forward(X.cpu(), p, h), FWD_TOL, "oracle")


# ----------------------------------------------------------------------------- #
# feature extraction                                                             #
# ----------------------------------------------------------------------------- #
def test_pack_points_golden(dev, golden_dataset):
    import dataset
    g = golden_dataset
    ds = dataset.ESC_pc(g["pc2d/x"], g["pc2d/y"], g["pc2d/farr"])
    assert len(ds) == int(g["pc2d/len"])
    for i in range(len(ds)):
        pc, lbl = ds[i]
        assert pc.dtype == torch.float32 and lbl.dtype == torch.int64 and lbl.dim() == 0
        np.testing.assert_array_equal(pc.numpy(), g[f"pc2d/item{i}"])      # bit-exact
        assert int(lbl) == int(g[f"pc2d/label{i}"])
    idx = torch.tensor([4, 0, 2], device=dev)
    pts, lab = ds.batch(idx)
    for j, i in enumerate([4, 0, 2]):
        np.testing.assert_array_equal(pts[j].cpu().numpy(), g[f"pc2d/item{i}"])
    assert lab.cpu().tolist() == [int(g[f"pc2d/label{i}"]) for i in (4, 0, 2)]

    ds3 = dataset.ESC_pc_temp(g["pc3d/x"], g["pc3d/y"], g["pc3d/farr"], g["pc3d/tarr"])
    assert len(ds3) == int(g["pc3d/len"])
    for i in range(len(ds3)):
        pc, lbl = ds3[i]
        np.testing.assert_array_equal(pc.numpy(), g[f"pc3d/item{i}"])
        assert int(lbl) == int(g[f"pc3d/label{i}"])


def test_pack_points_full_size(dev):
    """BASELINE sizes through properties: every set is a gather of its frame."""
    import dataset
    from oracle import st_oracle as orc
    rng = np.random.Generator(np.random.PCG64(11))
    F, Tn = 1025, 3000
    x = rng.normal(-9, 3, size=(F, Tn)).astype(np.float32)
    y = rng.integers(0, 50, size=(Tn,))
    farr = np.linspace(0, 22050, F) / 44100
    ds = dataset.ESC_pc(x, y, farr)
    loader = dataset.DeviceBatchLoader(ds, 128, shuffle=True, seed=3)
    seen = 0
    for pts, lab in loader:
        assert pts.shape[1:] == (F, 2)
        seen += pts.shape[0]
    assert seen == Tn and len(loader) == -(-Tn // 128)
    idx = torch.tensor([0, 2999, 1234], device=dev)
    pts, lab = ds.batch(idx)
    for j, i in enumerate([0, 2999, 1234]):
        np.testing.assert_array_equal(pts[j].cpu().numpy(), orc.pack_points_2d(x, farr, i))
    F3, Nt, S = 512, 10, 40
    x3 = rng.normal(-9, 3, size=(F3, Nt, S)).astype(np.float32)
    tarr = np.linspace(0, (512 / 44100) * Nt, Nt)
    farr3 = np.linspace(0, 22050, F3) / 44100
    ds3 = dataset.ESC_pc_temp(x3, rng.integers(0, 10, size=(S,)), farr3, tarr)
    pts, _ = ds3.batch(torch.tensor([39, 7], device=dev))
    assert pts.shape == (2, 5120, 3)
    np.testing.assert_array_equal(pts[0].cpu().numpy(), orc.pack_points_3d(x3, farr3, tarr, 39))
    np.testing.assert_array_equal(pts[1].cpu().numpy(), orc.pack_points_3d(x3, farr3, tarr, 7))


def test_stft_logmag_vs_oracle(dev):
    """fp64 LDS FFT against the float64 oracle: 5e-5 abs on log-magnitude for EVERY bin,
    including near-silent ones that the log(1e-8 + .) floor amplifies."""
    import pca_hip
    from oracle import st_oracle as orc
    wave = orc.synth_clip(3, 17, seconds=0.5)
    quiet = wave.copy()
    quiet[3000:9000] = 0.0                       # exact silence -> log(1e-8) floor
    for n_fft, win, drop in ((1024, 1024, True), (2048, 2048, False), (256, 200, False),
                             (4096, 4096, False), (64, 64, False), (512, 308, False)):
        hop = n_fft // 2
        for sig in (wave, quiet):
            ref = orc.stft_logmag(sig, n_fft, win, hop, drop_nyquist=drop)
            w = T(sig, dev)
            a = pca_hip.stft_logmag(w, n_fft, win, hop, drop_nyquist=drop)
            b = pca_hip.stft_logmag(w, n_fft, win, hop, drop_nyquist=drop, frame_major=True)
            assert tuple(a.shape) == ref.shape
            assert torch.equal(a, b.t())
            a = a.cpu().numpy()
            assert np.all(np.isfinite(a))
            assert np.max(np.abs(a - ref)) < 5e-5, (n_fft, float(np.max(np.abs(a - ref))))


def test_stft_logmag_batch_equals_per_clip(dev):
    """pca_stft_logmag_batch (a corpus in one launch, clips of different lengths) is bit-identical,
    clip by clip, to pca_stft_logmag, in both output layouts."""
    import pca_hip
    from oracle import st_oracle as orc
    secs = (0.5, 0.11, 0.37, 0.05, 0.5)
    waves = [T(orc.synth_clip(20 + i, 3 * i, seconds=s), dev) for i, s in enumerate(secs)]
    for n_fft, win, drop in ((1024, 1024, True), (256, 200, False), (2048, 2048, False)):
        hop = n_fft // 2
        for fm in (False, True):
            spec, off = pca_hip.stft_logmag_batch(waves, n_fft, win, hop, drop_nyquist=drop,
                                                  frame_major=fm)
            assert off[-1] == (spec.shape[0] if fm else spec.shape[1])
            for c, w in enumerate(waves):
                one = pca_hip.stft_logmag(w, n_fft, win, hop, drop_nyquist=drop, frame_major=fm)
                got = spec[off[c]:off[c + 1]] if fm else spec[:, off[c]:off[c + 1]]
                assert torch.equal(got, one), (n_fft, fm, c)
    with pytest.raises(pca_hip.PcaHipError):      # a clip shorter than the reflect padding
        pca_hip.stft_logmag_batch([waves[0], waves[0][:100]], 1024)


# ----------------------------------------------------------------------------- #
# loss / optimiser / training trajectory                                         #
# ----------------------------------------------------------------------------- #
def test_cross_entropy_and_adam(dev):
    import pca_hip
    from pca_hip import _lib
    from oracle import st_oracle as orc
    g = torch.Generator().manual_seed(9)
    logits = (torch.randn(37, 50, generator=g) * 4)
    labels = torch.randint(0, 50, (37,), generator=g)
    lg = logits.clone().to(dev).requires_grad_(True)
    loss = pca_hip.cross_entropy(lg, labels.to(dev))
    loss.backward()
    ref_l = logits.clone().requires_grad_(True)
    ref = orc.cross_entropy(ref_l, labels)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5
    close(lg.grad, ref_l.grad, 1e-6, "dlogits")

    n = 10007
    p = torch.randn(n, generator=g)
    P = {"w": p.clone()}
    opt = orc.AdamState(P, lr=1e-3, wd=1e-3)
    pd = p.clone().to(dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    step = torch.zeros(2, dtype=torch.int32, device=dev)       # [count, ticket]
    L = _lib.lib()
    for it in range(5):
        gr = torch.randn(n, generator=g)
        opt.step(P, {"w": gr})
        gd = gr.to(dev)
        _lib.check(L.pca_adam_step(pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n,
                                   1e-3, 0.9, 0.999, 1e-8, 1e-3, 1.0, step.data_ptr(), 0, None))
    assert step.tolist() == [5, 0]
    close(pd, P["w"], 1e-6, "adam params")


def test_train_trajectory_golden(dev, golden_train):
    """20 steps of Code/settransformer.py:100-108 (CE, Adam lr 1e-3 / wd 1e-3) from the
    reference's initial weights on the same batches: loss curve within 5e-4."""
    import models
    import pca_hip
    from pca_hip import _lib
    B, N, din, d, h, m, C, steps = [int(v) for v in golden_train["cfg"]]
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    net.load_state_dict({k: T(v) for k, v in golden_train.sub("p0/").items()})
    params = list(net.parameters())
    ms = [torch.zeros_like(p) for p in params]
    vs = [torch.zeros_like(p) for p in params]
    stepc = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in params]
    L = _lib.lib()
    losses = []
    for s in range(steps):
        X = T(gi.pc_input(5000 + s, B, N, din), dev)
        y = T(gi.labels(6000 + s, B, C), dev)
        loss = pca_hip.cross_entropy(net(X), y)
        net.zero_grad(set_to_none=True)
        loss.backward()
        for p, mm, vv, sc in zip(params, ms, vs, stepc):
            _lib.check(L.pca_adam_step(p.data_ptr(), p.grad.contiguous().data_ptr(),
                                       mm.data_ptr(), vv.data_ptr(), p.numel(), 1e-3, 0.9,
                                       0.999, 1e-8, 1e-3, 1.0, sc.data_ptr(), 0, None))
        losses.append(float(loss))
    np.testing.assert_allclose(losses, golden_train["losses"], rtol=0, atol=5e-4)
    for k, v in net.state_dict().items():
        close(v, golden_train[f"p20/{k}"], 3e-3, k)


# ----------------------------------------------------------------------------- #
# whole-model engine (pca_st_*) and trainer                                      #
# ----------------------------------------------------------------------------- #
@pytest.mark.parametrize("case", gi.ST_CASES[:4], ids=[c[0] for c in gi.ST_CASES[:4]])
def test_engine_golden(dev, golden_st, case):
    """pca_st_forward / pca_st_train_fwd_bwd on flat vectors vs the reference's logits,
    loss and all 45 gradients."""
    from pca_hip import trainer
    ci = gi.ST_CASES.index(case)
    name, B, N, din, d, h, m, C, full = case
    net = _build_st(golden_st, name, din, d, h, m, C, dev)
    X = T(gi.pc_input(600 + ci, B, N, din), dev)
    y = T(gi.labels(700 + ci, B, C), dev)
    eng = trainer.STEngine(net, B, N, training=True)
    assert eng.flat.numel() == sum(p.numel() for p in net.parameters())
    names = list(net.state_dict().keys())
    assert eng.split == sum(net.state_dict()[k].numel() for k in names
                            if k.startswith("enc.0."))
    inf = trainer.STEngine(net, B, N, training=False)
    close(inf.forward(X), golden_st[f"{name}/logits"].reshape(B, C), FWD_TOL, "logits(inf)")
    for phases in ((-1,), (0, 1)):
        eng.grads.zero_()
        eng.stats.zero_()
        for ph in phases:
            eng.fwd_bwd(X, y, phase=ph)
        close(eng.logits, golden_st[f"{name}/logits"].reshape(B, C), FWD_TOL, "logits")
        assert abs(float(eng.loss) - float(golden_st[f"{name}/loss"])) < 2e-5
        assert abs(float(eng.stats[0]) / B - float(golden_st[f"{name}/loss"])) < 2e-5
        off = 0
        for k, p in net.named_parameters():
            g = eng.grads[off:off + p.numel()].view_as(p)
            off += p.numel()
            if full:
                close(g, golden_st[f"{name}/g/{k}"], BWD_TOL, k)
            else:
                close(g.reshape(-1)[::gi.GRAD_SUBSAMPLE], golden_st[f"{name}/gsub/{k}"],
                      BWD_TOL, k)


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
def test_trainer_trajectory_golden(dev, golden_train, use_graph):
    """The Trainer (pack -> engine -> fused Adam, optionally replayed from a hipGraph)
    reproduces the reference's 20-step loss curve on the same batches."""
    import dataset
    import models
    from pca_hip import trainer
    B, N, din, d, h, m, C, steps = [int(v) for v in golden_train["cfg"]]
    x = np.concatenate([gi.pc_input(5000 + s, B, N, din)[:, :, 1].T for s in range(steps)],
                       axis=1)                                   # [F=N, T=steps*B]
    y = np.concatenate([gi.labels(6000 + s, B, C) for s in range(steps)])
    ds = dataset.ESC_pc(x, y, np.linspace(0.0, 0.5, N), device=dev)
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    net.load_state_dict({k: T(v) for k, v in golden_train.sub("p0/").items()})
    tr = trainer.Trainer(net, ds, B, lr=1e-3, weight_decay=1e-3, use_graph=use_graph,
                         shuffle=False)
    losses = []
    for s in range(steps):
        tr.step()
        losses.append(float(tr.eng.loss))
    np.testing.assert_allclose(losses, golden_train["losses"], rtol=0, atol=5e-4)
    for k, v in net.state_dict().items():          # module parameters ARE the flat vector
        close(v, golden_train[f"p20/{k}"], 3e-3, k)
    loss_sum, correct = tr.read_stats()
    assert abs(loss_sum / (steps * B) - float(np.mean(golden_train["losses"]))) < 5e-4
    acc, n = trainer.evaluate(net, ds, 64)
    assert n == steps * B and 0.0 <= acc <= 1.0


def test_trainer_bf16_graph_equals_eager_bitwise(dev):
    """BASELINE cfg2 size in the bench's mode (bf16): no reduction of the step uses fp32 atomics
    (per-workgroup partials + fixed-order sums), so 12 optimiser steps by hipGraph replay, 12 by
    eager launches and a second graph run give bit-identical parameters and gradients."""
    import bench
    import models
    from pca_hip import _lib, trainer
    cfg = dict(bench.CONFIGS["cfg2"])
    ds, _ = bench.build_dataset(cfg, 4, dev, seed=0)

    def run(graph):
        torch.manual_seed(1)
        net = models.ST(dim_input=2, dim_output=50, num_inds=16, dim_hidden=128,
                        num_heads=4).to(dev)
        tr = trainer.Trainer(net, ds, 128, mode=_lib.MODE_BF16, use_graph=graph, seed=1,
                             keep_grads=True)
        for _ in range(12):
            tr.step()
        torch.cuda.synchronize()
        return tr.eng.flat.clone(), tr.eng.grads.clone()

    g1, e1, g2 = run(True), run(False), run(True)
    assert torch.isfinite(g1[0]).all() and float(g1[1].abs().max()) > 0
    for other in (e1, g2):
        assert torch.equal(g1[0], other[0])
        assert torch.equal(g1[1], other[1])


@pytest.mark.parametrize("cfgname", ["cfg2", "cfg3"])
def test_deferred_pack_equals_own_launch(dev, cfgname, monkeypatch):
    """pca_pack_defer: the step's point-set pack as rider rows of the engine's first launch
    (k_prep_all) gives bit-identical parameters to the pack as a launch of its own
    (PCA_PACK_DEFER=0), 2-D and 3-D sets, hipGraph replay; and a deferred pack that no engine call
    consumed is reported, not dropped."""
    import bench
    import models
    from pca_hip import _lib, trainer
    cfg = dict(bench.CONFIGS[cfgname])
    ds, _ = bench.build_dataset(cfg, 3, dev, seed=0)

    def run():
        torch.manual_seed(1)
        net = models.ST(dim_input=cfg["din"], dim_output=cfg["C"], num_inds=cfg["m"],
                        dim_hidden=cfg["d"], num_heads=cfg["h"]).to(dev)
        tr = trainer.Trainer(net, ds, cfg["B"], mode=_lib.MODE_BF16, use_graph=True, seed=1)
        for _ in range(6):
            tr.step()
        torch.cuda.synchronize()
        return tr.eng.flat.clone(), tr.X.clone(), tr.labels.clone()

    a = run()
    monkeypatch.setenv("PCA_PACK_DEFER", "0")
    b = run()
    assert torch.isfinite(a[0]).all()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    monkeypatch.delenv("PCA_PACK_DEFER")
    # an armed, never-consumed pack must not vanish silently
    L = _lib.lib()
    _lib.check(L.pca_pack_defer(1))
    idx = torch.arange(cfg["B"], device=dev)
    step = torch.zeros(2, dtype=torch.int32, device=dev)
    base = torch.zeros(1, dtype=torch.int32, device=dev)
    keep = ds.batch_seq(idx, step, base, cfg["B"])      # (its outputs stay alive: the pack is pending)
    assert L.pca_pack_defer(0) != 0
    assert b"never consumed" in L.pca_last_error()
    # ... and is still there for the engine call that follows
    torch.manual_seed(1)
    net = models.ST(dim_input=cfg["din"], dim_output=cfg["C"], num_inds=cfg["m"],
                    dim_hidden=cfg["d"], num_heads=cfg["h"]).to(dev)
    eng = trainer.STEngine(net, cfg["B"], ds.num_points, _lib.MODE_BF16, training=False)
    X = torch.zeros(cfg["B"], ds.num_points, cfg["din"], device=dev)
    eng.forward(X)
    _lib.check(L.pca_pack_defer(0))
    torch.cuda.synchronize()
    assert float(keep[0].abs().max()) > 0           # the pack did run


def test_trainer_full_size_graph_vs_eager_vs_oracle(dev):
    """BASELINE cfg2 size (B=128, N=512, d=128, h=4, m=16, C=50) on STFT-derived synthetic
    clips: (1) hipGraph replay and eager launches give the same parameters after 12 steps
    with NO host sync in between (regression: hipMemsetAsync nodes inside a captured graph
    corrupted scratch buffers at this size); (2) the loss curve follows the CPU oracle fed
    the same batches."""
    import bench
    import models
    from oracle import st_oracle as orc
    from pca_hip import trainer
    cfg = dict(bench.CONFIGS["cfg2"])
    ds, _ = bench.build_dataset(cfg, 4, dev, seed=0)
    assert ds.num_points == 512

    def make(graph):
        torch.manual_seed(1)
        net = models.ST(dim_input=2, dim_output=50, num_inds=16, dim_hidden=128,
                        num_heads=4).to(dev)
        return net, trainer.Trainer(net, ds, 128, use_graph=graph, seed=1, keep_grads=True)

    net_g, tr_g = make(True)
    net_e, tr_e = make(False)
    for _ in range(12):
        tr_g.step()
    for _ in range(12):
        tr_e.step()
    torch.cuda.synchronize()
    assert float(tr_g.eng.grads.abs().max()) < 1e3
    close(tr_g.eng.flat, tr_e.eng.flat, 1e-4, "graph vs eager parameters")  # split-K atomics: order noise
    sg, se = tr_g.read_stats(), tr_e.read_stats()
    assert abs(sg[0] - se[0]) < 1e-3 * se[0] and sg[1] == se[1]

    net_o, tr_o = make(True)
    p = {k: v.detach().cpu().clone() for k, v in net_o.state_dict().items()}
    opt = orc.AdamState(p)
    torch.set_num_threads(min(16, torch.get_num_threads() * 2))
    for s in range(6):
        tr_o.step()
        loss, _ = orc.train_step(tr_o.X.cpu(), tr_o.labels.cpu(), p, opt, 4)
        assert abs(float(tr_o.eng.loss) - loss) < 2e-3, (s, float(tr_o.eng.loss), loss)


def test_reframe_sweep(dev, golden_ckpt, tmp_path):
    """Device counterpart of the re-framing loop of Code/pceval.py:61-104 with the shipped FST
    weights: every analysis length gives a well-formed accuracy, N = Nfft reproduces a direct
    evaluation on the oracle's STFT, and the JSON has the reference's structure."""
    import json
    import evalsweep
    import models
    from oracle import st_oracle as orc
    from pca_hip import trainer
    net = models.ST(dim_input=2, dim_hidden=64, num_heads=8, num_inds=64).to(dev)
    net.load_state_dict({k[len("module."):]: T(v) for k, v in golden_ckpt.sub("fst/p/").items()})
    fs, Nfft = 44100, 2048
    waves = [orc.synth_clip(i, i % 10, seconds=0.6, fs=fs) for i in range(4)]
    labels = [i % 10 for i in range(4)]
    clips = [T(w, dev) for w in waves]
    list_N = [2 * Nfft, int(1.25 * Nfft), Nfft, int(0.6 * Nfft), int(0.1 * Nfft)]
    jf = str(tmp_path / "expt1.json")
    out = evalsweep.reframe_sweep(net, clips, labels, fs, list_N, json_file=jf)
    assert out["list_N"] == list_N and list(out["data"].keys()) == [fs]
    assert all(0.0 <= a <= 1.0 for a in out["data"][fs])
    back = json.load(open(jf))
    assert back["list_N"] == list_N and len(back["data"][str(fs)]) == len(list_N)
    # N = Nfft against a direct evaluation on the oracle's spectrogram (same skipped tail)
    import dataset
    specs = [orc.stft_logmag(w, Nfft) for w in waves]
    x = np.concatenate(specs, 1)
    y = np.concatenate([np.full(s.shape[1], l) for s, l in zip(specs, labels)])
    full = (x.shape[1] // 8) * 8
    ds = dataset.ESC_pc(x[:, :full], y[:full], np.linspace(0, fs / 2, x.shape[0]) / fs, device=dev)
    acc, n = trainer.evaluate(net, ds, 8)
    assert n == full and abs(acc - out["data"][fs][2]) <= 1.0 / full + 1e-9
    # the read side of Code/pceval.py:23-47 composes with the sweep: load_run hands over the
    # nn.DataParallel-wrapped model on the device and the engine takes it as it is
    import runfiles
    import test_runfiles
    stem = str(tmp_path / "FST(x)")
    torch.save({k: T(v) for k, v in golden_ckpt.sub("fst/p/").items()}, stem + "_net.pth")
    json.dump(test_runfiles.FST_SHIPPED, open(stem + "_config.json", "w"))
    wrapped, _ = runfiles.load_run(stem + "_config.json")
    assert isinstance(wrapped, torch.nn.DataParallel)
    assert next(wrapped.parameters()).is_cuda
    out2 = evalsweep.reframe_sweep(wrapped, clips, labels, fs, list_N[2:4])
    assert out2["data"][fs] == out["data"][fs][2:4]
    acc2, _ = trainer.evaluate(wrapped, ds, 8)
    assert acc2 == acc
    # shorter window than n_fft: frame count and bin count follow librosa's framing
    d = evalsweep.framewise_dataset(clips, labels, fs, int(0.6 * Nfft))
    assert d.num_points == 1 + 2048 // 2
    assert len(d) == sum(1 + len(w) // int(0.6 * Nfft * 0.5) for w in waves)


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hipgraph"])
def test_device_cursor_equals_per_step_indices(dev, graph):
    """The device-side batch cursor (index batches of an epoch staged once, pack kernel picks
    batch step_count - epoch_base) against the per-step index upload, across three epoch
    boundaries, 2-D and padded 3-D datasets."""
    import dataset
    import models
    from pca_hip import trainer
    rng = np.random.Generator(np.random.PCG64(31))
    F, T_, C, B = 24, 83, 5, 16                       # 5 steps per epoch, ragged tail dropped
    x = rng.normal(-9, 3, size=(F, T_)).astype(np.float32)
    y = rng.integers(0, C, size=T_)
    Ft, Nt, S = 8, 3, 70
    x3 = rng.normal(-9, 3, size=(Ft, Nt, S)).astype(np.float32)
    y3 = rng.integers(0, C, size=S)
    ntv = rng.integers(1, Nt + 1, size=S).astype(np.int32)

    def run(ds, din, cursor):
        torch.manual_seed(4)
        net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=4, dim_hidden=16,
                        num_heads=4).to(dev)
        tr = trainer.Trainer(net, ds, B, use_graph=graph, seed=2, shuffle=True)
        assert tr._cursor_mode
        tr._cursor_mode = cursor
        for _ in range(17):
            tr.step()
        torch.cuda.synchronize()
        return tr.eng.flat.clone(), tr.read_stats()

    for ds, din in ((dataset.ESC_pc(x, y, np.linspace(0, 0.5, F), device=dev), 2),
                    (dataset.ESC_pc_temp(x3, y3, np.linspace(0, 0.5, Ft), np.linspace(0, 0.1, Nt),
                                         device=dev, nt_valid=ntv), 3)):
        a, sa = run(ds, din, True)
        b, sb = run(ds, din, False)
        close(a, b, 1e-5, "cursor vs uploaded indices")
        assert abs(sa[0] - sb[0]) < 1e-3 * abs(sb[0]) and sa[1] == sb[1]
