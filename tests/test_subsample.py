"""Sub-sampling datasets: the CPU oracle against the items the
REAL reference produced (golden_ss.npz, golden_dataset.npz), and the HIP selection kernel
against both."""
import numpy as np
import pytest
import torch

import inputs as gi
from conftest import Golden
from util import T, same_up_to_ties


@pytest.fixture(scope="module")
def golden_ss():
    return Golden("golden_ss.npz")


def _sel_of(pc, farr, tarr):
    F = farr.shape[0]
    f = np.searchsorted(farr, pc[:, 0])
    t = np.searchsorted(tarr, pc[:, 1])
    return (t * F + f).astype(np.int32)


def test_oracle_maxk_full_size(golden_ss):
    from oracle import st_oracle as orc
    x, y, farr, tarr = gi.ss_inputs()
    for K in gi.SS_K:
        for i in range(x.shape[2]):
            pc = orc.pc_maxk_3d(x, farr, tarr, i, K)
            assert pc.dtype == np.float64 and pc.shape == (K, 3)
            np.testing.assert_array_equal(pc[:, 2].astype(np.float32),
                                          golden_ss[f"maxK{K}/val{i}"])
            same_up_to_ties(_sel_of(pc, farr, tarr), golden_ss[f"maxK{K}/sel{i}"], pc[:, 2],
                            f"K={K} item {i}")


def test_oracle_pc_ss_and_utils(golden_ss):
    from oracle import st_oracle as orc
    x, y, farr, tarr = gi.ss_inputs()
    x2 = x[:, 0, :]
    xs, fs = orc.pc_maxk_2d(x2, farr, 51)
    assert xs.shape == fs.shape == (51, x2.shape[1])
    assert int(golden_ss["ss2d/len"]) == x2.shape[1]
    for i in range(x2.shape[1]):
        np.testing.assert_array_equal(orc.pack_points_2d_ss(xs, fs, i), golden_ss[f"ss2d/item{i}"])
    rep = orc.pc_maxk_replace(x2, 51)
    assert rep.dtype == np.float64 and rep.shape == x2.shape
    for i in range(x2.shape[1]):
        keep = np.zeros(x2.shape[0], bool)
        keep[golden_ss[f"ss2d/sel{i}"]] = True
        np.testing.assert_array_equal(rep[keep, i], x2[keep, i].astype(np.float64))
        assert (rep[~keep, i] == 0).all()


# ------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


@pytest.mark.gpu
def test_maxkss_items_and_batches(dev, golden_ss, golden_dataset):
    """ESC_pc_temp_maxKSS: items float64 [K, 3] equal to the reference's; batches float32."""
    import dataset
    x, y, farr, tarr = gi.ss_inputs()
    F = farr.shape[0]
    for K in gi.SS_K:
        ds = dataset.ESC_pc_temp_maxKSS(x, y, farr, tarr, K, device=dev)
        assert len(ds) == x.shape[2] and ds.num_points == K
        idx = torch.arange(x.shape[2], device=dev)
        pts, lab, sel = ds.batch(idx, want_sel=True)
        assert pts.shape == (x.shape[2], K, 3) and pts.dtype == torch.float32
        assert lab.cpu().tolist() == y.tolist()
        for i in range(x.shape[2]):
            ref_sel, ref_val = golden_ss[f"maxK{K}/sel{i}"], golden_ss[f"maxK{K}/val{i}"]
            np.testing.assert_array_equal(pts[i, :, 2].cpu().numpy(), ref_val)
            s = sel[i].cpu().numpy()
            same_up_to_ties(s, ref_sel, ref_val, f"K={K} set {i}")
            np.testing.assert_array_equal(pts[i, :, 0].cpu().numpy(),
                                          farr[s % F].astype(np.float32))
            np.testing.assert_array_equal(pts[i, :, 1].cpu().numpy(),
                                          tarr[s // F].astype(np.float32))
            pc, lbl = ds[i]
            assert pc.dtype == torch.float64 and tuple(pc.shape) == (K, 3)
            assert int(lbl) == int(golden_ss[f"maxK{K}/label{i}"])
            np.testing.assert_array_equal(pc[:, 2].numpy().astype(np.float32), ref_val)
    # the small fixture of golden_dataset.npz (exact float64 items, no ties)
    g = golden_dataset
    x3, y3, f3, t3 = g["pc3d/x"], g["pc3d/y"], g["pc3d/farr"], g["pc3d/tarr"]
    for K in (1, 7, 24):
        ds = dataset.ESC_pc_temp_maxKSS(x3, y3, f3, t3, K, device=dev)
        for i in range(3):
            np.testing.assert_array_equal(ds[i][0].numpy(), g[f"pc3d/maxK{K}/item{i}"])


@pytest.mark.gpu
def test_maxk_ties_and_ragged_sizes(dev):
    """Equal values keep ascending point order (the oracle's stable argsort); N that is not a
    power of two, K = 1, K = N, NaN last, -0 == +0."""
    import pca_hip
    from oracle import st_oracle as orc
    rng = np.random.Generator(np.random.PCG64(5))
    for F, Nt in ((7, 3), (100, 1), (1025, 1), (513, 9), (512, 20), (16384, 1)):
        S = 4
        x = rng.integers(-6, 3, size=(F, Nt, S)).astype(np.float32)       # many ties
        x[0, 0, 1] = -0.0
        farr = np.linspace(0, 0.5, F)
        tarr = np.linspace(0, 0.1, Nt)
        N = F * Nt
        for K in sorted({1, min(5, N), N // 2, N}):
            pts, _, sel = pca_hip.subsample_points(
                T(x, dev), T(farr.astype(np.float32), dev),
                T(tarr.astype(np.float32), dev) if Nt > 1 else None,
                torch.arange(S, device=dev), K, want_sel=True) if Nt > 1 else \
                pca_hip.subsample_points(T(x[:, 0, :], dev), T(farr.astype(np.float32), dev),
                                         None, torch.arange(S, device=dev), K, want_sel=True)
            for i in range(S):
                ref = orc.pc_maxk_3d(x, farr, tarr, i, K)
                np.testing.assert_array_equal(sel[i].cpu().numpy(), _sel_of(ref, farr, tarr),
                                              err_msg=f"F={F} Nt={Nt} K={K} set {i}")
                np.testing.assert_array_equal(pts[i, :, -1].cpu().numpy(),
                                              ref[:, 2].astype(np.float32))
    xn = np.array([[1.0, np.nan, 3.0, -np.inf, np.inf, 2.0]], dtype=np.float32).T   # [6, 1]
    _, _, sel = pca_hip.subsample_points(T(xn, dev), T(np.zeros(6, np.float32), dev), None,
                                         torch.zeros(1, dtype=torch.int64, device=dev), 6,
                                         want_sel=True)
    assert sel[0].cpu().tolist() == [4, 2, 5, 0, 3, 1]                    # NaN last (numpy)


@pytest.mark.gpu
def test_randk_is_a_uniform_subset(dev):
    """ESC_pc_temp_randKSS: every item is K distinct points of the set with their own values;
    over many draws each point is kept with probability K/N and first positions are uniform
    (the reference's np.random.permutation(N)[:K], Code/dataset.py:236; stream differs)."""
    import dataset
    F, Nt, S, K = 32, 4, 2, 16
    N = F * Nt
    rng = np.random.Generator(np.random.PCG64(9))
    x = rng.normal(-9, 3, size=(F, Nt, S)).astype(np.float32)
    farr, tarr = np.linspace(0, 0.5, F), np.linspace(0, 0.1, Nt)
    ds = dataset.ESC_pc_temp_randKSS(x, np.arange(S), farr, tarr, K, device=dev, seed=3)
    idx = torch.zeros(2048, dtype=torch.int64, device=dev)
    counts, first = np.zeros(N), np.zeros(N)
    draws = 8
    for _ in range(draws):
        pts, lab, sel = ds.batch(idx, want_sel=True)
        s = sel.cpu().numpy()
        assert ((s >= 0) & (s < N)).all()
        assert all(len(set(r)) == K for r in s[:64])
        np.testing.assert_array_equal(pts[:, :, 2].cpu().numpy(),
                                      x[:, :, 0].T.reshape(-1)[s])          # p = t*F + f
        np.testing.assert_array_equal(pts[:, :, 0].cpu().numpy(),
                                      farr.astype(np.float32)[s % F])
        counts += np.bincount(s.reshape(-1), minlength=N)
        first += np.bincount(s[:, 0], minlength=N)
    n = draws * 2048
    p = K / N
    z = (counts - n * p) / np.sqrt(n * p * (1 - p))
    assert np.abs(z).max() < 5.0, np.abs(z).max()
    zf = (first - n / N) / np.sqrt(n / N * (1 - 1 / N))
    assert np.abs(zf).max() < 5.0, np.abs(zf).max()
    # a different draw number gives a different selection; the same (seed, draw) repeats
    a = ds.batch(idx[:4], want_sel=True)[2].cpu().numpy()
    ds2 = dataset.ESC_pc_temp_randKSS(x, np.arange(S), farr, tarr, K, device=dev, seed=3)
    for _ in range(draws):
        ds2.batch(idx[:4])
    b = ds2.batch(idx[:4], want_sel=True)[2].cpu().numpy()
    np.testing.assert_array_equal(a, b)
    item, lbl = ds[1]
    assert item.dtype == torch.float64 and tuple(item.shape) == (K, 3) and int(lbl) == 1


@pytest.mark.gpu
def test_utils_and_pc_ss(dev, golden_ss):
    """utils.pc_maxK / pc_randK / *_replace and ESC_pc_ss against the oracle and the items the
    reference's ESC_pc_ss produced."""
    import dataset
    import utils
    from oracle import st_oracle as orc
    x, y, farr, tarr = gi.ss_inputs()
    x2 = x[:, 0, :]
    xs, fs = utils.pc_maxK(x2, farr, 51)
    oxs, ofs = orc.pc_maxk_2d(x2, farr, 51)
    assert xs.dtype == x2.dtype and fs.dtype == farr.dtype
    np.testing.assert_array_equal(xs, oxs)
    np.testing.assert_array_equal(fs, ofs)
    ds = dataset.ESC_pc_ss(xs, y, fs, device=dev)
    assert len(ds) == int(golden_ss["ss2d/len"])
    pts, lab = ds.batch(torch.arange(x2.shape[1], device=dev))
    for i in range(x2.shape[1]):
        np.testing.assert_array_equal(pts[i].cpu().numpy(), golden_ss[f"ss2d/item{i}"])
        np.testing.assert_array_equal(ds[i][0].numpy(), golden_ss[f"ss2d/item{i}"])
    assert lab.cpu().tolist() == y.tolist()
    np.testing.assert_array_equal(utils.pc_maxK_replace(x2, 51), orc.pc_maxk_replace(x2, 51))
    xr, fr = utils.pc_randK(x2, farr, 40, seed=1)
    assert xr.shape == fr.shape == (40, x2.shape[1])
    for i in range(x2.shape[1]):
        pos = np.searchsorted(farr, fr[:, i])
        assert len(set(pos)) == 40
        np.testing.assert_array_equal(xr[:, i], x2[pos, i])
    rr = utils.pc_randK_replace(x2, 40, seed=2)
    assert rr.dtype == np.float64 and ((rr != 0).sum(0) == 40).all()
    assert ((rr == 0) | (rr == x2)).all()
    model = torch.nn.Linear(3, 2)
    assert utils.count_parameters(model) == 8


# -------------------------------------------------------------------- importance sampling
def test_oracle_importance(golden_ss):
    """Oracle heat map against the reference's torch expressions; top-K rows against the
    reference dataset's items (the q-th hottest cell must carry the same heat)."""
    from oracle import st_oracle as orc
    x, y, farr, tarr = gi.ss_inputs()
    F = farr.shape[0]
    for winF in gi.IMP_WINF:
        for i in range(x.shape[2]):
            heat = orc.importance_heat(x[:, :, i], winF)
            ref_heat = golden_ss[f"imp{winF}/heat{i}"]
            np.testing.assert_allclose(heat, ref_heat, rtol=2e-6, atol=1e-6)
            for K in (51, 2551):
                pc = orc.pc_importance_topk(x, farr, tarr, i, K, winF)
                sel = _sel_of(pc, farr, tarr)
                ref_sel = golden_ss[f"imp{winF}/K{K}/sel{i}"]
                assert len(set(sel.tolist())) == K
                hq = ref_heat.reshape(-1)
                np.testing.assert_allclose(hq[sel], hq[ref_sel], rtol=5e-6, atol=0)
                assert (sel == ref_sel).mean() > 0.99
                same = sel == ref_sel
                np.testing.assert_array_equal(pc[same, 2].astype(np.float32),
                                              golden_ss[f"imp{winF}/K{K}/val{i}"][same])


@pytest.mark.gpu
def test_importance_topk_and_heat(dev, golden_ss):
    import dataset
    x, y, farr, tarr = gi.ss_inputs()
    F = farr.shape[0]
    idx = torch.arange(x.shape[2], device=dev)
    for winF in gi.IMP_WINF:
        for K in (51, 2551):
            ds = dataset.ESC_pc_temp_importancerandKSS(x, y, farr, tarr, K, 1, winF, device=dev)
            pts, lab, sel, heat = ds.batch(idx, want_sel=True, want_heat=True)
            assert lab.cpu().tolist() == y.tolist()
            for i in range(x.shape[2]):
                ref_heat = golden_ss[f"imp{winF}/heat{i}"]
                np.testing.assert_allclose(heat[i].cpu().numpy(), ref_heat, rtol=2e-6, atol=1e-6)
                s = sel[i].cpu().numpy()
                ref_sel = golden_ss[f"imp{winF}/K{K}/sel{i}"]
                assert len(set(s.tolist())) == K
                hq = ref_heat.reshape(-1)
                np.testing.assert_allclose(hq[s], hq[ref_sel], rtol=5e-6, atol=0)
                assert (s == ref_sel).mean() > 0.99
                # rows are row s of the time-major point table, as in the reference
                np.testing.assert_array_equal(pts[i, :, 0].cpu().numpy(),
                                              farr.astype(np.float32)[s % F])
                np.testing.assert_array_equal(pts[i, :, 1].cpu().numpy(),
                                              tarr.astype(np.float32)[s // F])
                np.testing.assert_array_equal(pts[i, :, 2].cpu().numpy(), x[s % F, s // F, i])
                same = s == ref_sel
                np.testing.assert_array_equal(pts[i, :, 2].cpu().numpy()[same],
                                              golden_ss[f"imp{winF}/K{K}/val{i}"][same])
            item, lbl = ds[1]
            assert item.dtype == torch.float64 and tuple(item.shape) == (K, 3)


@pytest.mark.gpu
def test_importance_multinomial_distribution(dev):
    """choice 0: draws with replacement follow heat / sum(heat) (torch.multinomial in the
    reference; the stream differs).  Chi-square style check per cell on a small set."""
    import dataset
    from oracle import st_oracle as orc
    F, Nt, K, winF = 16, 6, 64, 3
    rng = np.random.Generator(np.random.PCG64(11))
    x = rng.normal(-9, 3, size=(F, Nt, 1)).astype(np.float32)
    farr, tarr = np.linspace(0, 0.5, F), np.linspace(0, 0.1, Nt)
    ds = dataset.ESC_pc_temp_importancerandKSS(x, np.zeros(1, np.int64), farr, tarr, K, 0, winF,
                                               device=dev, seed=5)
    idx = torch.zeros(4096, dtype=torch.int64, device=dev)
    counts = np.zeros(F * Nt)
    for _ in range(4):
        pts, lab, sel = ds.batch(idx, want_sel=True)
        s = sel.cpu().numpy()
        assert ((s >= 0) & (s < F * Nt)).all()
        counts += np.bincount(s.reshape(-1), minlength=F * Nt)
        np.testing.assert_array_equal(pts[:, :, 2].cpu().numpy(), x[s % F, s // F, 0])
    heat = orc.importance_heat(x[:, :, 0], winF).reshape(-1)
    p = heat / heat.sum()
    n = counts.sum()
    z = (counts - n * p) / np.sqrt(n * p * (1 - p))
    assert np.abs(z).max() < 5.0, np.abs(z).max()
    # K may exceed the number of points when drawing with replacement
    big = dataset.ESC_pc_temp_importancerandKSS(x, np.zeros(1, np.int64), farr, tarr, 500, 0,
                                                winF, device=dev)
    assert tuple(big.batch(idx[:2])[0].shape) == (2, 500, 3)


@pytest.mark.gpu
def test_random_selection_advances_under_graph_replay(dev):
    """A train step captured into a hipGraph must draw a NEW random-K subset on every replay:
    the draw number reaches the kernel through a device counter (the optimiser's step count),
    not as a by-value argument frozen at capture time."""
    import dataset
    import models
    from pca_hip import _lib, trainer
    F, Nt, S, K, B = 64, 4, 8, 37, 8
    rng = np.random.Generator(np.random.PCG64(31))
    x = rng.normal(-9, 3, size=(F, Nt, S)).astype(np.float32)
    y = rng.integers(0, 5, size=S)
    farr, tarr = np.linspace(0, 0.5, F), np.linspace(0, 0.1, Nt)
    for cls, kw in ((dataset.ESC_pc_temp_randKSS, {}),
                    (dataset.ESC_pc_temp_importancerandKSS, dict(choice=0, winF=5))):
        ds = cls(x, y, farr, tarr, K, device=dev, seed=5, **kw)
        assert ds.stochastic
        torch.manual_seed(2)
        net = models.ST(dim_input=3, num_outputs=1, dim_output=5, num_inds=4, dim_hidden=16,
                        num_heads=4).to(dev)
        tr = trainer.Trainer(net, ds, B, mode=_lib.MODE_F32, use_graph=True, shuffle=False)
        seen = []
        for _ in range(4):
            tr.step()
            torch.cuda.synchronize()
            seen.append(tr.X.clone())
        # same sets every step (shuffle off, S == B), different selections
        for a in range(len(seen)):
            for b in range(a + 1, len(seen)):
                assert not torch.equal(seen[a], seen[b]), (cls.__name__, a, b)
    assert not dataset.ESC_pc_temp_maxKSS(x, y, farr, tarr, K, device=dev).stochastic
