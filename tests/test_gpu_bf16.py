"""GPU parity of the fused bf16 MFMA kernels (PCA_MODE_BF16) against the CPU oracle.

Tolerance: bf16 MFMA operands (8 significant bits) with fp32 accumulation, softmax, biases
and residuals.  Measured error of the fused block is ~3e-3 of max|Y|; the tests allow 1.5e-2
relative to max(1, max|ref|) for activations and 3e-2 for gradients.
"""
import numpy as np
import pytest
import torch

from emu import mab0_forward_bf16emu
from util import T, close, close_robust

pytestmark = pytest.mark.gpu

FWD_TOL = 1.5e-2
BWD_TOL = 3e-2


@pytest.fixture(scope="module")
def dev():
    import pca_hip
    pca_hip.lib()
    yield torch.device("cuda", 0)
    pca_hip.set_mode("f32")


@pytest.fixture(autouse=True)
def _guard_workspaces():
    """Every scratch / saved block the autograd glue hands to the library in this file is followed
    by a guard region, verified after the test (pca_hip.ops.check_canaries)."""
    from pca_hip import ops
    ops.CANARY = True
    ops._guards.clear()
    try:
        yield
        ops.check_canaries()
    finally:
        ops.CANARY = False
        ops._guards.clear()


def _mab_params(dq, dk, d, seed):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for nm, din in (("fc_q", dq), ("fc_k", dk), ("fc_v", dk), ("fc_o", d)):
        bound = 1.0 / np.sqrt(din)
        p[nm + ".weight"] = (torch.rand(d, din, generator=g) * 2 - 1) * bound
        p[nm + ".bias"] = (torch.rand(d, generator=g) * 2 - 1) * bound
    return p


MAB1_CASES = [      # B, N, m, dq, d, h
    (3, 200, 16, 128, 128, 4),        # ragged N (tile of 128)
    (2, 128, 32, 128, 128, 4),
    (2, 77, 16, 2, 128, 4),           # layer 1: exact VALU projection of (f, logmag)
    (1, 1, 16, 3, 128, 4),            # a single point
    (5, 512, 16, 128, 128, 4),
]


# BASELINE configs[3] block (d = 256, 8 heads, m = 32): Q phase + O phase for d -> d (one launch
# for layer 1); the backward is three launches + the 256-wide weight-gradient reduction
MAB1_D256_CASES = [
    (2, 300, 32, 256, 256, 8),        # ragged N
    (3, 128, 32, 256, 256, 8),
    (2, 77, 32, 3, 256, 8),           # layer 1
    (1, 1, 32, 256, 256, 8),          # a single point
    # 65 536 rows with N < 512: wgrad256_nwg() gives the long jobs 64 workgroups and the short
    # [B m]-row jobs of the same workspace 128 (the slabs ran past the workspace)
    (256, 256, 32, 256, 256, 8),
]


@pytest.mark.parametrize("case", MAB1_CASES + MAB1_D256_CASES,
                         ids=[str(c) for c in MAB1_CASES + MAB1_D256_CASES])
def test_mab1_fwd_bf16(dev, case):
    import modules
    import pca_hip
    from oracle import st_oracle as orc
    B, N, m, dq, d, h = case
    p = _mab_params(dq, d, d, seed=sum(case))
    g = torch.Generator().manual_seed(1 + sum(case))
    X = torch.randn(B, N, dq, generator=g)
    if dq <= 4:
        X[..., -1] = X[..., -1] * 3 - 9                      # log-magnitude-like column
    H = torch.randn(B, m, d, generator=g)
    ref = orc.mab_forward(X, H, p, h)
    mab = modules.MAB(dq, d, d, h).to(dev)
    mab.load_state_dict(p)
    pca_hip.set_mode("bf16")
    with torch.no_grad():
        Y = mab(X.to(dev), H.to(dev))
    pca_hip.set_mode("f32")
    err = close(Y, ref, FWD_TOL, f"mab1 fwd {case}")
    print(f"mab1 fwd {case}: max err {err:.3e} (max|ref| {float(ref.abs().max()):.2f})")


@pytest.mark.parametrize("case", MAB1_CASES + MAB1_D256_CASES,
                         ids=[str(c) for c in MAB1_CASES + MAB1_D256_CASES])
def test_mab1_bwd_bf16(dev, case):
    """Fused backward chain + MFMA weight-gradient reductions.

    Checked against autograd of the oracle's bf16-operand emulation (same rounding points as
    the kernel, hence the same ReLU mask): that isolates kernel correctness from the
    discontinuity of ReLU' under reduced precision.  The emulation itself is held to the exact
    fp32 oracle in rms (the precision statement of the bf16 mode)."""
    import modules
    import pca_hip
    from oracle import st_oracle as orc
    B, N, m, dq, d, h = case
    p = _mab_params(dq, d, d, seed=sum(case))
    g = torch.Generator().manual_seed(1 + sum(case))
    X = torch.randn(B, N, dq, generator=g)
    if dq <= 4:
        X[..., -1] = X[..., -1] * 3 - 9
    H = torch.randn(B, m, d, generator=g)
    G = torch.randn(B, N, d, generator=g)
    exact = orc.mab_backward(G, X, H, p, h)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    Xe, He = X.clone().requires_grad_(True), H.clone().requires_grad_(True)
    Ye = orc.mab1_forward_bf16emu(Xe, He, leaves, h)
    (Ye * G).sum().backward()
    emu = {k: v.grad for k, v in leaves.items()}
    emu["dQ"], emu["dK"] = Xe.grad, He.grad

    mab = modules.MAB(dq, d, d, h).to(dev)
    mab.load_state_dict(p)
    Xd = X.to(dev).requires_grad_(dq > 4)
    Hd = H.to(dev).requires_grad_(True)
    pca_hip.set_mode("bf16")
    Y = mab(Xd, Hd)
    (Y * G.to(dev)).sum().backward()
    pca_hip.set_mode("f32")
    # (d = 256: O crosses from the Q phase to the O phase in bf16, so the residual path of Y
    #  carries one more rounding of relative size 2^-9 than the emulation)
    close(Y, Ye, 2e-3 if d == 128 else 6e-3, "Y vs bf16 emulation")
    got = {"dK": Hd.grad}
    if dq > 4:
        got["dQ"] = Xd.grad
    for k, prm in mab.named_parameters():
        got[k] = prm.grad
    errs = {}
    for k, v in got.items():
        if k == "fc_k.bias":
            # d/d(bk) is identically 0 (softmax is shift invariant): what is left is the
            # rounding noise of sums of dKp rows, so judge it on the scale of d/d(Wk)
            sc = max(1.0, float(emu["fc_k.weight"].abs().max()))
            assert float((v.cpu() - emu[k]).abs().max()) <= 1.5e-2 * sc
        else:
            # (d = 256: dO, dQp and dZ additionally cross HBM in bf16 between the three launches)
            errs[k] = close_robust(v, emu[k], 1.5e-2, k, outlier_frac=2e-4 if d == 128 else 2e-3)
        sc = max(1.0, float(exact[k].abs().max()))
        if k == "fc_k.bias":
            sc = max(1.0, float(exact["fc_k.weight"].abs().max()))
        rms = float((emu[k] - exact[k]).pow(2).mean().sqrt()) / sc
        assert rms < 3e-2, (k, rms)     # bf16 emulation vs exact fp32 (incl. ReLU-mask flips)
    print(f"mab1 bwd {case}: " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))


@pytest.mark.parametrize("d,h,m", [(128, 4, 16), (256, 8, 32)])
def test_mab1_fwd_bf16_propagates_nan(dev, d, h, m):
    """A NaN / Inf in X or H must come out as a non-finite Y (never as finite garbage): the d = 256
    forward takes its softmax maxima with v_med3_f32 (max_nn, mfma_common.hpp; until round 4 the file was
    compiled with -fno-honor-nans), the contract is checked here."""
    import modules
    import pca_hip
    B, N = 2, 130
    p = _mab_params(d, d, d, seed=5)
    g = torch.Generator().manual_seed(6)
    X = torch.randn(B, N, d, generator=g)
    H = torch.randn(B, m, d, generator=g)
    mab = modules.MAB(d, d, d, h).to(dev)
    mab.load_state_dict(p)
    pca_hip.set_mode("bf16")
    try:
        with torch.no_grad():
            for bad in (float("nan"), float("inf")):
                Xn = X.clone()
                Xn[1, 77, 5] = bad
                Y = mab(Xn.to(dev), H.to(dev)).cpu()
                assert not torch.isfinite(Y[1, 77]).all(), "non-finite point came out finite"
                assert torch.isfinite(Y[0]).all() and torch.isfinite(Y[1, :77]).all()
                Hn = H.clone()
                Hn[0, 3, 9] = bad
                Y = mab(X.to(dev), Hn.to(dev)).cpu()
                assert not torch.isfinite(Y[0]).any(dim=-1).all(), "non-finite key came out finite"
                assert torch.isfinite(Y[1]).all()
    finally:
        pca_hip.set_mode("f32")


def test_engine_bf16_vs_golden(dev, golden_st):
    """Whole ST (BASELINE cfg1/2 architecture) with mode = BF16: fused kernels where built,
    exact fp32 elsewhere; logits and all gradients vs the reference's golden vectors."""
    import inputs as gi
    import models
    from pca_hip import _lib, trainer
    ci = 2
    name, B, N, din, d, h, m, C, full = gi.ST_CASES[ci]
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    net.load_state_dict({k: T(v) for k, v in golden_st.sub(f"{name}/p/").items()})
    X = T(gi.pc_input(600 + ci, B, N, din), dev)
    y = T(gi.labels(700 + ci, B, C), dev)
    eng = trainer.STEngine(net, B, N, mode=_lib.MODE_BF16, training=True)
    eng.fwd_bwd(X, y, phase=-1)
    ref = golden_st[f"{name}/logits"]
    err = close(eng.logits, ref, FWD_TOL, "logits")
    assert (eng.logits.argmax(1).cpu().numpy() == ref.argmax(1)).all()
    off, worst = 0, 0.0
    for k, prm in net.named_parameters():
        gr = eng.grads[off:off + prm.numel()].view_as(prm)
        off += prm.numel()
        worst = max(worst, close_robust(gr, golden_st[f"{name}/g/{k}"], BWD_TOL, k,
                                        outlier_frac=2e-3))
    print(f"engine bf16: logits err {err:.2e}, worst grad err {worst:.2e}")
    inf = trainer.STEngine(net, B, N, mode=_lib.MODE_BF16, training=False)
    close(inf.forward(X), ref, FWD_TOL, "logits(inference)")


MAB0_CASES = [      # B, N, m, dk, d, h   (query = learned [m, d], shared by all sets)
    (3, 200, 16, 128, 128, 4),      # ISAB mab0, ragged N
    (4, 130, 1, 128, 128, 4),       # PMA: one seed, 4 score rows padded to 16
    (3, 77, 16, 2, 128, 4),         # layer 1 (f, logmag): exact fp32
    (2, 33, 16, 3, 128, 4),
    (2, 1, 16, 128, 128, 4),        # a single key
    # d = 256 / 8 heads (BASELINE configs[3]): keys projected, flash attention per head
    (3, 200, 32, 256, 256, 8),      # ISAB mab0, ragged N
    (4, 130, 1, 256, 256, 8),       # PMA: one seed, 8 score rows, reassociated (X read once)
    (2, 515, 2, 256, 256, 8),       # two seeds: all 16 score rows live, ragged N
    (2, 300, 16, 256, 256, 8),      # 16 queries: the 16x16x16 products of the backward
    (2, 77, 32, 3, 256, 8),         # layer 1: reassociated fp32 kernels, 256 score rows
    (2, 1, 32, 256, 256, 8),        # a single key
]


@pytest.mark.parametrize("case", MAB0_CASES, ids=[str(c) for c in MAB0_CASES])
def test_mab0_fwd_bf16(dev, case):
    import modules
    import pca_hip
    from oracle import st_oracle as orc
    B, N, m, dk, d, h = case
    p = _mab_params(d, dk, d, seed=sum(case) + 7)
    g = torch.Generator().manual_seed(3 + sum(case))
    I = torch.randn(1, m, d, generator=g) * 0.5
    X = torch.randn(B, N, dk, generator=g)
    if dk <= 4:
        X[..., -1] = X[..., -1] * 3 - 9
    ref = orc.mab_forward(I.expand(B, -1, -1), X, p, h)
    mab = modules.MAB(d, dk, d, h).to(dev)
    mab.load_state_dict(p)
    pca_hip.set_mode("bf16")
    with torch.no_grad():
        Y = mab(I.to(dev), X.to(dev), q_shared=True)
    pca_hip.set_mode("f32")
    # (layer 1, dk <= 4: scores, softmax and fc_v are exact fp32 arithmetic on the points; fc_o
    #  at d = 256 runs on the MFMA with hi + lo bf16 operand pairs: fp32-level)
    err = close(Y, ref, FWD_TOL if dk > 4 else 2e-4, f"mab0 fwd {case}")
    print(f"mab0 fwd {case}: max err {err:.3e} (max|ref| {float(ref.abs().max()):.2f})")


@pytest.mark.parametrize("case", MAB0_CASES, ids=[str(c) for c in MAB0_CASES])
def test_mab0_bwd_bf16(dev, case):
    import modules
    import pca_hip
    from oracle import st_oracle as orc
    B, N, m, dk, d, h = case
    p = _mab_params(d, dk, d, seed=sum(case) + 7)
    g = torch.Generator().manual_seed(3 + sum(case))
    I = torch.randn(1, m, d, generator=g) * 0.5
    X = torch.randn(B, N, dk, generator=g)
    if dk <= 4:
        X[..., -1] = X[..., -1] * 3 - 9
    G = torch.randn(B, m, d, generator=g)
    # reference 1: exact fp32 oracle (explicit adjoint); dQ summed over the shared query
    exact = orc.mab_backward(G, I.expand(B, -1, -1).contiguous(), X, p, h)
    exact["dQ"] = exact["dQ"].sum(0, keepdim=True)
    # reference 2: autograd of the bf16-operand emulation
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    Ie, Xe = I.clone().requires_grad_(True), X.clone().requires_grad_(True)
    He = mab0_forward_bf16emu(Ie, Xe, leaves, h)
    (He * G).sum().backward()
    emu = {k: v.grad for k, v in leaves.items()}
    emu["dQ"], emu["dK"] = Ie.grad, Xe.grad
    emu["fc_k.bias"] = torch.zeros(d)
    if N == 1:      # one key: P = 1 whatever the scores, so nothing flows into the key projection
        emu["fc_k.weight"] = torch.zeros_like(emu["fc_k.weight"])

    mab = modules.MAB(d, dk, d, h).to(dev)
    mab.load_state_dict(p)
    Id = I.to(dev).requires_grad_(True)
    Xd = X.to(dev).requires_grad_(dk > 4)
    pca_hip.set_mode("bf16")
    H = mab(Id, Xd, q_shared=True)
    (H * G.to(dev)).sum().backward()
    pca_hip.set_mode("f32")
    close(H, He, 2e-3, "H vs emulation")
    got = {"dQ": Id.grad}
    if dk > 4:
        got["dK"] = Xd.grad
    for k, prm in mab.named_parameters():
        got[k] = prm.grad if prm.grad is not None else torch.zeros_like(prm)
    errs = {}
    for k, v in got.items():
        if k == "fc_k.bias":
            assert float(v.abs().max()) == 0.0           # exactly zero by construction
            continue
        # (d = 256: the online softmax feeds un-normalised probabilities to the MFMA, the emulation
        #  normalised ones - a few more ReLU pre-activations of the 96 epilogue rows change sign)
        errs[k] = close_robust(v, emu[k], 1.5e-2, k,
                               outlier_frac=2e-4 if d == 128 else max(3e-3, 2.5 / v.numel()))
        sc = max(1.0, float(exact[k].abs().max()))
        rms = float((emu[k] - exact[k]).pow(2).mean().sqrt()) / sc
        assert rms < 3e-2, (k, rms)
    print(f"mab0 bwd {case}: " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("layout", ["nt", "nn", "tn", "tt", "scalar", "scalar_tn", "strided"])
def test_gemm_bf16_tile_variants(dev, layout):
    """pca_gemm_bf16 over every tile shape (128x128, 256x32, 256x64, 32x256, 64x256, 32x32) and
    staging mode (k-vectors, row-vectors, scalar), with ragged edges, bias, alpha, accumulate,
    batches with head-slice strides and split-K, against an fp64 product of the bf16-rounded
    operands (so only the accumulation order differs)."""
    import ctypes as C
    from pca_hip import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    shapes = [(300, 32, 32), (517, 20, 64), (260, 64, 40), (32, 300, 32), (17, 1030, 96),
              (64, 515, 32), (32, 32, 4096), (9, 30, 777), (130, 140, 70), (256, 256, 256),
              (1, 1, 1)]
    if layout in ("nt", "tn"):      # the XCD-aware workgroup orders: tall A; long split K
        shapes += [(8203, 200, 40), (130, 140, 200000)]
    for (M, N, K) in shapes:
        pad = 1 if layout.startswith("scalar") else 0   # odd leading dimensions -> 4-byte loads
        # A as [M, K] (k contiguous) or [K, M] (rows contiguous); same for B as [N, K] / [K, N]
        a_t = layout[0] == "t" or layout == "scalar_tn"
        b_t = layout in ("nn", "tn", "scalar_tn")
        Ms, Ks, Ns = M + (-M) % 4, K + (-K) % 4, N + (-N) % 4      # 16-byte aligned rows
        if layout == "strided":                      # neither stride is 1 (every other float)
            if M * K > 4_000_000:
                continue
            Abuf = torch.randn(M, K, 2, generator=g).to(dev); sa_m, sa_k = 2 * K, 2
            Aref = Abuf[:, :, 0]
        elif a_t:
            Abuf = torch.randn(K, Ms + pad, generator=g).to(dev); sa_m, sa_k = 1, Ms + pad
            Aref = Abuf[:, :M].t()
        else:
            Abuf = torch.randn(M, Ks + pad, generator=g).to(dev); sa_m, sa_k = Ks + pad, 1
            Aref = Abuf[:, :K]
        if layout == "strided":
            Bbuf = torch.randn(N, K, 2, generator=g).to(dev); sb_n, sb_k = 2 * K, 2
            Bref = Bbuf[:, :, 0]
        elif b_t:
            Bbuf = torch.randn(K, Ns + pad, generator=g).to(dev); sb_n, sb_k = 1, Ns + pad
            Bref = Bbuf[:, :N].t()
        else:
            Bbuf = torch.randn(N, Ks + pad, generator=g).to(dev); sb_n, sb_k = Ks + pad, 1
            Bref = Bbuf[:, :K]
        bias = torch.randn(N, generator=g).to(dev)
        ldc = Ns + pad
        C0 = torch.randn(M, ldc, generator=g).to(dev)
        ref = 0.5 * (_bf16_round(Aref).cpu() @ _bf16_round(Bref).cpu().t()) + bias.double().cpu()
        scale = float(ref.abs().max()) + 1e-6
        for acc, split in ((0, 1), (1, 1), (1, 0), (1, 4)):
            Cout = C0.clone()
            d = _lib.GemmDesc(M, N, K, sa_m, sa_k, sb_k, sb_n, ldc, 1, 1, 0, 0, 0, 0, 0, 0,
                              acc, split, 0.5)
            _lib.check(L.pca_gemm_bf16(C.byref(d), Abuf.data_ptr(), Bbuf.data_ptr(),
                                       bias.data_ptr(), Cout.data_ptr(), None))
            want = ref + (C0[:, :N].double().cpu() if acc else 0)
            err = float((Cout[:, :N].double().cpu() - want).abs().max()) / scale
            assert err < 2e-5, f"{layout} {M}x{N}x{K} acc={acc} split={split}: {err:.2e}"
            # nothing written outside the [M, N] block
            assert torch.equal(Cout[:, N:], C0[:, N:]), f"{layout} {M}x{N}x{K}: wrote past N"
    # batched head slices, as the attention of the exact chain issues them:
    #   S[b, h] = Q[b, :, h*dh:(h+1)*dh] . K[b, :, h*dh:(h+1)*dh]^T   (many rows, few columns)
    #   O[b, :, h*dh:(h+1)*dh] = S[b, h] . V[b, :, h*dh:(h+1)*dh]      (row-vector B operand)
    Bn, h, nq, nk, dh = 3, 8, 300, 32, 32
    dmodel = h * dh
    Q = torch.randn(Bn, nq, dmodel, generator=g).to(dev)
    Kx = torch.randn(Bn, nk, dmodel, generator=g).to(dev)
    S = torch.zeros(Bn, h, nq, nk, device=dev)
    d = _lib.GemmDesc(nq, nk, dh, dmodel, 1, 1, dmodel, nk, Bn, h, nq * dmodel, dh,
                      nk * dmodel, dh, h * nq * nk, nq * nk, 0, 1, 1.0)
    _lib.check(L.pca_gemm_bf16(C.byref(d), Q.data_ptr(), Kx.data_ptr(), None, S.data_ptr(), None))
    Qh = _bf16_round(Q).cpu().view(Bn, nq, h, dh).permute(0, 2, 1, 3)
    Kh = _bf16_round(Kx).cpu().view(Bn, nk, h, dh).permute(0, 2, 1, 3)
    refS = Qh @ Kh.transpose(-1, -2)
    assert float((S.double().cpu() - refS).abs().max()) < 2e-5 * float(refS.abs().max())
    O = torch.zeros(Bn, nq, dmodel, device=dev)
    d = _lib.GemmDesc(nq, dh, nk, nk, 1, dmodel, 1, dmodel, Bn, h, h * nq * nk, nq * nk,
                      nk * dmodel, dh, nq * dmodel, dh, 0, 1, 1.0)
    _lib.check(L.pca_gemm_bf16(C.byref(d), S.data_ptr(), Kx.data_ptr(), None, O.data_ptr(), None))
    refO = (_bf16_round(S).cpu() @ Kh).permute(0, 2, 1, 3).reshape(Bn, nq, dmodel)
    assert float((O.double().cpu() - refO).abs().max()) < 2e-5 * float(refO.abs().max())
    # few rows, reduction over the points (32x32 tiles, batch of heads): dV[b,h] = S^T . dO_h
    dV = torch.zeros(Bn, nk, dmodel, device=dev)
    d = _lib.GemmDesc(nk, dh, nq, 1, nk, dmodel, 1, dmodel, Bn, h, h * nq * nk, nq * nk,
                      nq * dmodel, dh, nk * dmodel, dh, 0, 1, 1.0)
    _lib.check(L.pca_gemm_bf16(C.byref(d), S.data_ptr(), Q.data_ptr(), None, dV.data_ptr(), None))
    refdV = (_bf16_round(S).cpu().transpose(-1, -2) @ Qh).permute(0, 2, 1, 3).reshape(Bn, nk, dmodel)
    assert float((dV.double().cpu() - refdV).abs().max()) < 2e-5 * float(refdV.abs().max())


@pytest.mark.parametrize("ci", [1, 3], ids=["st_shipped", "st_cfg4"])
def test_engine_bf16_generic_gemm_path(dev, golden_st, ci):
    """mode = BF16 on architectures without fused kernels (shipped d=64 / 8 heads / 64 inducing
    points; BASELINE configs[3] d=256 / 8 heads / 32 inducing points): the exact chain with
    bf16 MFMA operands (k_gemm_bf16) against the reference's golden logits and gradients."""
    import inputs as gi
    import models
    from pca_hip import _lib, trainer
    name, B, N, din, d, h, m, C, full = gi.ST_CASES[ci]
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    net.load_state_dict({k: T(v) for k, v in golden_st.sub(f"{name}/p/").items()})
    X = T(gi.pc_input(600 + ci, B, N, din), dev)
    y = T(gi.labels(700 + ci, B, C), dev)
    eng = trainer.STEngine(net, B, N, mode=_lib.MODE_BF16, training=True)
    eng.fwd_bwd(X, y, phase=-1)
    ref = golden_st[f"{name}/logits"].reshape(B, C)
    err = close(eng.logits, ref, 3e-2, "logits")
    exact = trainer.STEngine(net, B, N, mode=_lib.MODE_F32, training=True)
    exact.fwd_bwd(X, y, phase=-1)
    assert float((eng.logits - exact.logits).abs().max()) > 1e-6, "bf16 mode ran the fp32 GEMMs"
    off, worst = 0, 0.0
    for k, prm in net.named_parameters():
        gr = eng.grads[off:off + prm.numel()].view_as(prm)
        off += prm.numel()
        if full:
            worst = max(worst, close_robust(gr, golden_st[f"{name}/g/{k}"], 5e-2, k,
                                            outlier_frac=5e-3))
        else:
            worst = max(worst, close_robust(gr.reshape(-1)[::gi.GRAD_SUBSAMPLE],
                                            golden_st[f"{name}/gsub/{k}"], 5e-2, k,
                                            outlier_frac=5e-3))
    print(f"{name} bf16 generic path: logits err {err:.2e}, worst grad err {worst:.2e}")
    inf = trainer.STEngine(net, B, N, mode=_lib.MODE_BF16, training=False)
    close(inf.forward(X), ref, 3e-2, "logits(inference)")


def test_engine_bf16_generic_path_large_rows(dev):
    """The GEMM chain at a size where its large-problem branches run (XCD-aware workgroup order,
    transposed-weight dX, float4 column sums): configs[3] architecture, 16 sets of 4096 points,
    bf16 chain against the exact fp32 chain on the same weights and inputs."""
    import inputs as gi
    import models
    from pca_hip import _lib, trainer
    B, N, din, d, h, m, C = 16, 4096, 3, 256, 8, 32, 50
    torch.manual_seed(5)
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    X = T(gi.pc_input(811, B, N, din), dev)
    y = T(gi.labels(812, B, C), dev)
    exact = trainer.STEngine(net, B, N, mode=_lib.MODE_F32, training=True)
    exact.fwd_bwd(X, y, phase=-1)
    eng = trainer.STEngine(net, B, N, mode=_lib.MODE_BF16, training=True)
    eng.fwd_bwd(X, y, phase=-1)
    err = close(eng.logits, exact.logits, 3e-2, "logits")
    assert float((eng.logits - exact.logits).abs().max()) > 1e-6, "bf16 mode ran the fp32 GEMMs"
    off, worst = 0, 0.0
    for k, prm in net.named_parameters():
        n = prm.numel()
        worst = max(worst, close_robust(eng.grads[off:off + n], exact.grads[off:off + n], 5e-2, k,
                                        outlier_frac=5e-3))
        off += n
    print(f"configs[3] architecture, B=16 N=4096: logits err {err:.2e}, worst grad err {worst:.2e}")
    inf = trainer.STEngine(net, B, N, mode=_lib.MODE_BF16, training=False)    # fused mab1 forward
    close(inf.forward(X), exact.logits, 3e-2, "logits(inference)")


@pytest.mark.parametrize("case", [(2, 300, 32, 256, 256, 8), (2, 77, 32, 3, 256, 8)],
                         ids=["d256", "d256_layer1"])
def test_mab1_fwd_d256_bf16_activations(dev, case):
    """The d = 256 forward pair with bf16 activations crossing the ABI (q_dtype = y_dtype =
    PCA_BF16; layer 1 takes its fp32 points and writes bf16), called through the C entry."""
    import ctypes as C
    from oracle import st_oracle as orc
    from pca_hip import _lib
    L = _lib.lib()
    B, N, m, dq, d, h = case
    p = _mab_params(dq, d, d, seed=sum(case) + 1)
    g = torch.Generator().manual_seed(3 + sum(case))
    X = torch.randn(B, N, dq, generator=g)
    if dq <= 4:
        X[..., -1] = X[..., -1] * 3 - 9
    else:
        X = X.to(torch.bfloat16).float()                 # what the bf16 tensor holds
    H = torch.randn(B, m, d, generator=g)
    ref = orc.mab_forward(X, H, p, h)
    qdt = _lib.PCA_F32 if dq <= 4 else _lib.PCA_BF16
    s = _lib.MabShape(B, N, m, dq, d, d, h, 0, _lib.MODE_BF16, qdt, _lib.PCA_F32, _lib.PCA_BF16,
                      None, 0)
    nws = L.pca_mab_fwd_ws_bytes(C.byref(s))
    assert nws > 0, L.pca_last_error()
    assert L.pca_mab_saved_bytes(C.byref(s)) > 0           # the shape trains on fused kernels
    Xd = X.to(dev) if dq <= 4 else X.to(dev).to(torch.bfloat16)
    Hd = H.to(dev)
    prm = [p[k].to(dev).contiguous() for k in ("fc_q.weight", "fc_q.bias", "fc_k.weight",
                                               "fc_k.bias", "fc_v.weight", "fc_v.bias",
                                               "fc_o.weight", "fc_o.bias")]
    pp = _lib.MabParams(*[t.data_ptr() for t in prm], None, None, None, None)
    Y = torch.empty(B, N, d, dtype=torch.bfloat16, device=dev)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    _lib.check(L.pca_mab_fwd(C.byref(s), Xd.data_ptr(), Hd.data_ptr(), C.byref(pp), Y.data_ptr(),
                             None, ws.data_ptr(), None))
    torch.cuda.synchronize()
    err = close(Y.float(), ref, FWD_TOL, f"mab1 fwd bf16 activations {case}")
    print(f"mab1 fwd {case} bf16 in/out: max err {err:.3e}")


def test_auto_mode_at_d256(dev):
    """'auto' probes the library per block: the d = 256 / 8-head / m = 32 block has fused forward
    AND backward kernels, so it trains in bf16 mode; a shape without any (d = 256 with m = 20
    inducing points) resolves to the exact path under 'auto' and refuses under 'bf16'."""
    import modules
    import pca_hip
    B, N, m, d, h = 2, 130, 32, 256, 8
    p = _mab_params(d, d, d, seed=77)
    g = torch.Generator().manual_seed(78)
    X, H = torch.randn(B, N, d, generator=g).to(dev), torch.randn(B, m, d, generator=g).to(dev)
    mab = modules.MAB(d, d, d, h).to(dev)
    mab.load_state_dict(p)
    try:
        pca_hip.set_mode("f32")
        with torch.no_grad():
            exact = mab(X, H)
        pca_hip.set_mode("auto")
        with torch.no_grad():
            fused = mab(X, H)
        diff = float((fused - exact).abs().max())
        assert 1e-6 < diff < FWD_TOL * max(1.0, float(exact.abs().max())), diff
        Y = mab(X.clone().requires_grad_(True), H)           # training: fused as well
        assert 1e-6 < float((Y.detach() - exact).abs().max()) < FWD_TOL * max(1.0, float(exact.abs().max()))
        Y.sum().backward()
        H20 = H[:, :20].contiguous()                          # 20 keys: no fused kernel
        with torch.no_grad():
            pca_hip.set_mode("f32")
            e20 = mab(X, H20)
            pca_hip.set_mode("auto")
            assert float((mab(X, H20) - e20).abs().max()) < 1e-5
            pca_hip.set_mode("bf16")
            with pytest.raises(pca_hip.PcaHipError):
                mab(X, H20)
    finally:
        pca_hip.set_mode("f32")


# ---- PCA_MODE_FP8: fp8 e4m3 operands in the d x d projections of the forward ---------------------
def test_mab0_fwd_fp8_projected_keys(dev):
    """Few-queries block at d = 256 in PCA_MODE_FP8: fc_k / fc_v over the N keys with fp8 e4m3
    operands (k_rowstream PROJ2, F8).  Kernel vs the fp8-operand emulation (tight) and emulation vs
    the bf16-operand emulation (the precision statement)."""
    import modules
    import pca_hip
    B, N, m, dk, d, h = 3, 200, 32, 256, 256, 8
    p = _mab_params(d, dk, d, seed=77)
    g = torch.Generator().manual_seed(78)
    I = torch.randn(1, m, d, generator=g) * 0.5
    X = torch.randn(B, N, dk, generator=g)
    e8 = mab0_forward_bf16emu(I, X, p, h, fp8=True)
    eb = mab0_forward_bf16emu(I, X, p, h)
    mab = modules.MAB(d, dk, d, h).to(dev)
    mab.load_state_dict(p)
    pca_hip.set_mode("fp8")
    try:
        with torch.no_grad():
            H8 = mab(I.to(dev), X.to(dev), q_shared=True)
    finally:
        pca_hip.set_mode("f32")
    e = close(H8, e8, 6e-3, "fp8 kernel vs fp8 emulation")
    sc = max(1.0, float(eb.abs().max()))
    rms = float((e8 - eb).pow(2).mean().sqrt()) / sc
    print(f"mab0 fwd fp8: kernel vs emulation {e:.2e}; fp8 vs bf16 emulation rms {rms:.2e}")
    assert 1e-5 < rms < 4e-2          # (and not the bf16 kernels under another name)



FP8_CASES = [      # B, N, m, dq, d, h
    (3, 200, 16, 128, 128, 4),
    (2, 77, 16, 2, 128, 4),           # layer 1: only fc_o is a d x d projection
    (2, 300, 32, 256, 256, 8),
    (2, 77, 32, 3, 256, 8),           # layer 1 at d = 256: single-launch kernel, fc_o in fp8
]


@pytest.mark.parametrize("case", FP8_CASES, ids=[str(c) for c in FP8_CASES])
def test_mab1_fwd_fp8(dev, case):
    """Kernel vs the oracle-side fp8 emulation (same rounding points: e4m3 of s*W and of the
    activations, fp32 accumulation) - tight; emulation vs the exact oracle - the precision
    statement of the mode (e4m3 keeps 4 significant bits)."""
    import ctypes as C
    from oracle import st_oracle as orc
    from pca_hip import _lib
    L = _lib.lib()
    B, N, m, dq, d, h = case
    p = _mab_params(dq, d, d, seed=sum(case) + 5)
    g = torch.Generator().manual_seed(9 + sum(case))
    X = torch.randn(B, N, dq, generator=g)
    if dq <= 4:
        X[..., -1] = X[..., -1] * 3 - 9
    H = torch.randn(B, m, d, generator=g)
    abf = d == 256                              # d = 256: bf16 activations at the ABI
    if abf and dq > 4:
        X = X.to(torch.bfloat16).float()
    emu = orc.mab1_forward_fp8emu(X, H, p, h)
    exact = orc.mab_forward(X, H, p, h)
    qdt = _lib.PCA_BF16 if (abf and dq > 4) else _lib.PCA_F32
    ydt = _lib.PCA_BF16 if abf else _lib.PCA_F32
    s = _lib.MabShape(B, N, m, dq, d, d, h, 0, _lib.MODE_FP8, qdt, _lib.PCA_F32, ydt, None, 0)
    nws = L.pca_mab_fwd_ws_bytes(C.byref(s))
    assert nws > 0, L.pca_last_error()
    Xd = X.to(dev).to(torch.bfloat16) if qdt == _lib.PCA_BF16 else X.to(dev)
    prm = [p[k].to(dev).contiguous() for k in ("fc_q.weight", "fc_q.bias", "fc_k.weight",
                                               "fc_k.bias", "fc_v.weight", "fc_v.bias",
                                               "fc_o.weight", "fc_o.bias")]
    pp = _lib.MabParams(*[t.data_ptr() for t in prm], None, None, None, None)
    Y = torch.empty(B, N, d, dtype=torch.bfloat16 if abf else torch.float32, device=dev)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    Hd = H.to(dev)
    _lib.check(L.pca_mab_fwd(C.byref(s), Xd.data_ptr(), Hd.data_ptr(), C.byref(pp), Y.data_ptr(),
                             None, ws.data_ptr(), None))
    torch.cuda.synchronize()
    e1 = close(Y.float(), emu, 1.5e-2, f"fp8 kernel vs fp8 emulation {case}")
    sc = max(1.0, float(exact.abs().max()))
    rms = float((emu - exact).pow(2).mean().sqrt()) / sc
    print(f"mab1 fwd fp8 {case}: kernel vs emulation {e1:.2e}; emulation vs exact rms {rms:.2e}")
    assert rms < 4e-2
    # not the bf16 kernels under another name
    s.mode = _lib.MODE_BF16
    Yb = torch.empty_like(Y)
    _lib.check(L.pca_mab_fwd(C.byref(s), Xd.data_ptr(), Hd.data_ptr(), C.byref(pp), Yb.data_ptr(),
                             None, ws.data_ptr(), None))
    assert float((Yb.float() - Y.float()).abs().max()) > 1e-3
