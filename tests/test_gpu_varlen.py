"""Padded batches of variable-size sets (the "padded variable-N
point sets" of the north star).  Not in the reference, whose batches are dense, so the pin is
the property the design promises: MASK == TRUNCATION.  The logits, the loss and every
parameter gradient of a padded batch with lengths L must equal what the same model gives on
the un-padded sets one by one - computed here by the CPU oracle (itself pinned by the
reference's golden vectors) and by the HIP path's own dense entry points."""
import numpy as np
import pytest
import torch

import inputs as gi
from util import T, close, close_robust

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def _oracle_truncated(net, X, y, lengths, h):
    """logits [B, C], mean CE loss and gradients from the oracle on X[b, :lengths[b]]."""
    from oracle import st_oracle as orc
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    lg = [orc.st_forward(torch.from_numpy(X[b:b + 1, :lengths[b]]), params, h).reshape(1, -1)
          for b in range(X.shape[0])]
    lg = torch.cat(lg, 0)
    loss = orc.cross_entropy(lg, torch.from_numpy(y))
    loss.backward()
    return lg.detach().numpy(), float(loss), {k: v.grad.numpy() for k, v in params.items()}


CASES = [  # name, din, d, h, m, C, N, lengths
    ("tiny_f32", 2, 16, 4, 4, 10, 9, [9, 1, 5, 7]),
    ("shipped_f32", 3, 64, 8, 64, 10, 60, [60, 33, 1]),
    ("cfg1", 2, 128, 4, 16, 50, 300, [300, 129, 128, 1, 257, 64]),
    ("cfg3", 3, 128, 4, 16, 50, 640, [640, 500, 131, 17]),
    # configs[4] architecture (d = 256, 8 heads, 32 inducing points): the fused d = 256 kernels with
    # lengths - range splits, ragged last tiles, a set shorter than one tile, a single point
    ("cfg5", 3, 256, 8, 32, 10, 300, [300, 129, 31, 1, 257]),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_mask_equals_truncation_engine(dev, case, mode):
    import models
    from pca_hip import _lib, trainer
    name, din, d, h, m, C, N, lengths = case
    if mode == "bf16" and d not in (128, 256):
        pytest.skip("no fused bf16 kernels for this architecture (runs the fp32 path)")
    B = len(lengths)
    torch.manual_seed(5)
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    X = gi.pc_input(4100, B, N, din)
    for b, L in enumerate(lengths):            # padding rows: zeros, as the pack kernel writes
        X[b, L:] = 0.0
    y = gi.labels(4200, B, C)
    ref_lg, ref_loss, ref_g = _oracle_truncated(net, X, y, lengths, h)
    md = _lib.MODE_F32 if mode == "f32" else _lib.MODE_BF16
    ld = torch.tensor(lengths, dtype=torch.int32, device=dev)
    inf = trainer.STEngine(net, B, N, md, training=False)
    tol_f, tol_b = (2e-4, 2e-3) if mode == "f32" else (3e-2, None)
    close(inf.forward(T(X, dev), ld), ref_lg, tol_f, "logits(inf)")
    eng = trainer.STEngine(net, B, N, md, training=True)
    eng.fwd_bwd(T(X, dev), T(y, dev), lengths=ld)
    close(eng.logits, ref_lg, tol_f, "logits(train)")
    assert abs(float(eng.loss) - ref_loss) < (2e-4 if mode == "f32" else 3e-2)
    off = 0
    for k, p in net.named_parameters():
        g = eng.grads[off:off + p.numel()].view_as(p)
        off += p.numel()
        if mode == "f32":
            close(g, ref_g[k], tol_b, k)
        else:
            close_robust(g, ref_g[k], 5e-2, k, outlier_frac=2e-3)
    # garbage in the padding must not matter as long as it is finite
    X2 = X.copy()
    for b, L in enumerate(lengths):
        X2[b, L:] = 7.5
    lg2 = inf.forward(T(X2, dev), ld).clone()
    close(lg2, ref_lg, tol_f, "logits with non-zero padding")
    # the padded batch through the dense entry point differs (the test is not vacuous)
    if min(lengths) < N:
        dense = inf.forward(T(X, dev)).cpu().numpy()
        assert np.abs(dense - ref_lg).max() > 1e-3


def _emu_truncated(net, X, y, lengths, h, fp8):
    """logits, mean CE loss and gradients of the operand-rounding emulation (tests/emu.py) on the
    un-padded sets X[b, :lengths[b]]."""
    from emu import st_forward_emu
    from oracle import st_oracle as orc
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    lg = torch.cat([st_forward_emu(torch.from_numpy(X[b:b + 1, :lengths[b]]), params, h, fp8=fp8)
                    for b in range(X.shape[0])], 0)
    loss = orc.cross_entropy(lg, torch.from_numpy(y))
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
             for k, v in params.items()}
    return lg.detach().numpy(), float(loss), grads


@pytest.mark.parametrize("mode", ["bf16", "fp8"])
def test_cfg5_train_step_vs_emulation(dev, mode):
    """BASELINE configs[4] (the configs[3] architecture on padded variable-size sets) as a TRAINING
    step in PCA_MODE_FP8 (and PCA_MODE_BF16 beside it): logits, loss and all 45 gradients of
    STEngine(..., training=True) with lengths against (i) the emulation of the mode's forward
    (e4m3 operands in fc_o of the many-queries blocks and fc_k / fc_v of the d -> d few-queries
    block, bf16 elsewhere) differentiated with the mode's straight-through bf16 backward, and
    (ii) the exact fp32 oracle on the truncated sets at the bf16 tolerances of
    test_mask_equals_truncation_engine.  The fp8 forward must also differ from the bf16 one."""
    import models
    from pca_hip import _lib, trainer
    name, din, d, h, m, C, N, lengths = CASES[-1]
    assert name == "cfg5"
    B = len(lengths)
    torch.manual_seed(5)
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    X = gi.pc_input(4100, B, N, din)
    for b, L in enumerate(lengths):
        X[b, L:] = 0.0
    y = gi.labels(4200, B, C)
    fp8 = mode == "fp8"
    emu_lg, emu_loss, emu_g = _emu_truncated(net, X, y, lengths, h, fp8)
    ref_lg, ref_loss, ref_g = _oracle_truncated(net, X, y, lengths, h)
    md = _lib.MODE_FP8 if fp8 else _lib.MODE_BF16
    ld = torch.tensor(lengths, dtype=torch.int32, device=dev)
    eng = trainer.STEngine(net, B, N, md, training=True)
    eng.fwd_bwd(T(X, dev), T(y, dev), lengths=ld)
    torch.cuda.synchronize()
    e1 = close(eng.logits, emu_lg, 1.5e-2, "logits vs emulation")
    close(eng.logits, ref_lg, 3e-2 if not fp8 else 6e-2, "logits vs exact oracle")
    assert abs(float(eng.loss) - emu_loss) < 1.5e-2, (float(eng.loss), emu_loss)
    assert abs(float(eng.loss) - ref_loss) < (3e-2 if not fp8 else 6e-2)
    off, worst = 0, 0.0
    for k, p in net.named_parameters():
        g = eng.grads[off:off + p.numel()].view_as(p)
        off += p.numel()
        worst = max(worst, close_robust(g, emu_g[k], 3e-2, k + " vs emulation", outlier_frac=3e-3))
        close_robust(g, ref_g[k], 5e-2 if not fp8 else 1e-1, k + " vs exact oracle",
                     outlier_frac=2e-3 if not fp8 else 1e-2)
    assert off == eng.grads.numel()
    print(f"cfg5 {mode} train step: logits vs emulation {e1:.2e}, worst grad vs emulation {worst:.2e}")
    if fp8:       # not the bf16 kernels under another name
        engb = trainer.STEngine(net, B, N, _lib.MODE_BF16, training=True)
        engb.fwd_bwd(T(X, dev), T(y, dev), lengths=ld)
        assert float((engb.logits - eng.logits).abs().max()) > 1e-4


def test_mask_equals_truncation_modules(dev):
    """nn.Module path (autograd through pca_mab_fwd/bwd with k_lengths): ST(X, lengths) against
    per-set dense calls of the same module, logits and gradients."""
    import models
    torch.manual_seed(3)
    net = models.ST(dim_input=2, num_outputs=1, dim_output=7, num_inds=8, dim_hidden=32,
                    num_heads=4).to(dev)
    lengths = [40, 13, 1, 27]
    B, N = len(lengths), 40
    X = gi.pc_input(77, B, N, 2)
    for b, L in enumerate(lengths):
        X[b, L:] = 0.0
    Xd = T(X, dev)
    w = T(gi.randn(78, B, 7), dev)
    lg = net(Xd, torch.tensor(lengths))
    (lg * w).sum().backward()
    got = {k: p.grad.clone() for k, p in net.named_parameters()}
    net.zero_grad()
    ref = []
    for b, L in enumerate(lengths):
        ref.append(net(Xd[b:b + 1, :L]).reshape(1, -1))
    ref = torch.cat(ref, 0)
    (ref * w).sum().backward()
    close(lg, ref, 2e-5, "logits")
    for k, p in net.named_parameters():
        close(got[k], p.grad, 2e-4, k)
    with torch.no_grad():
        close(net(Xd, torch.tensor(lengths)), ref, 2e-5, "logits(no_grad)")


def test_variable_length_dataset_and_trainer(dev):
    """ESC_pc_temp(nt_valid=...): padded batches + lengths from one pack launch; the Trainer
    (hipGraph and eager) consumes them; its loss equals the oracle's on the truncated sets."""
    import dataset
    import models
    from pca_hip import _lib, trainer
    from oracle import st_oracle as orc
    F, Nt, S = 32, 6, 12
    rng = np.random.Generator(np.random.PCG64(21))
    x = rng.normal(-9, 3, size=(F, Nt, S)).astype(np.float32)
    y = rng.integers(0, 5, size=S)
    ntv = rng.integers(1, Nt + 1, size=S).astype(np.int32)
    ntv[0] = Nt
    farr, tarr = np.linspace(0, 0.5, F), np.linspace(0, 0.1, Nt)
    ds = dataset.ESC_pc_temp(x, y, farr, tarr, device=dev, nt_valid=ntv)
    assert ds.variable_length
    idx = torch.arange(S, device=dev)
    pts, lab, lens = ds.batch(idx)
    assert lens.dtype == torch.int32 and lens.cpu().tolist() == (ntv * F).tolist()
    for s in range(S):
        ref = orc.pack_points_3d(x, farr, tarr, s)
        L = int(ntv[s]) * F
        np.testing.assert_array_equal(pts[s, :L].cpu().numpy(), ref[:L])
        assert (pts[s, L:] == 0).all()
        item, lbl = ds[s]
        assert tuple(item.shape) == (L, 3) and int(lbl) == int(y[s])
    torch.manual_seed(9)
    B = 4
    for graph in (False, True):
        torch.manual_seed(9)
        net = models.ST(dim_input=3, num_outputs=1, dim_output=5, num_inds=4, dim_hidden=16,
                        num_heads=4).to(dev)
        p0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        tr = trainer.Trainer(net, ds, B, mode=_lib.MODE_F32, use_graph=graph, shuffle=False)
        tr.step()
        loss = float(tr.eng.loss)
        # oracle on the first batch (sets 0..3), truncated
        lg = [orc.st_forward(torch.from_numpy(orc.pack_points_3d(x, farr, tarr, s)[None, :int(ntv[s]) * F]),
                             p0, 4).reshape(1, -1) for s in range(B)]
        ref = float(orc.cross_entropy(torch.cat(lg, 0), torch.from_numpy(y[:B].astype(np.int64))))
        assert abs(loss - ref) < 2e-5, (graph, loss, ref)
    acc, n = trainer.evaluate(net, ds, 5)
    assert n == S and 0.0 <= acc <= 1.0
