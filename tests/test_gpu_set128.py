"""The set-resident forward of the d = 128 / 4 heads / m = 16 train step (csrc/set128_fwd.hip: a pair of
workgroups carries a set through ISAB -> ISAB -> PMA attention, activations resident in LDS, the
few-queries attention partials handed across by write-through stores + a flag) against

* the per-block launches it replaces (``PCA_SET128=0``): same arithmetic on the same operand
  roundings, so logits, loss and all 45 gradients agree to reduction-order rounding of the few fp32
  merges that run in another order (layer-1 attention, partial merges) - a much tighter check of the
  kernel's indexing than the oracle tolerance of the bf16 mode allows;
* the CPU oracle (``oracle/st_oracle.py:st_grads``: the reference's ``ST`` + autograd restated,
  Code/models.py:34-44, modules.py:19-33), at the bf16 mode's tolerance.

Sizes: both N the kernel accepts (256, 512: one or two 32-point units per quad of waves, 2 or 4 PMA
partials per set), din 2 and 3, batch sizes that are not a multiple of 8 (idle workgroup pairs) up to the
128 sets whose 256 workgroups fill the chip; B = 130 (260 workgroups would not all be resident: the
engine keeps the per-block launches) checks the fall-back."""
import os

import numpy as np
import pytest
import torch

from util import T, close, close_robust

import inputs as gi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import pca_hip
    pca_hip.lib()
    return torch.device("cuda", 0)


def _run(dev, net, X, y, B, N, set128, head=True):
    """head: the PMA epilogue / classifier / loss stages in the set-resident launch's tail (the default)
    or as the launch of their own (k_pma_head1, ``PCA_SET128_HEAD=0``)."""
    from pca_hip import _lib, trainer
    old = {k: os.environ.get(k) for k in ("PCA_SET128", "PCA_SET128_HEAD")}
    os.environ["PCA_SET128"] = "1" if set128 else "0"
    os.environ["PCA_SET128_HEAD"] = "1" if head else "0"
    try:
        eng = trainer.STEngine(net, B, N, _lib.MODE_BF16, training=True)
        eng.fwd_bwd(X, y, phase=-1)
        torch.cuda.synchronize()
        eng.check_handoffs()           # no bounded spin-wait of the pair hand-offs expired
        return eng.logits.clone(), float(eng.loss), eng.grads.clone()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("head", [True, False], ids=["head-fused", "head-launch"])
@pytest.mark.parametrize("B,N,din", [(5, 256, 2), (8, 256, 3), (3, 512, 2), (16, 512, 2), (7, 512, 3),
                                     (128, 512, 2), (130, 512, 2)])
def test_set128_forward_equals_per_block_launches(dev, B, N, din, head):
    import models
    d, h, m, C = 128, 4, 16, 50
    torch.manual_seed(100 + N + din)
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    X = T(gi.pc_input(7000 + N, B, N, din), dev)
    y = T(gi.labels(7001 + N, B, C), dev)
    lg0, loss0, g0 = _run(dev, net, X, y, B, N, set128=False)
    lg1, loss1, g1 = _run(dev, net, X, y, B, N, set128=True, head=head)
    assert torch.isfinite(lg1).all() and torch.isfinite(g1).all()
    # bf16 activations: a one-ulp difference of a merged fp32 statistic can flip the rounding of a few
    # hidden activations; everything else is the same arithmetic
    e = close(lg1, lg0.cpu(), 4e-3, "logits")
    assert abs(loss1 - loss0) < 2e-3 * max(1.0, abs(loss0)), (loss1, loss0)
    off, worst = 0, 0.0
    for k, prm in net.named_parameters():
        a = g1[off:off + prm.numel()].view_as(prm)
        bref = g0[off:off + prm.numel()].view_as(prm).cpu()
        off += prm.numel()
        worst = max(worst, close_robust(a, bref, 6e-3, k, outlier_frac=1e-3))
    print(f"B={B} N={N} din={din}: logits {e:.2e}, worst grad {worst:.2e}")


@pytest.mark.parametrize("B,N,din", [(6, 256, 2), (12, 512, 3)])
def test_set128_train_step_vs_oracle(dev, B, N, din):
    import models
    from oracle import st_oracle as orc
    d, h, m, C = 128, 4, 16, 50
    torch.manual_seed(200 + N)
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    p = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    Xn = gi.pc_input(7100 + N, B, N, din)
    yn = gi.labels(7101 + N, B, C)
    ref_loss, ref_lg, ref_g = orc.st_grads(torch.from_numpy(Xn), torch.from_numpy(yn), p, h)
    lg, loss, g = _run(dev, net, T(Xn, dev), T(yn, dev), B, N, set128=True)
    close(lg, ref_lg.reshape(B, C), 3e-2, "logits")
    assert abs(loss - ref_loss) < 3e-2 * max(1.0, abs(ref_loss))
    off = 0
    for k, prm in net.named_parameters():
        close_robust(g[off:off + prm.numel()].view_as(prm), ref_g[k], 5e-2, k, outlier_frac=5e-3)
        off += prm.numel()


def test_handoff_timeouts_are_surfaced(dev):
    """The pair hand-offs poll with a bounded spin; an expiry is counted in a word of the workspace
    (``pca_st_handoff_counter``) that the library never clears - not by the per-step flag reset either -
    and ``STEngine.check_handoffs`` / ``Trainer.read_stats`` raise on it.  A shape without the
    set-resident launch has no such word."""
    import models
    from pca_hip import _lib, trainer
    torch.manual_seed(5)
    net = models.ST(dim_input=2, num_outputs=1, dim_output=7, num_inds=16, dim_hidden=128,
                    num_heads=4).to(dev)
    B, N = 6, 512
    eng = trainer.STEngine(net, B, N, _lib.MODE_BF16, training=True)
    assert eng._handoff_word is not None
    X, y = T(gi.pc_input(31, B, N, 2), dev), T(gi.labels(32, B, 7), dev)
    for _ in range(3):                 # the flag block is reset every step, the counter is not
        eng.fwd_bwd(X, y, phase=-1)
    torch.cuda.synchronize()
    eng.check_handoffs()
    assert int(eng._handoff_word.item()) == 0
    eng._handoff_word.fill_(2)         # what two expired waits would leave
    eng.fwd_bwd(X, y, phase=-1)
    torch.cuda.synchronize()
    assert int(eng._handoff_word.item()) == 2          # survived the step's flag reset
    with pytest.raises(_lib.PcaHipError, match="hand-off"):
        eng.check_handoffs()
    other = trainer.STEngine(net, B, 300, _lib.MODE_BF16, training=True)   # N = 300: per-block launches
    assert other._handoff_word is None
