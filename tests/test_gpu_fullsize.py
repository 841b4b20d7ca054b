"""Oracle parity of the whole-model engine at the FULL sizes of BASELINE configs[1] / [2] and at
the large-row branches of configs[3]: logits, loss and all 45 parameter gradients of one
forward + backward through ``STEngine`` against ``oracle/st_oracle.py`` (autograd of the CPU
restatement, itself pinned to the reference's golden vectors by tests/test_oracle_golden.py).

Why these sizes: the fused kernels take different branches with the problem size - the point
range of a set is split over S = 8 workgroups at cfg3 (``mab0_splits``), several tiles per
workgroup in ``k_mab1_bwd`` (tpw > 1), 1024-row weight-gradient workgroups - which the small
parity cases never reach.

Tolerances: exact mode 2e-4 (logits) / 2e-3 (gradients) relative to max(1, max|ref|) (fp32
reduction order over 65 536 rows); bf16 mode 3e-2 on logits / loss and the robust criterion of
tests/util.py (rms <= 2.5e-2, <= 0.1 % of the elements beyond 5e-2) on gradients (round 4: was 0.5 %;
the fp8 / variable-size cases at the end compare with the mode's emulation at 1.5e-2 instead)."""
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


def _check(dev, B, N, din, d, h, m, C, mode, seed):
    import models
    from oracle import st_oracle as orc
    from pca_hip import _lib, trainer
    torch.manual_seed(seed)
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    p = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    X = gi.pc_input(seed + 1, B, N, din)
    y = gi.labels(seed + 2, B, C)
    torch.set_num_threads(8)
    ref_loss, ref_lg, ref_g = orc.st_grads(torch.from_numpy(X), torch.from_numpy(y), p, h)
    md = _lib.MODE_F32 if mode == "f32" else _lib.MODE_BF16
    eng = trainer.STEngine(net, B, N, md, training=True)
    eng.fwd_bwd(T(X, dev), T(y, dev), phase=-1)
    torch.cuda.synchronize()
    tol_l = 2e-4 if mode == "f32" else 3e-2
    e_lg = close(eng.logits, ref_lg.reshape(B, C), tol_l, "logits")
    assert abs(float(eng.loss) - ref_loss) < tol_l * max(1.0, abs(ref_loss)), (float(eng.loss), ref_loss)
    off, worst = 0, 0.0
    for k, prm in net.named_parameters():
        g = eng.grads[off:off + prm.numel()].view_as(prm)
        off += prm.numel()
        if mode == "f32":
            worst = max(worst, close(g, ref_g[k], 2e-3, k))
        else:
            worst = max(worst, close_robust(g, ref_g[k], 5e-2, k, outlier_frac=1e-3))
    assert off == eng.grads.numel()
    inf = trainer.STEngine(net, B, N, md, training=False)
    close(inf.forward(T(X, dev)), ref_lg.reshape(B, C), tol_l, "logits(inference)")
    print(f"B={B} N={N} din={din} d={d} m={m} {mode}: logits err {e_lg:.2e}, worst grad err {worst:.2e}")
    return eng


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_cfg3_full_size_vs_oracle(dev, mode):
    """BASELINE configs[2]: B = 32 sets of N = 2048 3-D points, d = 128, h = 4, m = 16, C = 50:
    the S = 8 point-range split of the few-queries attention and its backward merge."""
    from pca_hip import _lib
    eng = _check(dev, 32, 2048, 3, 128, 4, 16, 50, mode, seed=9100)
    if mode == "bf16":
        L = _lib.lib()
        import ctypes as C
        s = _lib.MabShape(32, 16, 2048, 128, 128, 128, 4, 1, _lib.MODE_BF16, 0, _lib.PCA_BF16, 0,
                          None, 0)
        assert L.pca_mab_saved_bytes(C.byref(s)) > 0      # the fused few-queries block serves it


def test_cfg2_full_size_bf16_vs_oracle(dev):
    """BASELINE configs[1] at its bench size in the FAST mode: B = 128 sets of N = 512 2-D points."""
    _check(dev, 128, 512, 2, 128, 4, 16, 50, "bf16", seed=9200)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_cfg4_architecture_large_rows_vs_oracle(dev, mode):
    """configs[3] architecture (d = 256, 8 heads, m = 32) at N = 4096 points per set, B = 3:
    the large-problem branches of whichever kernels serve this shape, against the ORACLE (not
    against another mode of this library)."""
    _check(dev, 3, 4096, 3, 256, 8, 32, 50, mode, seed=9300)


def test_cfg4_bench_size_bf16_vs_oracle(dev):
    """BASELINE configs[3] at its BENCH size (B = 128 sets per GPU, N = 4096 points, d = 256, 8
    heads, m = 32, C = 50) in the fast mode: 524 288 rows - the 256-workgroup slab path of
    k_wgrad256 with 2048 rows per workgroup, the deferred job hand-offs between the blocks and the
    long-job k_wgrad256_sum, none of which the B = 3 case above reaches.  The CPU oracle needs
    ~15 GB and ~1 min for this size."""
    _check(dev, 128, 4096, 3, 256, 8, 32, 50, "bf16", seed=9400)


@pytest.mark.parametrize("shape", [(128, 1025, 2), (16, 5120, 3)], ids=["fst", "3st"])
def test_shipped_shapes_bench_size_bf16_vs_oracle(dev, shape):
    """The reference's own hyper-parameters (Code/settransformer.py:81-83, Code/settransformertemp.py:95-97:
    d = 64, 8 heads of dim 8, m = 64) at the sizes ``bench.py --config fst / 3st`` times - B = 128 sets
    of N = 1025 2-D points, B = 16 sets of N = 5120 3-D points - in the fast mode, whose attention core
    is fused (csrc/attn_core.hip: no matrix A; at B = 16 the key range of the few-queries blocks and the
    query range of the many-queries blocks' dK / dV are cut into ranges and merged) and whose weight /
    bias gradients come from csrc/wgrad64.hip.  Logits, loss and all 45 gradients against the oracle."""
    B, N, din = shape
    _check(dev, B, N, din, 64, 8, 64, 10, "bf16", seed=9500 + N)


def test_d256_step_is_bit_reproducible(dev):
    """The d = 256 / 8 heads / m = 32 training step uses no fp32 atomics (weight-gradient slabs,
    per-workgroup partials + fixed-order sums everywhere): two forward + backward passes over the
    same batch give bit-identical gradients and loss.  (So does the d = 128 path: next test.)"""
    import models
    from pca_hip import _lib, trainer
    B, N, din, d, h, m, C = 16, 1000, 3, 256, 8, 32, 50
    torch.manual_seed(4)
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    X = T(gi.pc_input(5, B, N, din), dev)
    y = T(gi.labels(6, B, C), dev)
    eng = trainer.STEngine(net, B, N, _lib.MODE_BF16, training=True)
    runs = []
    for _ in range(3):
        eng.grads.zero_()
        eng.fwd_bwd(X, y, phase=-1)
        torch.cuda.synchronize()
        runs.append((eng.grads.clone(), float(eng.loss)))
    assert torch.isfinite(runs[0][0]).all()
    for g, loss in runs[1:]:
        assert torch.equal(g, runs[0][0])
        assert loss == runs[0][1]


@pytest.mark.parametrize("din", [2, 3])
def test_d128_step_is_bit_reproducible(dev, din, monkeypatch):
    """The d = 128 training step (BASELINE configs[0..2] architecture) reduces without fp32 atomics
    as well - per-workgroup partials of k_wgrad128, k_mab0_bwd, the layer-1 fc_q / fc_v gradients
    and dG, all added in a fixed order by rider rows of later launches - so three passes over the
    same batch give bit-identical gradients and loss, and the result agrees with the atomic mode
    (PCA_WGRAD_SLABS=0) to reduction-order rounding."""
    import models
    from pca_hip import _lib, trainer
    B, N, d, h, m, C = 32, 501, 128, 4, 16, 50
    torch.manual_seed(7)
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    X = T(gi.pc_input(8, B, N, din), dev)
    y = T(gi.labels(9, B, C), dev)

    def passes(n):
        eng = trainer.STEngine(net, B, N, _lib.MODE_BF16, training=True)
        out = []
        for _ in range(n):
            eng.grads.zero_()
            eng.fwd_bwd(X, y, phase=-1)
            torch.cuda.synchronize()
            out.append((eng.grads.clone(), float(eng.loss)))
        return out

    monkeypatch.setenv("PCA_WGRAD_SLABS", "0")
    ref = passes(1)[0]
    monkeypatch.delenv("PCA_WGRAD_SLABS")
    runs = passes(3)
    assert torch.isfinite(runs[0][0]).all()
    for g, loss in runs[1:]:
        assert torch.equal(g, runs[0][0])
        assert loss == runs[0][1]
    sc = max(1.0, float(ref[0].abs().max()))
    assert float((runs[0][0] - ref[0]).abs().max()) <= 2e-5 * sc
    assert abs(runs[0][1] - ref[1]) <= 1e-6 * max(1.0, abs(ref[1]))


# ---- round 4: the fp8 mode and the variable-size sets at their BENCH sizes ---------------------------
def _emu_oracle_per_set(net, X, y, lengths, h, fp8):
    """(emulation, exact oracle) of logits / loss / gradients on the un-padded sets X[b, :lengths[b]]
    (tests/emu.py: the mode's operand roundings, differentiated with its straight-through bf16
    backward; oracle/st_oracle.py: the reference restated in fp32)."""
    from emu import st_forward_emu
    from oracle import st_oracle as orc
    out = []
    for fwd in (lambda x, p: st_forward_emu(x, p, h, fp8=fp8),
                lambda x, p: orc.st_forward(x, p, h).reshape(1, -1)):
        params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}
        lg = torch.cat([fwd(torch.from_numpy(X[b:b + 1, :lengths[b]]), params)
                        for b in range(X.shape[0])], 0)
        loss = orc.cross_entropy(lg, torch.from_numpy(y))
        loss.backward()
        g = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).numpy() for k, v in params.items()}
        out.append((lg.detach().numpy(), float(loss), g))
    return out


def _bench_lengths(B, N, F=512):
    """Valid points per set as bench.py's configs[4] corpus produces them (clips of 1-4 s cut into sets
    of 8 frames of F bins: full sets, plus each clip's short last chunk of 1..8 frames), with the two
    edge cases the kernels branch on forced in: one set shorter than a 32-point tile, one odd length."""
    rng = np.random.Generator(np.random.PCG64(44))
    L = np.full(B, N, dtype=np.int64)
    short = rng.choice(B, size=B // 4, replace=False)
    L[short] = F * rng.integers(1, N // F + 1, size=short.size)
    L[short[0]] = 17
    L[short[1]] = 1000
    L[short[2]] = N
    return [int(v) for v in L]


def _check_vs_emulation(dev, B, N, mode, lengths, seed):
    import models
    from pca_hip import _lib, trainer
    din, d, h, m, C = 3, 256, 8, 32, 10
    torch.manual_seed(seed)
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    X = gi.pc_input(seed + 1, B, N, din)
    for b, L in enumerate(lengths):
        X[b, L:] = 0.0
    y = gi.labels(seed + 2, B, C)
    fp8 = mode == "fp8"
    torch.set_num_threads(16)
    (emu_lg, emu_loss, emu_g), (ref_lg, ref_loss, ref_g) = _emu_oracle_per_set(net, X, y, lengths, h, fp8)
    md = _lib.MODE_FP8 if fp8 else _lib.MODE_BF16
    ld = None if min(lengths) == N else torch.tensor(lengths, dtype=torch.int32, device=dev)
    eng = trainer.STEngine(net, B, N, md, training=True)
    eng.fwd_bwd(T(X, dev), T(y, dev), lengths=ld)
    torch.cuda.synchronize()
    e1 = close(eng.logits, emu_lg, 1.5e-2, "logits vs emulation")
    close(eng.logits, ref_lg, 3e-2 if not fp8 else 6e-2, "logits vs exact oracle")
    assert abs(float(eng.loss) - emu_loss) < 1.5e-2, (float(eng.loss), emu_loss)
    off, worst = 0, 0.0
    for k, p in net.named_parameters():
        g = eng.grads[off:off + p.numel()].view_as(p)
        off += p.numel()
        # against the emulation (same operand roundings, same ReLU masks up to rounding): 1.5e-2 with
        # at most 1e-3 of the elements beyond it - a wrong 64-row tile of a [256, 256] weight gradient
        # (1.6e-2 of its elements, off by O(1) of its scale) cannot hide in that
        worst = max(worst, close_robust(g, emu_g[k], 1.5e-2, k + " vs emulation", outlier_frac=1e-3))
        close_robust(g, ref_g[k], 5e-2 if not fp8 else 1e-1, k + " vs exact oracle",
                     outlier_frac=1e-3 if not fp8 else 1e-2)
    assert off == eng.grads.numel()
    print(f"B={B} N={N} {mode} lengths {min(lengths)}..{max(lengths)}: logits vs emulation {e1:.2e}, "
          f"worst grad vs emulation {worst:.2e}")


def test_cfg4_bench_size_fp8_vs_emulation(dev):
    """BASELINE configs[3] at its BENCH size (B = 128, N = 4096) in PCA_MODE_FP8: the multi-tile rings
    and 256-workgroup grids of k_isab1_fwd256_ab<.., F8O> and k_fq_proj_fwd<F8> (VERDICT round 3, weak
    1: checked at B = 5, N = 300 only) against the fp8 emulation of the whole model and the oracle."""
    _check_vs_emulation(dev, 128, 4096, "fp8", [4096] * 128, seed=9500)


@pytest.mark.parametrize("mode", ["bf16", "fp8"])
def test_cfg5_bench_size_varlen_vs_emulation(dev, mode):
    """BASELINE configs[4] at its BENCH size: B = 128 padded sets of up to N = 4096 points with the
    lengths bench.py's corpus produces (incl. a set shorter than one tile and a full one), bf16 and fp8,
    against the emulation on the TRUNCATED sets (mask == truncation at full size) and the oracle."""
    _check_vs_emulation(dev, 128, 4096, mode, _bench_lengths(128, 4096), seed=9600)
