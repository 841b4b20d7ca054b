"""The fused forward of the reference's SHIPPED block shape (d = 64, 8 heads of dim 8, up to 64
inducing points; csrc/sd64_fwd.hip) against the CPU oracle: every orientation (many queries,
few shared queries, PMA with one seed), layer-1 inputs (2 / 3 columns), ragged and tiny N, key
lengths.  The kernels compute in fp32, so the exact-mode tolerances apply (1e-4 relative to
max(1, max|ref|)), although they serve the "bf16" (fused) mode of the library."""
import pytest
import torch

from util import close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import pca_hip
    pca_hip.lib()
    yield torch.device("cuda", 0)
    pca_hip.set_mode("f32")


def _params(dq, dk, d, seed):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for nm, din in (("fc_q", dq), ("fc_k", dk), ("fc_v", dk), ("fc_o", d)):
        bound = 1.0 / din ** 0.5
        p[nm + ".weight"] = (torch.rand(d, din, generator=g) * 2 - 1) * bound
        p[nm + ".bias"] = (torch.rand(d, generator=g) * 2 - 1) * bound
    return p


MQ_CASES = [(3, 1025, 64, 64), (2, 5120, 64, 3), (8, 1, 64, 64), (4, 51, 17, 2), (2, 300, 1, 64)]


@pytest.mark.parametrize("case", MQ_CASES, ids=[str(c) for c in MQ_CASES])
def test_many_queries(dev, case):
    import modules
    import pca_hip
    from oracle import st_oracle as orc
    B, N, m, dq = case
    p = _params(dq, 64, 64, seed=sum(case))
    g = torch.Generator().manual_seed(1 + sum(case))
    X = torch.randn(B, N, dq, generator=g)
    if dq <= 4:
        X[..., -1] = X[..., -1] * 3 - 9
    Hk = torch.randn(B, m, 64, generator=g)
    ref = orc.mab_forward(X, Hk, p, 8)
    mab = modules.MAB(dq, 64, 64, 8).to(dev)
    mab.load_state_dict(p)
    pca_hip.set_mode("bf16")
    try:
        with torch.no_grad():
            Y = mab(X.to(dev), Hk.to(dev))
    finally:
        pca_hip.set_mode("f32")
    close(Y, ref, 1e-4, f"sd64 many queries {case}")


FQ_CASES = [(3, 1025, 64, 64), (2, 5120, 64, 3), (4, 130, 1, 64), (2, 1, 64, 64), (3, 77, 33, 2),
            (128, 200, 64, 64)]


@pytest.mark.parametrize("case", FQ_CASES, ids=[str(c) for c in FQ_CASES])
def test_few_queries(dev, case):
    import modules
    import pca_hip
    from oracle import st_oracle as orc
    B, N, m, dk = case
    p = _params(64, dk, 64, seed=sum(case) + 3)
    g = torch.Generator().manual_seed(5 + sum(case))
    I = torch.randn(1, m, 64, generator=g) * 0.5
    X = torch.randn(B, N, dk, generator=g)
    if dk <= 4:
        X[..., -1] = X[..., -1] * 3 - 9
    ref = orc.mab_forward(I.expand(B, -1, -1), X, p, 8)
    mab = modules.MAB(64, dk, 64, 8).to(dev)
    mab.load_state_dict(p)
    pca_hip.set_mode("bf16")
    try:
        with torch.no_grad():
            Y = mab(I.to(dev), X.to(dev), q_shared=True)
            close(Y, ref, 1e-4, f"sd64 few queries {case}")
            if N > 3:      # key lengths: mask == truncation
                lens = [N, 1, max(1, N // 2), N - 1][:B] + [N] * max(0, B - 4)
                Yl = mab(I.to(dev), X.to(dev), q_shared=True, key_lengths=torch.tensor(lens))
                for b, L in enumerate(lens[:4]):
                    rb = orc.mab_forward(I, X[b:b + 1, :L], p, 8)
                    close(Yl[b:b + 1], rb, 1e-4, f"sd64 few queries, length {L}")
    finally:
        pca_hip.set_mode("f32")


# ---- round 4: the TRAINING step of the shipped shapes in the fused mode: the attention core without A ----
TRAIN_CASES = [  # din, d, h, m, C, B, N, lengths
    (2, 64, 8, 64, 10, 6, 1025, None),          # FST (Code/settransformer.py:81-83)
    (3, 64, 8, 64, 10, 3, 5120, None),          # 3ST (Code/settransformertemp.py:95-97)
    (2, 64, 4, 16, 7, 5, 130, None),            # head dim 16
    (3, 32, 8, 24, 10, 4, 77, None),            # head dim 4, ragged tiles on both sides
    (2, 64, 8, 64, 10, 4, 300, [300, 129, 17, 1]),   # padded variable-size sets
]


@pytest.mark.parametrize("case", TRAIN_CASES, ids=[str(c[:7]) for c in TRAIN_CASES])
def test_shipped_shape_train_step_fused_core(dev, case):
    """PCA_MODE_BF16 on an architecture without fully fused block kernels runs the GEMM chain with bf16
    MFMA operands; for head dims <= 16 its attention core is fused since round 4 (csrc/attn_core.hip:
    scores, softmax, A V and their adjoint in three kernels, the [B h, nq, nk] matrix A never exists).
    One forward + backward of the whole model through STEngine against the CPU oracle (logits, loss,
    all 45 gradients; bf16-operand tolerances) and against the exact fp32 mode of the library."""
    import inputs as gi
    import models
    from oracle import st_oracle as orc
    from pca_hip import _lib, trainer
    from util import T, close_robust
    din, d, h, m, C, B, N, lengths = case
    torch.manual_seed(77 + N)
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h).to(dev)
    p = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    X = gi.pc_input(880 + N, B, N, din)
    y = gi.labels(881 + N, B, C)
    ld = None
    if lengths is None:
        ref_loss, ref_lg, ref_g = orc.st_grads(torch.from_numpy(X), torch.from_numpy(y), p, h)
    else:
        for b, L in enumerate(lengths):
            X[b, L:] = 0.0
        params = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        lg = torch.cat([orc.st_forward(torch.from_numpy(X[b:b + 1, :lengths[b]]), params, h).reshape(1, -1)
                        for b in range(B)], 0)
        loss = orc.cross_entropy(lg, torch.from_numpy(y))
        loss.backward()
        ref_loss, ref_lg, ref_g = float(loss), lg.detach(), {k: v.grad for k, v in params.items()}
        ld = torch.tensor(lengths, dtype=torch.int32, device=dev)
    eng = trainer.STEngine(net, B, N, _lib.MODE_BF16, training=True)
    eng.fwd_bwd(T(X, dev), T(y, dev), lengths=ld)
    torch.cuda.synchronize()
    close(eng.logits, ref_lg.reshape(B, C), 3e-2, "logits")
    assert abs(float(eng.loss) - ref_loss) < 3e-2 * max(1.0, abs(ref_loss))
    off = 0
    for k, prm in net.named_parameters():
        close_robust(eng.grads[off:off + prm.numel()].view_as(prm), ref_g[k], 5e-2, k, outlier_frac=2e-3)
        off += prm.numel()
    # the exact mode of the library on the same batch (it materialises A like the reference)
    e32 = trainer.STEngine(net, B, N, _lib.MODE_F32, training=True)
    e32.fwd_bwd(T(X, dev), T(y, dev), lengths=ld)
    torch.cuda.synchronize()
    close(eng.logits, e32.logits.cpu(), 3e-2, "logits vs fp32 mode")
    close_robust(eng.grads, e32.grads.cpu(), 5e-2, "grads vs fp32 mode", outlier_frac=2e-3)
