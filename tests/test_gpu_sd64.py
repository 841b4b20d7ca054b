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
