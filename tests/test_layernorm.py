"""MAB(ln=True): the LayerNorm variants of set_transformer-master/modules.py:14-16,30,32 (no
reference caller enables them).  Oracle against the reference's golden vectors on the CPU, the
HIP path (exact chain + k_layernorm_fwd/bwd) against the same vectors on the GPU."""
import numpy as np
import pytest
import torch

import inputs as gi
from conftest import Golden
from util import T, close


@pytest.fixture(scope="module")
def golden_ln():
    return Golden("golden_ln.npz")


def test_oracle_layernorm_variants(golden_ln):
    from oracle import st_oracle as orc
    for ci, (name, B, nq, nk, dq, dk, d, h) in enumerate(gi.LN_MAB_CASES):
        p = {k: torch.from_numpy(v).clone().requires_grad_(True)
             for k, v in golden_ln.sub(f"{name}/p/").items()}
        Q = torch.from_numpy(gi.randn(910 + ci, B, nq, dq)).requires_grad_(True)
        K = torch.from_numpy(gi.randn(920 + ci, B, nk, dk)).requires_grad_(True)
        G = torch.from_numpy(gi.randn(930 + ci, B, nq, d))
        Y = orc.mab_forward(Q, K, p, h)
        (Y * G).sum().backward()
        close(Y, golden_ln[f"{name}/Y"], 2e-5, f"{name} Y")
        close(Q.grad, golden_ln[f"{name}/dQ"], 5e-5, f"{name} dQ")
        close(K.grad, golden_ln[f"{name}/dK"], 5e-5, f"{name} dK")
        for k, v in p.items():
            close(v.grad, golden_ln[f"{name}/g/{k}"], 5e-5, f"{name} {k}")
    name, B, N, din, d, h, m, C = gi.LN_ST_CASE
    p = {k: torch.from_numpy(v) for k, v in golden_ln.sub(f"{name}/p/").items()}
    lg = orc.st_forward(torch.from_numpy(gi.pc_input(951, B, N, din)), p, h)
    close(lg, golden_ln[f"{name}/logits"], 2e-5, "ST(ln=True) logits")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "auto"])
def test_mab_layernorm_golden(dev, golden_ln, mode):
    """'auto' takes fused kernels where they exist: LayerNorm blocks have none, so both modes run
    the exact chain; an explicit 'bf16' demand is refused (no silent change of arithmetic)."""
    import modules
    import pca_hip
    pca_hip.set_mode(mode)
    try:
        for ci, (name, B, nq, nk, dq, dk, d, h) in enumerate(gi.LN_MAB_CASES):
            mab = modules.MAB(dq, dk, d, h, ln=True).to(dev)
            mab.load_state_dict({k: T(v) for k, v in golden_ln.sub(f"{name}/p/").items()})
            Q = T(gi.randn(910 + ci, B, nq, dq), dev).requires_grad_(True)
            K = T(gi.randn(920 + ci, B, nk, dk), dev).requires_grad_(True)
            G = T(gi.randn(930 + ci, B, nq, d), dev)
            Y = mab(Q, K)
            (Y * G).sum().backward()
            ft, bt = 1e-4, 3e-4
            close(Y, golden_ln[f"{name}/Y"], ft, f"{name} Y")
            close(Q.grad, golden_ln[f"{name}/dQ"], bt, f"{name} dQ")
            close(K.grad, golden_ln[f"{name}/dK"], bt, f"{name} dK")
            for k, prm in mab.named_parameters():
                close(prm.grad, golden_ln[f"{name}/g/{k}"], bt, f"{name} {k}")
            with torch.no_grad():
                close(mab(Q, K), golden_ln[f"{name}/Y"], ft, f"{name} Y(no_grad)")
        pca_hip.set_mode("bf16")
        with pytest.raises(pca_hip.PcaHipError):
            mab(Q, K)
    finally:
        pca_hip.set_mode("f32")


@pytest.mark.gpu
def test_st_layernorm_golden(dev, golden_ln):
    """ST(ln=True) through the nn.Module path: logits, loss, all 57 gradients; the
    whole-model engine refuses it with a clear message."""
    import models
    import pca_hip
    from pca_hip import trainer
    name, B, N, din, d, h, m, C = gi.LN_ST_CASE
    net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d,
                    num_heads=h, ln=True).to(dev)
    sd = golden_ln.sub(f"{name}/p/")
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict({k: T(v) for k, v in sd.items()})
    X = T(gi.pc_input(951, B, N, din), dev)
    y = T(gi.labels(952, B, C), dev)
    logits = net(X)
    loss = pca_hip.cross_entropy(logits, y)
    loss.backward()
    close(logits, golden_ln[f"{name}/logits"], 1e-4, "logits")
    assert abs(float(loss) - float(golden_ln[f"{name}/loss"])) < 2e-5
    for k, prm in net.named_parameters():
        close(prm.grad, golden_ln[f"{name}/g/{k}"], 3e-4, k)
    with pytest.raises(pca_hip.PcaHipError):
        trainer.STEngine(net, B, N)
