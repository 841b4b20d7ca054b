"""Operand-rounding emulations of the fused kernels, on the CPU in plain PyTorch, for the GPU
parity tests: the same arithmetic as oracle/st_oracle.py (which they build on) with the MFMA
operands rounded where the kernels round them, so that autograd of an emulation is the gradient
the fused backward computes (including its ReLU mask).  Test infrastructure only."""
import math

import torch

from oracle.st_oracle import _lin, _sub, f8_weight_scale, rb, rb8


class _LinearBf16Wgrad(torch.autograd.Function):
    """fc_o of the d = 256 few-queries block as csrc/d256_host.hip runs it: y = x W^T + b and
    dx = g W with hi + lo bf16 operand pairs on the MFMA (fp32-level, emulated exactly), the
    weight gradient dW = rb(g)^T rb(x) with single bf16 operands (k_wgrad256)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return x @ W.t() + b

    @staticmethod
    def backward(ctx, g):
        r = lambda t: t.to(torch.bfloat16).to(torch.float32)
        x, W = ctx.saved_tensors
        g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
        return g @ W, r(g2).t() @ r(x2), g2.sum(0)


def mab0_forward_bf16emu(I, X, p, h, fp8=False):
    """Reassociated mab0 with the MFMA operands (G' = sl2e Qp_h Wk_h, X, P) rounded to bf16 as
    csrc/mab0_bf16.hip does; epilogue fp32 (d = 256: fc_o with hi + lo bf16 pairs).  Returns H.
    d = 256 with dk = 256 (csrc/d256_*.hip): the keys ARE projected - Kp, Vp, the scaled query
    and P are the bf16 operands."""
    B, N, dk = X.shape
    m = I.shape[1]
    d = p["fc_q.weight"].shape[0]
    dh = d // h
    sl2e = math.log2(math.e) / math.sqrt(d)
    Qp = I[0] @ p["fc_q.weight"].t() + p["fc_q.bias"]                  # [m, d]
    if d == 256 and dk == 256 and h * m > 16:
        Xb = rb(X)
        if fp8:            # PCA_MODE_FP8: fc_k / fc_v with e4m3 operands (oracle/st_oracle.py:_lin8)
            Kp = rb(_LinF8.apply(Xb, p["fc_k.weight"], p["fc_k.bias"])).view(B, N, h, dh)
            Vp = rb(_LinF8.apply(Xb, p["fc_v.weight"], p["fc_v.bias"])).view(B, N, h, dh)
        else:
            Kp = rb(Xb @ rb(p["fc_k.weight"]).t() + p["fc_k.bias"]).view(B, N, h, dh)
            Vp = rb(Xb @ rb(p["fc_v.weight"]).t() + p["fc_v.bias"]).view(B, N, h, dh)
        S2 = torch.einsum("qjf,bnjf->bjqn", rb(Qp * sl2e).view(m, h, dh), Kp)
        P = torch.softmax(S2 * math.log(2.0), dim=-1)
        O = Qp.view(1, m, h, dh) + torch.einsum("bjqn,bnjf->bqjf", rb(P), Vp)
        O = O.reshape(B, m, d)
        Z = _LinearBf16Wgrad.apply(O, p["fc_o.weight"], p["fc_o.bias"])
        return O + torch.relu(Z)
    Wk = p["fc_k.weight"].view(h, dh, dk)
    G = torch.einsum("qjf,jfc->jqc", Qp.view(m, h, dh), Wk) * sl2e      # [h, m, dk]
    small = dk <= 4
    Gs, Xs = (G, X) if small else (rb(G), rb(X))
    S2 = torch.einsum("jqc,bnc->bjqn", Gs, Xs)                           # log2-domain scores
    P = torch.softmax(S2 * math.log(2.0), dim=-1)
    T = torch.einsum("bjqn,bnc->bjqc", P if small else rb(P), Xs)
    Wv = p["fc_v.weight"].view(h, dh, dk)
    O = Qp.view(1, m, h, dh) + torch.einsum("bjqc,jfc->bqjf", T, Wv) + p["fc_v.bias"].view(1, 1, h, dh)
    O = O.reshape(B, m, d)
    if d == 256:       # csrc/d256_host.hip: fc_o on the MFMA with hi + lo operand pairs
        Z = _LinearBf16Wgrad.apply(O, p["fc_o.weight"], p["fc_o.bias"])
    else:
        Z = O @ p["fc_o.weight"].t() + p["fc_o.bias"]
    return O + torch.relu(Z)


class _LinF8(torch.autograd.Function):
    """y = rb8(x) rb8(s W)^T / s + b (e4m3 operands, per-tensor power-of-two weight scale: the
    forward of PCA_MODE_FP8) with the mode's straight-through backward in bf16:
    dx = rb(g) rb(W), dW = rb(g)^T rb(x)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        s = f8_weight_scale(W)
        return (rb8(x) @ rb8(W * s).t()) / s + b

    @staticmethod
    def backward(ctx, g):
        r = lambda t: t.to(torch.bfloat16).to(torch.float32)
        x, W = ctx.saved_tensors
        g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
        return r(g) @ r(W), r(g2).t() @ r(x2), g2.sum(0)


def mab1_forward_emu(X, H, p, h, fp8=False):
    """Many-queries block (csrc/mab1_bf16.hip, d256_fused.hip) with bf16 operands; fp8: fc_o with
    e4m3 operands (oracle/st_oracle.py:mab1_forward_fp8emu), differentiable."""
    B, nq, dq = X.shape
    nk = H.shape[1]
    d = p["fc_q.weight"].shape[0]
    dh = d // h
    if dq <= 4:
        Qp = X @ p["fc_q.weight"].t() + p["fc_q.bias"]
    else:
        Qp = rb(X) @ rb(p["fc_q.weight"]).t() + p["fc_q.bias"]
    if d == 256:
        Kp = rb(rb(H) @ rb(p["fc_k.weight"]).t() + p["fc_k.bias"])
        Vp = rb(rb(H) @ rb(p["fc_v.weight"]).t() + p["fc_v.bias"])
    else:
        Kp = rb(_lin(H, p, "fc_k"))
        Vp = rb(_lin(H, p, "fc_v"))
    Qh = rb(Qp).view(B, nq, h, dh).permute(0, 2, 1, 3)
    Kh = Kp.view(B, nk, h, dh).permute(0, 2, 1, 3)
    Vh = Vp.view(B, nk, h, dh).permute(0, 2, 1, 3)
    A = torch.softmax(Qh @ Kh.transpose(-1, -2) / math.sqrt(d), dim=-1)
    Oh = Qp.view(B, nq, h, dh).permute(0, 2, 1, 3) + rb(A) @ Vh
    O = Oh.permute(0, 2, 1, 3).reshape(B, nq, d)
    if fp8:
        Z = _LinF8.apply(rb(O) if d == 256 else O, p["fc_o.weight"], p["fc_o.bias"])
    else:
        Z = rb(O) @ rb(p["fc_o.weight"]).t() + p["fc_o.bias"]
    return O + torch.relu(Z)


def st_forward_emu(X, p, h, fp8=False):
    """Whole ST (Code/models.py:34-44) as the engine's fused bf16 / fp8 mode runs it: the blocks'
    emulations above with bf16 activations between the blocks; logits [B, C]."""
    p = {k[7:] if k.startswith("module.") else k: v for k, v in p.items()}
    Y = X
    for name in ("enc.0.", "enc.1."):
        q = _sub(p, name)
        Hm = mab0_forward_bf16emu(q["I"], Y, _sub(q, "mab0."), h, fp8=fp8)
        Y = rb(mab1_forward_emu(Y, Hm, _sub(q, "mab1."), h, fp8=fp8))
    q = _sub(p, "dec.0.")
    P = mab0_forward_bf16emu(q["S"], Y, _sub(q, "mab."), h)
    return (P @ p["dec.1.weight"].t() + p["dec.1.bias"]).reshape(X.shape[0], -1)
