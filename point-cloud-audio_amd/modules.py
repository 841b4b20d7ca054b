"""Set-attention blocks on MI355X: drop-in for ``set_transformer-master/modules.py``.

Same class names, constructor signatures, sub-module / parameter names and state_dict
layout as the reference (modules.py:6-63), so ``from modules import ISAB, PMA, SAB``
(Code/models.py:10) and the shipped checkpoints keep working.  The arithmetic of
``MAB.forward`` (modules.py:19-33) runs in libpca_hip.so; nothing here falls back to
PyTorch ops or to the CPU.

Differences a caller can observe:
  * tensors must live on a HIP device ('cuda'); a CPU tensor raises PcaHipError;
  * ``ln=True`` (never enabled by any reference caller, Code/models.py:31) is supported
    through the exact strided-GEMM chain (no fused kernels), with the reference's
    ``ln0`` / ``ln1`` sub-modules and state_dict keys;
  * an optional ``lengths`` / ``key_lengths`` argument (int[B]) marks padded batches of
    variable-size sets: points at and beyond ``lengths[b]`` are ignored exactly as if set b
    had been truncated (the reference has dense batches only, so this is an extension);
  * ISAB / PMA do not materialise ``I.repeat(B,1,1)`` / ``S.repeat(B,1,1)``
    (modules.py:52,63): the learned query is projected once and shared by all sets.
"""
import torch
import torch.nn as nn

import pca_hip

__all__ = ["MAB", "SAB", "ISAB", "PMA"]


class MAB(nn.Module):
    """Multihead attention block, reference modules.py:6-33."""

    def __init__(self, dim_Q, dim_K, dim_V, num_heads, ln=False):
        super().__init__()
        if dim_V % num_heads != 0:
            raise ValueError(f"dim_V={dim_V} must be divisible by num_heads={num_heads}")
        self.dim_V = dim_V
        self.num_heads = num_heads
        self.fc_q = nn.Linear(dim_Q, dim_V)
        self.fc_k = nn.Linear(dim_K, dim_V)
        self.fc_v = nn.Linear(dim_K, dim_V)
        if ln:      # registration order of the reference (modules.py:14-17): before fc_o
            self.ln0 = nn.LayerNorm(dim_V)
            self.ln1 = nn.LayerNorm(dim_V)
        self.fc_o = nn.Linear(dim_V, dim_V)

    def _params(self):
        return (self.fc_q.weight, self.fc_q.bias, self.fc_k.weight, self.fc_k.bias,
                self.fc_v.weight, self.fc_v.bias, self.fc_o.weight, self.fc_o.bias)

    def _ln_params(self):
        ln0 = getattr(self, "ln0", None)          # the reference probes them the same way
        if ln0 is None:
            return None
        return (ln0.weight, ln0.bias, self.ln1.weight, self.ln1.bias)

    def forward(self, Q, K, q_shared=False, key_lengths=None):
        """Q [B,nq,dim_Q] (or the shared learned query [1,nq,dim_Q] with q_shared),
        K [B,nk,dim_K] -> [B,nq,dim_V].  key_lengths int[B]: valid keys per set."""
        if torch.is_grad_enabled() and (
                Q.requires_grad or K.requires_grad
                or any(p.requires_grad for p in self.parameters())):
            return pca_hip.mab(Q, K, *self._params(), self.num_heads, q_shared, key_lengths,
                               self._ln_params())
        return pca_hip.mab_infer(Q, K, self._params(), self.num_heads, q_shared, key_lengths,
                                 self._ln_params())


class SAB(nn.Module):
    """Self-attention block MAB(X, X), reference modules.py:35-41."""

    def __init__(self, dim_in, dim_out, num_heads, ln=False):
        super().__init__()
        self.mab = MAB(dim_in, dim_in, dim_out, num_heads, ln=ln)

    def forward(self, X, lengths=None):
        return self.mab(X, X, key_lengths=lengths)


class ISAB(nn.Module):
    """Induced set-attention block, reference modules.py:43-53."""

    def __init__(self, dim_in, dim_out, num_heads, num_inds, ln=False):
        super().__init__()
        self.I = nn.Parameter(torch.empty(1, num_inds, dim_out))
        nn.init.xavier_uniform_(self.I)
        self.mab0 = MAB(dim_out, dim_in, dim_out, num_heads, ln=ln)
        self.mab1 = MAB(dim_in, dim_out, dim_out, num_heads, ln=ln)

    def forward(self, X, lengths=None):
        H = self.mab0(self.I, X, q_shared=True, key_lengths=lengths)     # [B, m, d]
        return self.mab1(X, H)      # rows at and beyond lengths[b] are padding: never read


class PMA(nn.Module):
    """Pooling by multihead attention, reference modules.py:55-63."""

    def __init__(self, dim, num_heads, num_seeds, ln=False):
        super().__init__()
        self.S = nn.Parameter(torch.empty(1, num_seeds, dim))
        nn.init.xavier_uniform_(self.S)
        self.mab = MAB(dim, dim, dim, num_heads, ln=ln)

    def forward(self, X, lengths=None):
        return self.mab(self.S, X, q_shared=True, key_lengths=lengths)
