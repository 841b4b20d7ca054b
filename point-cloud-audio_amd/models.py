"""Set Transformer classifier on MI355X: drop-in for ``Code/models.py`` (class ``ST``).

Constructor arguments, attribute names (``enc``, ``dec``) and the 45 state_dict keys are
those of the reference (Code/models.py:13-44), so ``load_state_dict`` of the shipped
``*_net.pth`` files works bare or under ``nn.DataParallel`` ('module.' prefix).
"""
import torch
import torch.nn as nn

import pca_hip
from modules import ISAB, PMA, SAB  # noqa: F401  (SAB re-exported as in the reference)

__all__ = ["ST"]


class _Linear(nn.Linear):
    """nn.Linear whose matmul runs on the library GEMM (keeps keys weight / bias)."""

    def forward(self, x):
        return pca_hip.linear(x, self.weight, self.bias)


class ST(nn.Module):
    """enc = ISAB, ISAB ; dec = PMA, Linear ; forward = dec(enc(X)).squeeze()
    (Code/models.py:23-44).

    dim_input   width of a point (2: (f, logmag); 3: (f, t, logmag))
    num_outputs seeds of the PMA pooling
    dim_output  classes
    num_inds    inducing points per ISAB
    dim_hidden  hidden width d
    num_heads   attention heads (score scale is 1/sqrt(dim_hidden))
    """

    def __init__(self, dim_input=2, num_outputs=1, dim_output=10, num_inds=4, dim_hidden=4,
                 num_heads=2, ln=False):
        super().__init__()
        self.enc = nn.Sequential(
            ISAB(dim_input, dim_hidden, num_heads, num_inds, ln=ln),
            ISAB(dim_hidden, dim_hidden, num_heads, num_inds, ln=ln),
        )
        self.dec = nn.Sequential(
            PMA(dim_hidden, num_heads, num_outputs, ln=ln),
            _Linear(dim_hidden, dim_output),
        )

    def forward(self, X, lengths=None):
        """X [B, N, dim_input] -> logits [B, dim_output] (squeezed as the reference does).
        ``lengths`` (optional int[B]): X is a padded batch of variable-size sets; set b has
        lengths[b] points (extension: the reference only has dense batches)."""
        if lengths is None:
            return self.dec(self.enc(X)).squeeze()
        for isab in self.enc:
            X = isab(X, lengths)
        return self.dec[1](self.dec[0](X, lengths)).squeeze()
