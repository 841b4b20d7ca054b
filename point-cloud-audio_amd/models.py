"""Set Transformer classifier on MI355X: drop-in for ``Code/models.py`` (class ``ST``).

Constructor arguments, attribute names (``enc``, ``dec``) and the 45 state_dict keys are
those of the reference (Code/models.py:13-44), so ``load_state_dict`` of the shipped
``*_net.pth`` files works bare or under ``nn.DataParallel`` ('module.' prefix).

``baseline_ff`` (FB) and ``CNN_classifier`` (CNN_temp) are the paper's two comparison models
(Code/models.py:47-119).  They are not set models and not on the accelerated path: they are
plain PyTorch modules kept here only so that ``from models import *`` in
``Code/baseline_eval.py:18`` / ``Code/baseline_temp_eval.py`` resolves against this directory
and the shipped ``FB(...)`` / ``CNNTemp(...)`` checkpoints load (same sub-module names).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

import pca_hip
from modules import ISAB, PMA, SAB  # noqa: F401  (SAB re-exported as in the reference)

__all__ = ["ST", "baseline_ff", "CNN_classifier", "ISAB", "PMA", "SAB"]


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


def _mlp(dims, head_name, nclasses):
    """Linear + LeakyReLU stack ``dims[0] -> ... -> dims[-1]`` followed by a Linear named
    ``head_name`` to ``nclasses`` (layer names are the checkpoint keys of the reference)."""
    layers = OrderedDict()
    for i in range(len(dims) - 1):
        layers[f"Encoder_Layer_{i:d}"] = nn.Linear(dims[i], dims[i + 1])
        layers[f"Activation_{i:d}"] = nn.LeakyReLU()
    layers[head_name] = nn.Linear(dims[-1], nclasses)
    return layers


class baseline_ff(nn.Module):
    """FB: framewise feed-forward baseline (Code/models.py:47-88).

    layer_dims  widths from the input frame to the last hidden layer, e.g. [1025, 513, 256]
    nclasses    classes
    p           dropout on the INPUT frame (emulates random sub-sampling of bins)

    Reference quirk kept on purpose: the network ends in ``nn.Softmax()`` (implicit dim), so
    ``forward`` returns probabilities and the train script feeds them to CrossEntropyLoss.
    state_dict keys: ``ENC_NN.Encoder_Layer_{i}.*``, ``ENC_NN.Code_Linear.*``.
    """

    def __init__(self, layer_dims, nclasses, p=0.5):
        super().__init__()
        self.layer_dims = layer_dims
        self.dpout = nn.Dropout(p=p)
        layers = _mlp(layer_dims, "Code_Linear", nclasses)
        layers["Softmax"] = nn.Softmax()
        self.ENC_NN = nn.Sequential(layers)

    def forward(self, x):
        return self.ENC_NN(self.dpout(x))


class CNN_classifier(nn.Module):
    """CNN_temp: temporal baseline (Code/models.py:91-119).  Input [batch, Nt, Nf]; dropout
    on the input, one Conv2d(1, 1, (Nt, Nf + 1 - layer_dims[0])) that collapses the Nt frames,
    then an MLP over the remaining ``layer_dims[0]`` columns; returns logits (no softmax).
    state_dict keys: ``cnn.*``, ``linear.Encoder_Layer_{i}.*``, ``linear.Logits.*``.
    """

    def __init__(self, Nt, Nf, layer_dims, nclass, p=0.5):
        super().__init__()
        self.cnn = nn.Conv2d(1, 1, (Nt, Nf + 1 - layer_dims[0]), stride=(1, 1), padding=(0, 0))
        self.dpout = nn.Dropout(p=p)
        self.linear = nn.Sequential(_mlp(layer_dims, "Logits", nclass))

    def forward(self, x):
        y = self.cnn(self.dpout(x.unsqueeze(1)))
        return self.linear(y.squeeze())
