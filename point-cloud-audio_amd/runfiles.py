"""Run files of a trained model: ``<ARCH>(<timestamp>)_net.pth`` + ``<ARCH>(<timestamp>)_config.json``
in the layout the reference's training scripts write and its evaluation scripts read.

Writer  <- ``Code/settransformer.py:134-162`` (FST) / ``Code/settransformertemp.py:147-177`` (3ST):
           ``torch.save(model.state_dict())`` of the DataParallel-wrapped model (keys carry the
           ``module.`` prefix) and a flat JSON dict of everything needed to rebuild it.
Reader  <- ``Code/pceval.py:23-47`` / ``Code/pc_temp3d_eval.py:23-49``: JSON -> ``ST(dim_hidden=dhidden,
           num_heads=nheads, num_inds=ninds)`` (3ST: ``dim_input=3``) -> ``nn.DataParallel`` ->
           ``load_state_dict``.

The key set and key order of the JSON follow the shipped files
(``Code/model_saves/FST(...)_config.json``: ``numpy_seed``; ``3ST(...)_config.json``: ``Ntemp`` after
``trim_dB`` and ``np_seed``), so either side of the pair can be swapped with the reference's.
Weights are read with ``torch.load(weights_only=True)`` only.
"""
import json
import os
from datetime import datetime
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from models import ST

__all__ = ["ARCHITECTURES", "run_config", "save_run", "load_run"]

ARCHITECTURES = {
    "FST": "FST (Framewise Set Transformer)",
    "3ST": "3ST (Set Transformer Temporal)",
}


def run_config(arch: str, *, epochs: int, weight_decay: float, window_size: int,
               hop_factor: float, trim_dB: float, sampling_rate: int, classes: int, dhidden: int,
               nheads: int, ninds: int, batch_size: int, learning_rate: float, dataset: str,
               numpy_seed: int, torch_seed: int, model_params: int,
               Ntemp: Optional[int] = None) -> Dict:
    """The config dict of one run, keys in the order of the reference's writer."""
    if arch not in ARCHITECTURES:
        raise ValueError(f"unknown architecture {arch!r} (one of {sorted(ARCHITECTURES)})")
    if (arch == "3ST") != (Ntemp is not None):
        raise ValueError("Ntemp belongs to the 3ST config and only to it")
    c = dict(epochs=epochs, weight_decay=weight_decay, window_size=window_size,
             hop_factor=hop_factor, trim_dB=trim_dB)
    if arch == "3ST":
        c["Ntemp"] = Ntemp
    c.update(sampling_rate=sampling_rate, classes=classes, dhidden=dhidden, nheads=nheads,
             ninds=ninds, batch_size=batch_size, learning_rate=learning_rate, dataset=dataset,
             architecture=ARCHITECTURES[arch])
    # the temporal script names the numpy seed differently (settransformertemp.py:163)
    c["np_seed" if arch == "3ST" else "numpy_seed"] = numpy_seed
    c.update(torch_seed=torch_seed, model_params=model_params)
    return c


def _arch_of(config: Dict) -> str:
    for k, v in ARCHITECTURES.items():
        if config.get("architecture") == v:
            return k
    raise ValueError(f"not a Set Transformer run: architecture = {config.get('architecture')!r}")


def save_run(model, config: Dict, directory: str, now: Optional[datetime] = None
             ) -> Tuple[str, str]:
    """Write the weight and config files of a run; returns (pth path, json path).

    ``model`` may be an ``ST`` or its ``nn.DataParallel`` wrapper; the file always holds the
    wrapped key names, like the reference's."""
    arch = _arch_of(config)
    sd = model.state_dict()
    if not isinstance(model, nn.DataParallel):
        sd = {"module." + k: v for k, v in sd.items()}
    sd = {k: v.detach().to("cpu").contiguous() for k, v in sd.items()}
    stamp = str(now or datetime.now())
    os.makedirs(directory, exist_ok=True)
    stem = os.path.join(directory, f"{arch}({stamp})")
    torch.save(sd, stem + "_net.pth")
    with open(stem + "_config.json", "w") as fp:
        json.dump(config, fp)
    return stem + "_net.pth", stem + "_config.json"


def load_run(json_path: str, pth_path: Optional[str] = None, device=None,
             dim_output: Optional[int] = None):
    """(DataParallel-wrapped ST, config) of a run.  ``pth_path`` defaults to the JSON's sibling;
    ``device`` defaults to the current HIP device when there is one (the engine and
    ``evalsweep.reframe_sweep`` take the wrapped model as it is), else the CPU.

    The evaluation scripts build the model with the default ``dim_output=10``; ``dim_output``
    overrides it, otherwise it is read off the classifier weight in the file."""
    with open(json_path) as fp:
        config = json.load(fp)
    arch = _arch_of(config)
    if pth_path is None:
        pth_path = json_path[: -len("_config.json")] + "_net.pth"
    sd = torch.load(pth_path, map_location="cpu", weights_only=True)
    if not all(k.startswith("module.") for k in sd):
        sd = {"module." + k: v for k, v in sd.items()}
    if dim_output is None:
        dim_output = int(sd["module.dec.1.weight"].shape[0])
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    model = ST(dim_input=3 if arch == "3ST" else 2, dim_output=dim_output,
               dim_hidden=config["dhidden"], num_heads=config["nheads"],
               num_inds=config["ninds"]).to(device)
    model = nn.DataParallel(model)
    model.load_state_dict(sd)
    return model, config
