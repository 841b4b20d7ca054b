"""Run files (weights + config JSON) in the layout of Code/settransformer.py:134-162 and
Code/settransformertemp.py:147-177, read back the way Code/pceval.py:23-47 does."""
import json
import os
from datetime import datetime

import pytest
import torch

import models
import runfiles

# the two config files the reference ships (Code/model_saves/*_config.json), as data
FST_SHIPPED = {"epochs": 500, "weight_decay": 0.001, "window_size": 2048, "hop_factor": 0.5,
               "trim_dB": 60, "sampling_rate": 44100, "classes": 10, "dhidden": 64, "nheads": 8,
               "ninds": 64, "batch_size": 128, "learning_rate": 0.001, "dataset": "ESC10",
               "architecture": "FST (Framewise Set Transformer)", "numpy_seed": 1,
               "torch_seed": 1, "model_params": 80202}
TST_SHIPPED = {"epochs": 500, "weight_decay": 0.001, "window_size": 1024, "hop_factor": 0.5,
               "trim_dB": 60, "Ntemp": 10, "sampling_rate": 44100, "classes": 10, "dhidden": 64,
               "nheads": 8, "ninds": 64, "batch_size": 16, "learning_rate": 0.001,
               "dataset": "ESC10", "architecture": "3ST (Set Transformer Temporal)",
               "np_seed": 1, "torch_seed": 1, "model_params": 80394}


def _kwargs(shipped):
    kw = {k: v for k, v in shipped.items() if k != "architecture"}
    if "np_seed" in kw:
        kw["numpy_seed"] = kw.pop("np_seed")
    return kw


@pytest.mark.parametrize("arch,shipped", [("FST", FST_SHIPPED), ("3ST", TST_SHIPPED)])
def test_config_matches_the_shipped_files(arch, shipped):
    cfg = runfiles.run_config(arch, **_kwargs(shipped))
    assert cfg == shipped
    assert list(cfg) == list(shipped)                      # same key order in the JSON text
    assert json.dumps(cfg) == json.dumps(shipped)


def test_config_rejects_mixed_up_arguments():
    with pytest.raises(ValueError):
        runfiles.run_config("FST", **_kwargs(TST_SHIPPED))   # Ntemp on a framewise run
    kw = _kwargs(FST_SHIPPED)
    with pytest.raises(ValueError):
        runfiles.run_config("3ST", **kw)                       # no Ntemp
    with pytest.raises(ValueError):
        runfiles.run_config("CNN", **kw)


@pytest.mark.parametrize("arch,din,wrap", [("FST", 2, False), ("3ST", 3, True)])
def test_round_trip(tmp_path, arch, din, wrap):
    torch.manual_seed(3)
    shipped = FST_SHIPPED if arch == "FST" else TST_SHIPPED
    net = models.ST(dim_input=din, dim_hidden=64, num_heads=8, num_inds=64)
    cfg = runfiles.run_config(arch, **_kwargs(shipped))
    assert sum(p.numel() for p in net.parameters()) == cfg["model_params"]
    saved = torch.nn.DataParallel(net) if wrap else net
    now = datetime(2021, 4, 26, 21, 49, 40, 977943)
    pth, js = runfiles.save_run(saved, cfg, str(tmp_path / "model_saves"), now=now)
    assert os.path.basename(pth) == f"{arch}(2021-04-26 21:49:40.977943)_net.pth"
    assert os.path.basename(js) == f"{arch}(2021-04-26 21:49:40.977943)_config.json"
    sd = torch.load(pth, weights_only=True)
    assert len(sd) == 45 and all(k.startswith("module.") for k in sd)     # as saved from DP
    assert json.load(open(js)) == shipped
    back, cfg2 = runfiles.load_run(js)
    assert cfg2 == shipped and isinstance(back, torch.nn.DataParallel)
    for (k, a), (_, b) in zip(net.state_dict().items(), back.module.state_dict().items()):
        assert torch.equal(a, b.cpu()), k


def test_loads_the_reference_layout(tmp_path, golden_ckpt):
    """A weight file with the shipped FST tensors + the shipped JSON -> the evaluation model."""
    sd = {k: torch.from_numpy(v) for k, v in golden_ckpt.sub("fst/p/").items()}
    stem = str(tmp_path / "FST(x)")
    torch.save(sd, stem + "_net.pth")
    json.dump(FST_SHIPPED, open(stem + "_config.json", "w"))
    model, cfg = runfiles.load_run(stem + "_config.json")
    assert cfg["dhidden"] == 64 and model.module.dec[1].weight.shape == (10, 64)
    for k, v in sd.items():
        assert torch.equal(model.state_dict()[k].cpu(), v)
