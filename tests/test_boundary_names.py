"""Drop-in boundary: every name the reference's train / eval scripts use
after ``from dataset import *`` / ``from data_processing import *`` / ``from utils import *`` /
``from models import *`` / ``from modules import ...`` resolves against point-cloud-audio_amd/,
and the stock-PyTorch / numpy pieces (the two comparison baselines, their datasets, the ESC-50
metadata helpers) reproduce the reference's outputs (tests/golden/golden_base.npz, made by
tests/golden/make_golden_base.py from the real reference).  CPU only: none of these is on the
accelerated path."""
import warnings

import numpy as np
import pytest
import torch

from conftest import Golden

import inputs as gi

# names of SURVEY.md section 8(b), by module
SURFACE = {
    "modules": ["MAB", "SAB", "ISAB", "PMA"],
    "models": ["ST", "baseline_ff", "CNN_classifier", "ISAB", "PMA", "SAB"],
    "dataset": ["ESC_baseline", "ESC_pc", "ESC_pc_ss", "ESC_baseline_temporal",
                "ESC_baseline_temporal_maxK", "ESC_pc_temp", "ESC_pc_temp_maxKSS",
                "ESC_pc_temp_randKSS", "ESC_pc_temp_importancerandKSS"],
    "utils": ["count_parameters", "pc_maxK", "pc_randK", "pc_maxK_replace", "pc_randK_replace"],
    "data_processing": ["load_esc", "tt_split"],
}


@pytest.fixture(scope="module")
def gb():
    return Golden("golden_base.npz")


def test_every_reference_name_resolves_through_star_imports():
    for mod, names in SURFACE.items():
        ns = {}
        exec(f"from {mod} import *", ns)           # what the reference scripts do
        for n in names:
            assert n in ns, f"from {mod} import * does not provide {n}"


def test_baseline_ff_matches_reference(gb):
    import models
    net = models.baseline_ff(gi.BASE_FF_DIMS, gi.BASE_NCLASS, p=0.5)
    sd = {k: torch.from_numpy(v) for k, v in gb.sub("ff/p/").items()}
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    net.eval()
    with warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("ignore")            # nn.Softmax() implicit dim, as the reference
        y = net(torch.from_numpy(gi.base_ff_input())).numpy()
    np.testing.assert_allclose(y, gb["ff/y"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(y.sum(1), 1.0, atol=1e-5)     # Softmax inside the model (quirk)
    assert isinstance(net.dpout, torch.nn.Dropout) and net.dpout.p == 0.5
    assert net.layer_dims == gi.BASE_FF_DIMS


def test_cnn_classifier_matches_reference(gb):
    import models
    net = models.CNN_classifier(gi.BASE_NT, gi.BASE_NF, gi.BASE_CNN_DIMS, gi.BASE_NCLASS)
    sd = {k: torch.from_numpy(v) for k, v in gb.sub("cnn/p/").items()}
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    net.eval()
    x = torch.from_numpy(gi.base_cnn_input())
    with torch.no_grad():
        np.testing.assert_allclose(net(x).numpy(), gb["cnn/y"], rtol=0, atol=1e-6)
        y1 = net(x[:1]).numpy()
    assert y1.shape == gb["cnn/y1"].shape == (gi.BASE_NCLASS,)       # .squeeze() drops the batch
    np.testing.assert_allclose(y1, gb["cnn/y1"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("tag,build", [
    ("fb", lambda m: m.baseline_ff([1025, 513, 256], 10, p=0.5)),
    ("cnntemp", lambda m: m.CNN_classifier(10, 512, [512, 256, 100], 10, p=0.5)),
])
def test_shipped_baseline_checkpoint_layout(gb, tag, build):
    """Key names and shapes of Code/model_saves/FB(...)_net.pth / CNNTemp(...)_net.pth at the
    sizes their _config.json name: load_state_dict of the shipped files would succeed."""
    import models
    sd = build(models).state_dict()
    assert list(sd.keys()) == list(gb[f"shipped/{tag}/keys"])
    assert [",".join(map(str, v.shape)) for v in sd.values()] == list(gb[f"shipped/{tag}/shapes"])


def test_baseline_datasets_match_reference(gb):
    import dataset
    x2, y2, x3, y3 = gi.base_dataset_inputs()
    ds = dataset.ESC_baseline(x2, y2)
    assert len(ds) == int(gb["ds/base/len"])
    for i in (0, 3):
        lbl, v = ds[i]                                  # (label, x): NOT (x, label)
        assert int(lbl) == int(gb[f"ds/base/label{i}"])
        assert v.dtype == torch.float32 and np.array_equal(v.numpy(), gb[f"ds/base/item{i}"])
    dt = dataset.ESC_baseline_temporal(x3, y3)
    assert len(dt) == int(gb["ds/temp/len"])
    for i in (0, 2):
        lbl, v = dt[i]
        assert torch.is_tensor(lbl) and int(lbl) == int(gb[f"ds/temp/label{i}"])
        assert np.array_equal(v.numpy(), gb[f"ds/temp/item{i}"])
    for K in gi.BASE_K:
        dm = dataset.ESC_baseline_temporal_maxK(x3, y3, K, "max")
        for i in (0, 2):
            assert np.array_equal(dm[i][1].numpy(), gb[f"ds/maxK{K}/item{i}"]), (K, i)
        np.random.seed(5)                               # same global-RNG consumption
        dr = dataset.ESC_baseline_temporal_maxK(x3, y3, K, "rand")
        assert np.array_equal(dr[1][1].numpy(), gb[f"ds/randK{K}/item1"]), K


def test_load_esc_and_tt_split_match_reference(gb, tmp_path):
    import data_processing as dp
    csv = tmp_path / "esc50.csv"
    csv.write_text(gi.base_csv_text())
    locs, lab = dp.load_esc(loc=str(csv), loc_audio="audio/")
    assert list(locs) == list(gb["dp/locs"]) and np.array_equal(lab, gb["dp/labels"])
    np.random.seed(3)
    a, la, b, lb = dp.tt_split(locs, lab, f=0.8)
    assert a == list(gb["dp/train"]) and b == list(gb["dp/test"])
    assert la == list(gb["dp/l_train"]) and lb == list(gb["dp/l_test"])
    assert not set(a) & set(b) and len(a) + len(b) == len(locs)
    locs2, lab2 = dp.load_esc(loc=str(csv), loc_audio="x/", list_categories=["rain", "dog"])
    assert list(locs2) == list(gb["dp/locs2"]) and np.array_equal(lab2, gb["dp/labels2"])
