#!/usr/bin/env python3
"""golden_base.npz: the two comparison baselines and the metadata helpers, from the REAL
reference on CPU.  Run in the build container only: ``python tests/golden/make_golden_base.py``.

  * baseline_ff / CNN_classifier (Code/models.py:47-119) at small widths: the reference's own
    randomly initialised state_dict (arrays) + eval-mode outputs on seeded inputs; the key
    names and shapes of the shipped FB / CNNTemp checkpoints (names + shapes only);
  * ESC_baseline / ESC_baseline_temporal / ESC_baseline_temporal_maxK items
    (Code/dataset.py:10-27, 82-135) on seeded arrays;
  * load_esc / tt_split (Code/data_processing.py:8-65) on a synthetic esc50-style CSV that the
    test regenerates from ``inputs.base_csv_rows()``.
Only data is stored: inputs come from seeds (inputs.py), outputs from the reference."""
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PCA_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REF, "set_transformer-master"))
sys.path.insert(0, os.path.join(REF, "Code"))
os.chdir(os.path.join(REF, "Code"))

import inputs as gi  # noqa: E402
import models as ref_models  # noqa: E402  (reference)
import dataset as ref_dataset  # noqa: E402
import data_processing as ref_dp  # noqa: E402


def main():
    out = {}
    # ---- models -------------------------------------------------------------------------
    torch.manual_seed(11)
    ff = ref_models.baseline_ff(gi.BASE_FF_DIMS, gi.BASE_NCLASS, p=0.5).eval()
    for k, v in ff.state_dict().items():
        out["ff/p/" + k] = v.numpy()
    x = gi.base_ff_input()
    import warnings
    with warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("ignore")
        out["ff/y"] = ff(torch.from_numpy(x)).numpy()
    cnn = ref_models.CNN_classifier(gi.BASE_NT, gi.BASE_NF, gi.BASE_CNN_DIMS, gi.BASE_NCLASS).eval()
    for k, v in cnn.state_dict().items():
        out["cnn/p/" + k] = v.numpy()
    xc = gi.base_cnn_input()
    with torch.no_grad():
        out["cnn/y"] = cnn(torch.from_numpy(xc)).numpy()
        out["cnn/y1"] = cnn(torch.from_numpy(xc[:1])).numpy()          # batch of one: squeeze quirk
    for tag, pat in (("fb", "FB(2021-04-26 17_45_43.476736)"),
                     ("cnntemp", "CNNTemp(2021-04-27 00_35_22.823854)")):
        sd = torch.load(os.path.join(REF, "Code", "model_saves", pat + "_net.pth"),
                        weights_only=True, map_location="cpu")
        out[f"shipped/{tag}/keys"] = np.array(list(sd.keys()))
        out[f"shipped/{tag}/shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    # ---- datasets -----------------------------------------------------------------------
    x2, y2, x3, y3 = gi.base_dataset_inputs()
    ds = ref_dataset.ESC_baseline(x2, y2)
    out["ds/base/len"] = np.int64(len(ds))
    for i in (0, 3):
        lbl, v = ds[i]
        out[f"ds/base/item{i}"] = v.numpy()
        out[f"ds/base/label{i}"] = np.int64(lbl)
    dt = ref_dataset.ESC_baseline_temporal(x3, y3)
    out["ds/temp/len"] = np.int64(len(dt))
    for i in (0, 2):
        lbl, v = dt[i]
        out[f"ds/temp/item{i}"] = v.numpy()
        out[f"ds/temp/label{i}"] = lbl.numpy()
    for K in gi.BASE_K:
        dm = ref_dataset.ESC_baseline_temporal_maxK(x3, y3, K, "max")
        for i in (0, 2):
            lbl, v = dm[i]
            out[f"ds/maxK{K}/item{i}"] = v.numpy()
        np.random.seed(5)
        dr = ref_dataset.ESC_baseline_temporal_maxK(x3, y3, K, "rand")
        out[f"ds/randK{K}/item1"] = dr[1][1].numpy()
    # ---- metadata -----------------------------------------------------------------------
    csv_path = "/tmp/pca_golden_esc50.csv"
    open(csv_path, "w").write(gi.base_csv_text())
    locs, lab = ref_dp.load_esc(loc=csv_path, loc_audio="audio/")
    out["dp/locs"] = np.array(list(locs))
    out["dp/labels"] = np.asarray(lab, dtype=np.int64)
    np.random.seed(3)
    a, la, b, lb = ref_dp.tt_split(locs, lab, f=0.8)
    out["dp/train"] = np.array(a); out["dp/l_train"] = np.asarray(la, dtype=np.int64)
    out["dp/test"] = np.array(b); out["dp/l_test"] = np.asarray(lb, dtype=np.int64)
    locs2, lab2 = ref_dp.load_esc(loc=csv_path, loc_audio="x/", list_categories=["rain", "dog"])
    out["dp/locs2"] = np.array(list(locs2)); out["dp/labels2"] = np.asarray(lab2, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "golden_base.npz"), **out)
    print("wrote golden_base.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
