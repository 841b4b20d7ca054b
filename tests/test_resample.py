"""The sampling-rate axis of the evaluation sweep (Code/pceval.py:55,61,74:
``librosa.resample(x, fsog, fs, res_type='kaiser_fast', scale=True)``).

PARITY UNPINNED: librosa 0.8 / resampy 0.2.2 are third-party, not vendored in the reference and not
installed here, and the reference holds no resampled fixture.  What these tests pin:

* CPU: ``oracle/resample_oracle.py`` (resampy's published algorithm, documented kaiser_fast design)
  behaves like a resampler - sinusoids below the new Nyquist frequency keep their amplitude (times
  librosa's 1/sqrt(ratio)), content above it is rejected, and it agrees with
  ``scipy.signal.resample_poly`` (a different windowed-sinc design) to a few 1e-3 in the pass band.
* GPU: ``pca_resample`` against that oracle at <= 1e-5, up- and down-sampling incl. the reference's
  ratios (44.1 kHz -> 32 kHz, x0.5, x0.25), and the JSON structure of the two-axis sweep."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import resample_oracle as ro


def _tone(f, fs, n):
    return np.sin(2 * np.pi * f * np.arange(n) / fs)


@pytest.mark.parametrize("fs_new", [32000, 22050, 11025, 48000])
def test_oracle_is_a_resampler(fs_new):
    fs = 44100
    ratio = fs_new / fs
    x = _tone(1000.0, fs, 6000)
    y = ro.resample(x, fs, fs_new, scale=True)
    assert len(y) == int(np.ceil(len(x) * ratio))
    mid = slice(len(y) // 4, 3 * len(y) // 4)
    ref = _tone(1000.0, fs_new, len(y)) / np.sqrt(ratio)
    assert np.abs(y[mid] - ref[mid]).max() < 2e-3 / np.sqrt(ratio)
    # a tone above the new Nyquist frequency is rejected (stop band of the kaiser_fast design)
    if fs_new < fs:
        hi = ro.resample(_tone(0.45 * fs, fs, 6000), fs, fs_new, scale=False)
        assert np.abs(hi[mid]).max() < 2e-2


def test_oracle_close_to_scipy_resample_poly():
    from scipy.signal import resample_poly
    rng = np.random.Generator(np.random.PCG64(3))
    fs, fs_new = 44100, 22050
    # band-limited test signal: a few tones well inside both pass bands
    n = 8000
    x = sum(rng.normal() * _tone(f, fs, n) for f in (220.0, 997.0, 3100.0, 7000.0))
    y = ro.resample(x, fs, fs_new, scale=False)
    z = resample_poly(x, 1, 2)
    mid = slice(len(y) // 4, 3 * len(y) // 4)
    assert np.abs(y[mid] - z[mid]).max() < 2e-2 * np.abs(z).max()


@pytest.mark.gpu
@pytest.mark.parametrize("fs_new", [32000, 22050, 11025, 48000, 44100])
def test_pca_resample_vs_oracle(fs_new):
    import pca_hip
    dev = torch.device("cuda", 0)
    fs = 44100
    rng = np.random.Generator(np.random.PCG64(9))
    x = (0.3 * rng.normal(size=5000) + _tone(440.0, fs, 5000)).astype(np.float32)
    ref = ro.resample(x.astype(np.float64), fs, fs_new, scale=True)
    got = pca_hip.resample(torch.from_numpy(x).to(dev), fs, fs_new, scale=True).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    got2 = pca_hip.resample(torch.from_numpy(x).to(dev), fs, fs_new, scale=False).cpu().numpy()
    assert np.abs(got2 - ro.resample(x.astype(np.float64), fs, fs_new, scale=False)).max() <= 1e-5 * 4


@pytest.mark.gpu
def test_sweep_over_sampling_rates_json(tmp_path):
    """reframe_sweep(list_Fs=...): the {"list_Fs", "list_N", "data": {Fs: [acc per N]}} dictionary of
    Code/pceval.py:57-59,99-104, every rate resampled on the device first; the fs_og column equals the
    single-rate sweep."""
    import evalsweep
    import models
    from oracle import st_oracle as orc
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    net = models.ST(dim_input=2, dim_output=10, num_inds=8, dim_hidden=32, num_heads=4).to(dev)
    fs, Nfft = 44100, 1024
    waves = [orc.synth_clip(i, i % 10, seconds=0.4, fs=fs) for i in range(3)]
    labels = [i % 10 for i in range(3)]
    clips = [torch.from_numpy(np.ascontiguousarray(w)).to(dev) for w in waves]
    list_Fs = [fs, 32000, 0.5 * fs, 0.25 * fs]
    list_N = [Nfft, int(0.6 * Nfft)]
    jf = str(tmp_path / "FST_expt1.json")
    out = evalsweep.reframe_sweep(net, clips, labels, fs, list_N, list_Fs=list_Fs, json_file=jf)
    assert out["list_Fs"] == list_Fs and out["list_N"] == list_N
    assert list(out["data"].keys()) == list_Fs
    assert all(len(v) == len(list_N) and all(0.0 <= a <= 1.0 for a in v) for v in out["data"].values())
    single = evalsweep.reframe_sweep(net, clips, labels, fs, list_N)
    assert out["data"][fs] == single["data"][fs]
    back = json.load(open(jf))
    assert set(back) == {"data", "list_Fs", "list_N"} and len(back["data"]) == len(list_Fs)
