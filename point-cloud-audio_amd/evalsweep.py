"""Evaluation-time re-framing sweep on the device: the counterpart of the loop of
``Code/pceval.py:61-101`` (framewise model) for waveforms that are already decoded.

For every analysis length N the reference re-computes
``librosa.stft(x, n_fft=2**ceil(log2 N), win_length=N, hop_length=int(N*hf), window='hann') / N``
on the host, builds ``ESC_pc`` and runs the model over shuffled batches of 8, skipping the
short tail (``Code/pceval.py:76-97``).  Here the STFT, the point-set packing and the model all
run on the GPU; accuracy does not depend on the batch order, so batches are taken in order
and, as in the reference, an incomplete last batch is left out.

The sampling-rate axis (``for F in list_Fs``, ``librosa.resample(x, fsog, fs, 'kaiser_fast',
scale=True)``, ``Code/pceval.py:55,61,74``) runs on the device too: ``pca_hip.resample`` is a
band-limited sinc interpolation with resampy's documented ``kaiser_fast`` design.  librosa / resampy are
third-party, not vendored and not installed here, and the reference holds no resampled fixture, so this
axis is **parity unpinned** (SURVEY.md 8c); the kernel is checked against the CPU restatement of the
same algorithm (tests/test_resample.py).
Not covered: ``librosa.load`` / ``effects.trim`` (file decoding and silence trimming: the clips passed
in are waveforms already).
"""
import json
import math
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

import pca_hip
from dataset import ESC_pc
from pca_hip import _lib
from pca_hip.trainer import STEngine

__all__ = ["reframe_sweep", "framewise_dataset"]


def framewise_dataset(clips: Sequence[torch.Tensor], labels: Sequence[int], fs: float, N: int,
                      hf: float = 0.5) -> ESC_pc:
    """ESC_pc over all frames of ``clips`` analysed with window length N
    (Code/pceval.py:74-83): n_fft = next power of two, hop = int(N*hf), all 1 + n_fft/2 bins,
    farr = linspace(0, fs/2, F) / fs.  Everything stays on the device."""
    n_fft = 1 << int(math.ceil(math.log2(N)))
    hop = int(N * hf)
    specs, labs = [], []
    for x, y in zip(clips, labels):
        s = pca_hip.stft_logmag(x, n_fft, win_length=N, hop=hop, frame_major=True)   # [T, F]
        specs.append(s)
        labs.append(torch.full((s.shape[0],), int(y), dtype=torch.int64, device=s.device))
    spec = torch.cat(specs, 0)
    F = spec.shape[1]
    farr = np.linspace(0, fs / 2, F) / fs
    return ESC_pc.from_device(spec, torch.cat(labs, 0), farr)


@torch.no_grad()
def reframe_sweep(model, clips: Sequence[torch.Tensor], labels: Sequence[int], fs: float,
                  list_N: Iterable[int], hf: float = 0.5, batch_size: int = 8,
                  mode: int = _lib.MODE_F32, json_file: Optional[str] = None,
                  list_Fs: Optional[Iterable[float]] = None) -> Dict:
    """Accuracy of ``model`` for every analysis length in ``list_N`` - and, with ``list_Fs``, for every
    sampling rate the clips (recorded at ``fs``) are resampled to first, as the double loop of
    ``Code/pceval.py:61-98`` does; returns (and optionally writes) the dictionary
    ``Code/pceval.py:57-59,99-104`` stores: ``{"data": {Fs: [acc per N]}, "list_Fs": [...],
    "list_N": [...]}``."""
    list_N = [int(n) for n in list_N]
    if list_Fs is not None:
        list_Fs = list(list_Fs)
        data = {}
        for F in list_Fs:
            rs = [pca_hip.resample(x, fs, F, scale=True) for x in clips]     # pceval.py:74
            data[F] = reframe_sweep(model, rs, labels, F, list_N, hf, batch_size, mode)["data"][F]
        out = {"data": data, "list_Fs": list_Fs, "list_N": list_N}
        if json_file is not None:
            with open(json_file, "w") as f:
                json.dump(out, f)
        return out
    accs: List[float] = []
    for N in list_N:
        ds = framewise_dataset(clips, labels, fs, N, hf)
        n = len(ds)
        full = (n // batch_size) * batch_size          # pceval.py:88-89 skips the short batch
        if full == 0:
            accs.append(float("nan"))
            continue
        # one engine launch covers many of the reference's 8-set batches
        chunk = batch_size * max(1, 256 // batch_size)
        dev = ds._resident()[0].device
        correct, done = 0, 0
        eng, eng_b = None, 0
        while done < full:
            b = min(chunk, full - done)
            if b != eng_b:
                eng, eng_b = STEngine(model, b, ds.num_points, mode, training=False), b
            X, lab = ds.batch(torch.arange(done, done + b, device=dev))
            correct += int((eng.forward(X).argmax(1) == lab).sum())
            done += b
        accs.append(correct / full)
    out = {"data": {fs: accs}, "list_Fs": [fs], "list_N": list_N}
    if json_file is not None:
        with open(json_file, "w") as f:
            json.dump(out, f)
    return out
