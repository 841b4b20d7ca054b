"""torch.autograd glue over the C ABI: device pointers and the current HIP stream are the
only things that cross the boundary (no torch types in the library).

PyTorch is plumbing here -- it owns device memory and streams; all arithmetic of the hot
path runs in libpca_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import math

import torch

from . import _lib
from ._lib import MabGrads, MabParams, MabShape, check, lib

_MODE = "f32"           # "f32" | "bf16" | "fp8" | "auto"


def set_mode(mode: str) -> None:
    """Arithmetic mode of MAB blocks: 'f32' = exact fp32 kernels (parity mode),
    'bf16' = bf16 MFMA operands with fp32 accumulate/softmax (fails for shapes the fused
    kernels do not cover), 'fp8' = as bf16 with fp8 (e4m3) operands in fc_o of the many-queries
    blocks and fc_k / fc_v of the d = 256 few-queries block (fc_q stays bf16:
    include/pca_hip.h), 'auto' = bf16 where covered, else f32."""
    global _MODE
    if mode not in ("f32", "bf16", "fp8", "auto"):
        raise ValueError(mode)
    _MODE = mode


def get_mode() -> str:
    return _MODE


def _stream(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _need_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.PcaHipError(
                "point-cloud-audio_amd runs on MI355X only: got a CPU tensor "
                "(move the module and its inputs to 'cuda'; there is no CPU fallback)")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# tests: with CANARY set, every scratch / saved block handed to the library is followed by a
# guard region that check_canaries() verifies (a kernel writing past its workspace)
CANARY = False
_GUARD = 1 << 16
_guards: list = []


def _bytes(n: int, like: torch.Tensor) -> torch.Tensor:
    n = max(int(n), 256)
    if not CANARY:
        return torch.empty(n, dtype=torch.uint8, device=like.device)
    n = (n + 255) // 256 * 256
    full = torch.empty(n + _GUARD, dtype=torch.uint8, device=like.device)
    full[n:] = 0xA5
    _guards.append(full[n:])
    return full[:n]


def check_canaries() -> int:
    """Number of guard regions checked; raises if a library call wrote past a block it was given."""
    torch.cuda.synchronize()
    k = len(_guards)
    for g in _guards:
        if not bool((g == 0xA5).all()):
            _guards.clear()
            raise _lib.PcaHipError("a kernel wrote past the end of its workspace")
    _guards.clear()
    return k


def _shape(B, nq, nk, dq, dk, d, h, q_shared, mode=_lib.MODE_F32, k_lengths=None,
           ln: bool = False) -> MabShape:
    return MabShape(B, nq, nk, dq, dk, d, h, int(q_shared), mode, _lib.PCA_F32,
                    _lib.PCA_F32, _lib.PCA_F32, 0 if k_lengths is None else k_lengths.data_ptr(),
                    int(ln))


def _lengths(key_lengths, B: int, like: torch.Tensor):
    """int32[B] on the device of ``like`` (the library reads it there), or None."""
    if key_lengths is None:
        return None
    kl = torch.as_tensor(key_lengths).to(like.device, torch.int32).contiguous()
    if kl.shape != (B,):
        raise RuntimeError(f"key_lengths must have shape ({B},), got {tuple(kl.shape)}")
    return kl


def _pick_mode(s: MabShape, inference: bool = False) -> MabShape:
    """Resolve the arithmetic mode of one block: 'bf16' demands the fused kernel, 'auto' takes
    it where the library has one for this shape (pca_mab_saved_bytes() > 0; forward-only calls
    ask pca_mab_fwd_ws_bytes(), which also covers the kernels that have no backward yet) else
    exact fp32."""
    if _MODE == "f32":
        return s
    s.mode = _lib.MODE_FP8 if _MODE == "fp8" else _lib.MODE_BF16
    probe = lib().pca_mab_fwd_ws_bytes if inference else lib().pca_mab_saved_bytes
    if _MODE == "auto" and probe(C.byref(s)) == 0:
        s.mode = _lib.MODE_F32
    return s


class _MabFn(torch.autograd.Function):
    """set_transformer-master/modules.py:19-33 MAB.forward + its adjoint."""

    @staticmethod
    def forward(ctx, Q, K, wq, bq, wk, bk, wv, bv, wo, bo, num_heads: int, q_shared: bool,
                key_lengths=None, ln0w=None, ln0b=None, ln1w=None, ln1b=None):
        _need_cuda(Q, K, wq)
        Q, K = _f32c(Q), _f32c(K)
        params = [_f32c(p) for p in (wq, bq, wk, bk, wv, bv, wo, bo)]
        ln = ln0w is not None
        if ln:
            params += [_f32c(p) for p in (ln0w, ln0b, ln1w, ln1b)]
        B, nk, dk = K.shape
        if q_shared:
            nq, dq = Q.shape[-2], Q.shape[-1]
        else:
            if Q.shape[0] != B:
                raise RuntimeError(f"MAB: batch mismatch Q {tuple(Q.shape)} K {tuple(K.shape)}")
            nq, dq = Q.shape[1], Q.shape[2]
        d = params[0].shape[0]
        if params[0].shape[1] != dq or params[2].shape[1] != dk:
            raise RuntimeError("MAB: input width does not match fc_q / fc_k")
        kl = _lengths(key_lengths, B, K)
        s = _pick_mode(_shape(B, nq, nk, dq, dk, d, num_heads, q_shared, k_lengths=kl, ln=ln))
        ctx.kl = kl                       # keeps the device array alive for the backward
        ctx.ln = ln
        L = lib()
        with torch.cuda.device(K.device):
            Y = torch.empty((B, nq, d), dtype=torch.float32, device=K.device)
            nsaved = L.pca_mab_saved_bytes(C.byref(s))
            if nsaved == 0:
                raise _lib.PcaHipError("pca_mab_saved_bytes: " + L.pca_last_error().decode())
            saved = _bytes(nsaved, K)
            ws = _bytes(L.pca_mab_fwd_ws_bytes(C.byref(s)), K)
            pp = MabParams(*[_ptr(p) for p in params])      # (ln pointers stay NULL without ln)
            check(L.pca_mab_fwd(C.byref(s), _ptr(Q), _ptr(K), C.byref(pp), _ptr(Y),
                                _ptr(saved), _ptr(ws), _stream(K)), "pca_mab_fwd")
        ctx.s = s
        ctx.save_for_backward(Q, K, saved, *params)
        return Y

    @staticmethod
    def backward(ctx, dY):
        Q, K, saved, *params = ctx.saved_tensors
        s = ctx.s
        L = lib()
        dY = _f32c(dY)
        need_dq, need_dk = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        with torch.cuda.device(K.device):
            sizes = [p.numel() for p in params]
            flat = torch.zeros(sum(sizes), dtype=torch.float32, device=K.device)
            gviews = list(torch.split(flat, sizes))
            gg = MabGrads(*[_ptr(g) for g in gviews])
            pp = MabParams(*[_ptr(p) for p in params])
            dQ = dK = None
            if need_dq:
                dQ = (torch.zeros_like(Q) if s.q_shared else torch.empty_like(Q))
            if need_dk:
                dK = torch.empty_like(K)
            ws = _bytes(L.pca_mab_bwd_ws_bytes(C.byref(s)), K)
            check(L.pca_mab_bwd(C.byref(s), _ptr(Q), _ptr(K), C.byref(pp), _ptr(saved),
                                _ptr(dY), _ptr(dQ), _ptr(dK), 0, C.byref(gg), _ptr(ws),
                                _stream(K)), "pca_mab_bwd")
        grads = [g.view_as(p) for g, p in zip(gviews, params)]
        if ctx.ln:
            return (dQ, dK, *grads[:8], None, None, None, *grads[8:])
        return (dQ, dK, *grads, None, None, None, None, None, None, None)


def mab(Q, K, wq, bq, wk, bk, wv, bv, wo, bo, num_heads: int, q_shared: bool = False,
        key_lengths=None, ln_params=None):
    """key_lengths (optional, int[B]): valid keys per set of a padded batch.
    ln_params (optional): (ln0.weight, ln0.bias, ln1.weight, ln1.bias) of MAB(ln=True)."""
    extra = tuple(ln_params) if ln_params is not None else (None, None, None, None)
    return _MabFn.apply(Q, K, wq, bq, wk, bk, wv, bv, wo, bo, num_heads, q_shared, key_lengths,
                        *extra)


def mab_infer(Q, K, params, num_heads: int, q_shared: bool = False,
              key_lengths=None, ln_params=None) -> torch.Tensor:
    """Forward only, nothing saved (used under torch.no_grad())."""
    _need_cuda(Q, K)
    Q, K = _f32c(Q), _f32c(K)
    params = [_f32c(p) for p in params]
    ln = ln_params is not None
    if ln:
        params += [_f32c(p) for p in ln_params]
    B, nk, dk = K.shape
    nq, dq = Q.shape[-2], Q.shape[-1]
    d = params[0].shape[0]
    kl = _lengths(key_lengths, B, K)
    s = _pick_mode(_shape(B, nq, nk, dq, dk, d, num_heads, q_shared, k_lengths=kl, ln=ln),
                   inference=True)
    L = lib()
    with torch.cuda.device(K.device):
        Y = torch.empty((B, nq, d), dtype=torch.float32, device=K.device)
        ws = _bytes(L.pca_mab_fwd_ws_bytes(C.byref(s)), K)
        pp = MabParams(*[_ptr(p) for p in params])
        check(L.pca_mab_fwd(C.byref(s), _ptr(Q), _ptr(K), C.byref(pp), _ptr(Y), None,
                            _ptr(ws), _stream(K)), "pca_mab_fwd")
    return Y


class _LinearFn(torch.autograd.Function):
    """nn.Linear (Code/models.py:40) on the library's GEMM."""

    @staticmethod
    def forward(ctx, X, W, b):
        _need_cuda(X, W, b)
        X, W, b = _f32c(X), _f32c(W), _f32c(b)
        lead = X.shape[:-1]
        M = X.numel() // X.shape[-1]
        with torch.cuda.device(X.device):
            Y = torch.empty((*lead, W.shape[0]), dtype=torch.float32, device=X.device)
            check(lib().pca_linear_fwd(_ptr(X), _ptr(W), _ptr(b), _ptr(Y), M, W.shape[1],
                                       W.shape[0], _stream(X)), "pca_linear_fwd")
        ctx.save_for_backward(X, W)
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, W = ctx.saved_tensors
        dY = _f32c(dY)
        M = X.numel() // X.shape[-1]
        with torch.cuda.device(X.device):
            dX = torch.empty_like(X) if ctx.needs_input_grad[0] else None
            dW = torch.zeros_like(W)
            db = torch.zeros(W.shape[0], dtype=torch.float32, device=X.device)
            check(lib().pca_linear_bwd(_ptr(X), _ptr(W), _ptr(dY), _ptr(dX), _ptr(dW),
                                       _ptr(db), M, W.shape[1], W.shape[0], None,
                                       _stream(X)), "pca_linear_bwd")
        return dX, dW, db


def linear(X, W, b):
    return _LinearFn.apply(X, W, b)


class _CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean), Code/settransformer.py:88,104."""

    @staticmethod
    def forward(ctx, logits, labels):
        _need_cuda(logits, labels)
        logits = _f32c(logits)
        labels = labels.to(torch.int64).contiguous()
        B, Cc = logits.shape
        with torch.cuda.device(logits.device):
            loss = torch.empty(1, dtype=torch.float32, device=logits.device)
            dlog = torch.empty_like(logits)
            check(lib().pca_cross_entropy(_ptr(logits), _ptr(labels), B, Cc, 1.0, _ptr(loss),
                                          _ptr(dlog), None, _stream(logits)),
                  "pca_cross_entropy")
        ctx.save_for_backward(dlog)
        return loss.squeeze(0)

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        return dlog * g, None


def cross_entropy(logits, labels):
    return _CrossEntropyFn.apply(logits, labels)


# --------------------------------------------------------------------------- #
# feature extraction                                                           #
# --------------------------------------------------------------------------- #
def stft_logmag(wave: torch.Tensor, n_fft: int, win_length: Optional[int] = None,
                hop: Optional[int] = None, drop_nyquist: bool = False,
                frame_major: bool = False) -> torch.Tensor:
    """log(1e-8 + |stft|/n_fft) of a 1-D float32 device waveform.
    Returns [F, T] (reference layout) or, with frame_major, [T, F] (each frame -- one
    2-D point set -- contiguous, the layout the packers read coalesced)."""
    _need_cuda(wave)
    wave = _f32c(wave)
    win_length = n_fft if win_length is None else win_length
    hop = n_fft // 2 if hop is None else hop
    L = lib()
    T = L.pca_stft_num_frames(wave.numel(), hop)
    F = n_fft // 2 if drop_nyquist else n_fft // 2 + 1
    with torch.cuda.device(wave.device):
        if frame_major:
            out = torch.empty((T, F), dtype=torch.float32, device=wave.device)
            sf, st = 1, F
        else:
            out = torch.empty((F, T), dtype=torch.float32, device=wave.device)
            sf, st = T, 1
        check(L.pca_stft_logmag(_ptr(wave), wave.numel(), n_fft, win_length, hop, F,
                                _ptr(out), sf, st, _stream(wave)), "pca_stft_logmag")
    return out


# ---- resampling (the sampling-rate axis of the evaluation sweep, Code/pceval.py:74) ----------------
# librosa.resample(.., res_type='kaiser_fast', scale=True) = resampy's band-limited interpolation; the
# filter design below is resampy's documented kaiser_fast (16 zero crossings, 512 table entries per
# zero crossing, roll-off 0.85, Kaiser beta 8.5555): "parity unpinned" (neither package is available).
KAISER_FAST = dict(num_zeros=16, precision=9, rolloff=0.85, beta=8.555504641634386)
_resample_tables = {}


def _resample_filter(ratio: float, device, design):
    import numpy as np
    key = (round(ratio, 12), str(device), tuple(sorted(design.items())))
    if key not in _resample_tables:
        num_table = 2 ** design["precision"]
        n = num_table * design["num_zeros"]
        win = design["rolloff"] * np.sinc(design["rolloff"] * np.linspace(0, design["num_zeros"], n + 1))
        win = win * np.kaiser(2 * n + 1, design["beta"])[n:]
        if ratio < 1:
            win = win * ratio
        delta = np.zeros_like(win)
        delta[:-1] = np.diff(win)
        _resample_tables[key] = (torch.from_numpy(win).to(device), torch.from_numpy(delta).to(device),
                                 num_table)
    return _resample_tables[key]


def resample(wave: torch.Tensor, fs_old: float, fs_new: float, scale: bool = True,
             **design) -> torch.Tensor:
    """1-D float32 device waveform at fs_old -> ceil(n * fs_new / fs_old) samples at fs_new
    (librosa.resample(x, fs_old, fs_new, res_type='kaiser_fast', fix=True, scale=scale))."""
    _need_cuda(wave)
    wave = _f32c(wave)
    ratio = float(fs_new) / float(fs_old)
    if ratio == 1.0:
        return wave.clone()
    d = dict(KAISER_FAST)
    d.update(design)
    win, delta, num_table = _resample_filter(ratio, wave.device, d)
    n_in = wave.numel()
    n_res = int(n_in * ratio)
    n_out = int(math.ceil(n_in * ratio))
    with torch.cuda.device(wave.device):
        out = torch.zeros(n_out, dtype=torch.float32, device=wave.device)
        if n_res > 0:
            check(lib().pca_resample(_ptr(wave), n_in, ratio, _ptr(win), _ptr(delta), win.numel(),
                                     num_table, (1.0 / math.sqrt(ratio)) if scale else 1.0, _ptr(out),
                                     min(n_res, n_out), _stream(wave)), "pca_resample")
    return out


def stft_logmag_batch(waves, n_fft: int, win_length: Optional[int] = None,
                      hop: Optional[int] = None, drop_nyquist: bool = False,
                      frame_major: bool = False):
    """log(1e-8 + |stft|/n_fft) of a list of 1-D float32 device waveforms in ONE launch.
    Returns (spec, frame_off): spec is [F, T_total] (or [T_total, F] with frame_major), clip c
    occupies columns (rows) frame_off[c] : frame_off[c + 1]; each clip's block is bit-identical
    to stft_logmag(clip)."""
    assert len(waves) > 0
    _need_cuda(*waves)
    dev = waves[0].device
    win_length = n_fft if win_length is None else win_length
    hop = n_fft // 2 if hop is None else hop
    lens = [int(w.numel()) for w in waves]
    frames = [1 + n // hop for n in lens]
    woff = [0]
    foff = [0]
    for n, t in zip(lens, frames):
        woff.append(woff[-1] + n)
        foff.append(foff[-1] + t)
    T = foff[-1]
    F = n_fft // 2 if drop_nyquist else n_fft // 2 + 1
    L = lib()
    with torch.cuda.device(dev):
        cat = torch.cat([_f32c(w).reshape(-1) for w in waves])
        woff_d = torch.tensor(woff, dtype=torch.int64, device=dev)
        foff_d = torch.tensor(foff, dtype=torch.int64, device=dev)
        if frame_major:
            out = torch.empty((T, F), dtype=torch.float32, device=dev)
            sf, st = 1, F
        else:
            out = torch.empty((F, T), dtype=torch.float32, device=dev)
            sf, st = T, 1
        check(L.pca_stft_logmag_batch(_ptr(cat), _ptr(woff_d), _ptr(foff_d), len(waves),
                                      max(lens), min(lens), n_fft, win_length, hop, F, _ptr(out),
                                      sf, st, _stream(cat)), "pca_stft_logmag_batch")
    return out, foff


def pack_points_2d(spec: torch.Tensor, farr: torch.Tensor, idx: torch.Tensor,
                   labels: Optional[torch.Tensor] = None, frame_major: bool = False,
                   out: Optional[torch.Tensor] = None, labels_out: Optional[torch.Tensor] = None
                   ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Batch of ESC_pc items: spec [F,T] (or [T,F] frame_major), idx int64[B] ->
    ([B,F,2] float32, labels[idx])."""
    _need_cuda(spec, farr, idx)
    assert spec.dtype == torch.float32 and farr.dtype == torch.float32
    assert idx.dtype == torch.int64
    if frame_major:
        T, F = spec.shape
        sf, st = spec.stride(1), spec.stride(0)
    else:
        F, T = spec.shape
        sf, st = spec.stride(0), spec.stride(1)
    B = idx.numel()
    with torch.cuda.device(spec.device):
        if out is None:
            out = torch.empty((B, F, 2), dtype=torch.float32, device=spec.device)
        if labels is not None and labels_out is None:
            labels_out = torch.empty(B, dtype=torch.int64, device=spec.device)
        check(lib().pca_pack_points_2d(_ptr(spec), sf, st, _ptr(farr), _ptr(idx), B, F,
                                       _ptr(out), _ptr(labels), _ptr(labels_out),
                                       _stream(spec)), "pca_pack_points_2d")
    return out, labels_out


def pack_points_3d(spec: torch.Tensor, farr: torch.Tensor, tarr: torch.Tensor,
                   idx: torch.Tensor, labels: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None,
                   labels_out: Optional[torch.Tensor] = None,
                   nt_valid: Optional[torch.Tensor] = None,
                   lengths_out: Optional[torch.Tensor] = None):
    """Batch of ESC_pc_temp items: spec indexed [f, t, s] through its strides (any
    layout), idx int64[B] -> ([B, Nt*F, 3] float32, labels[idx]).
    With ``nt_valid`` (device int32[S], frames held by each chunk) the batch is padded:
    returns (points, labels[idx], lengths int32[B]) (pca_pack_points_3d_var)."""
    _need_cuda(spec, farr, tarr, idx)
    assert spec.dtype == torch.float32 and idx.dtype == torch.int64
    F, Nt, S = spec.shape
    B = idx.numel()
    with torch.cuda.device(spec.device):
        if out is None:
            out = torch.empty((B, F * Nt, 3), dtype=torch.float32, device=spec.device)
        if labels is not None and labels_out is None:
            labels_out = torch.empty(B, dtype=torch.int64, device=spec.device)
        if nt_valid is not None:
            _need_cuda(nt_valid)
            assert nt_valid.dtype == torch.int32 and nt_valid.numel() == S
            if lengths_out is None:
                lengths_out = torch.empty(B, dtype=torch.int32, device=spec.device)
            check(lib().pca_pack_points_3d_var(
                _ptr(spec), spec.stride(0), spec.stride(1), spec.stride(2), _ptr(farr),
                _ptr(tarr), _ptr(nt_valid), _ptr(idx), B, F, Nt, _ptr(out), _ptr(lengths_out),
                _ptr(labels), _ptr(labels_out), _stream(spec)), "pca_pack_points_3d_var")
            return out, labels_out, lengths_out
        check(lib().pca_pack_points_3d(_ptr(spec), spec.stride(0), spec.stride(1),
                                       spec.stride(2), _ptr(farr), _ptr(tarr), _ptr(idx), B,
                                       F, Nt, _ptr(out), _ptr(labels), _ptr(labels_out),
                                       _stream(spec)), "pca_pack_points_3d")
    return out, labels_out


MAXK, RANDK = 0, 1


def _draw_ptr(draw_dev):
    if draw_dev is None:
        return None
    assert draw_dev.is_cuda and draw_dev.dtype == torch.int32 and draw_dev.numel() >= 1
    return draw_dev.data_ptr()


def subsample_points(spec: torch.Tensor, farr: torch.Tensor, tarr: Optional[torch.Tensor],
                     idx: torch.Tensor, K: int, mode: int = MAXK, seed: int = 0,
                     draw: int = 0, labels: Optional[torch.Tensor] = None,
                     out: Optional[torch.Tensor] = None,
                     labels_out: Optional[torch.Tensor] = None, want_sel: bool = False,
                     draw_dev: Optional[torch.Tensor] = None):
    """Batch of sub-sampled point sets selected on the device (pca_subsample_points).
    ``draw_dev`` (device int32, optional): its first element is added to ``draw`` on the device,
    so a launch captured into a hipGraph draws a fresh selection on every replay.

    spec indexed [f, t, s] through its strides ([F, T] with tarr=None for the framewise
    2-D case); idx int64[B].  mode MAXK: the K largest values per set, descending
    (Code/dataset.py:196, Code/utils.py:42); RANDK: K points of a random permutation
    (Code/dataset.py:236, Code/utils.py:70) from the stream (seed, draw).
    Returns (points [B, K, din] float32, labels[idx] or None[, sel int32 [B, K]])."""
    _need_cuda(spec, farr, idx)
    assert spec.dtype == torch.float32 and idx.dtype == torch.int64
    if tarr is None:
        assert spec.dim() == 2
        F, Nt = spec.shape[0], 1
        sf, st, ss = spec.stride(0), 0, spec.stride(1)
        din = 2
    else:
        _need_cuda(tarr)
        F, Nt, _ = spec.shape
        sf, st, ss = spec.stride(0), spec.stride(1), spec.stride(2)
        din = 3
    B = idx.numel()
    with torch.cuda.device(spec.device):
        if out is None:
            out = torch.empty((B, K, din), dtype=torch.float32, device=spec.device)
        if labels is not None and labels_out is None:
            labels_out = torch.empty(B, dtype=torch.int64, device=spec.device)
        sel = torch.empty((B, K), dtype=torch.int32, device=spec.device) if want_sel else None
        check(lib().pca_subsample_points(_ptr(spec), sf, st, ss, _ptr(farr), _ptr(tarr),
                                         _ptr(idx), B, F, Nt, int(K), int(mode),
                                         int(seed) & (2 ** 64 - 1), int(draw) & (2 ** 64 - 1),
                                         _draw_ptr(draw_dev), _ptr(out), _ptr(sel),
                                         _ptr(labels), _ptr(labels_out), _stream(spec)),
              "pca_subsample_points")
    return (out, labels_out, sel) if want_sel else (out, labels_out)


def pack_points_2d_ss(x_tk: torch.Tensor, f_tk: torch.Tensor, idx: torch.Tensor,
                      labels: Optional[torch.Tensor] = None,
                      out: Optional[torch.Tensor] = None,
                      labels_out: Optional[torch.Tensor] = None
                      ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Batch of ESC_pc_ss items from frame-major tables x_tk, f_tk [T, K] (contiguous):
    ([B, K, 2] float32, labels[idx])."""
    _need_cuda(x_tk, f_tk, idx)
    assert x_tk.dtype == torch.float32 and f_tk.dtype == torch.float32
    assert x_tk.is_contiguous() and f_tk.is_contiguous() and x_tk.shape == f_tk.shape
    K = x_tk.shape[1]
    B = idx.numel()
    with torch.cuda.device(x_tk.device):
        if out is None:
            out = torch.empty((B, K, 2), dtype=torch.float32, device=x_tk.device)
        if labels is not None and labels_out is None:
            labels_out = torch.empty(B, dtype=torch.int64, device=x_tk.device)
        check(lib().pca_pack_points_2d_ss(_ptr(x_tk), _ptr(f_tk), _ptr(idx), B, K, _ptr(out),
                                          _ptr(labels), _ptr(labels_out), _stream(x_tk)),
              "pca_pack_points_2d_ss")
    return out, labels_out


def importance_kernel(winF: int) -> torch.Tensor:
    """The [2, winF] smoothing kernel of Code/dataset.py:283: outer product of two periodic
    Kaiser windows (beta 5.09), built with the same torch calls so that the weights are
    bit-identical to the reference's."""
    return (torch.kaiser_window(window_length=2, periodic=True, beta=5.09)[:, None]
            @ torch.kaiser_window(window_length=int(winF), periodic=True, beta=5.09)[None, :]
            ).contiguous()


def importance_points(spec: torch.Tensor, farr: torch.Tensor, tarr: torch.Tensor,
                      idx: torch.Tensor, K: int, choice: int, kern: torch.Tensor,
                      seed: int = 0, draw: int = 0, labels: Optional[torch.Tensor] = None,
                      out: Optional[torch.Tensor] = None,
                      labels_out: Optional[torch.Tensor] = None, want_sel: bool = False,
                      want_heat: bool = False, draw_dev: Optional[torch.Tensor] = None):
    """Batch of ESC_pc_temp_importancerandKSS items (pca_importance_points): spec [F, Nt, S]
    through its strides, kern [2, winF] float32 on the device.  Returns
    (points [B, K, 3], labels[idx] or None[, sel int32 [B, K]][, heat [B, F, Nt]])."""
    _need_cuda(spec, farr, tarr, idx, kern)
    assert spec.dtype == torch.float32 and idx.dtype == torch.int64
    assert kern.dtype == torch.float32 and kern.is_contiguous() and kern.shape[0] == 2
    F, Nt, _ = spec.shape
    B = idx.numel()
    with torch.cuda.device(spec.device):
        if out is None:
            out = torch.empty((B, K, 3), dtype=torch.float32, device=spec.device)
        if labels is not None and labels_out is None:
            labels_out = torch.empty(B, dtype=torch.int64, device=spec.device)
        sel = torch.empty((B, K), dtype=torch.int32, device=spec.device) if want_sel else None
        heat = torch.empty((B, F, Nt), dtype=torch.float32, device=spec.device) \
            if want_heat else None
        check(lib().pca_importance_points(
            _ptr(spec), spec.stride(0), spec.stride(1), spec.stride(2), _ptr(farr), _ptr(tarr),
            _ptr(idx), B, F, Nt, int(K), int(choice), _ptr(kern), int(kern.shape[1]),
            int(seed) & (2 ** 64 - 1), int(draw) & (2 ** 64 - 1), _draw_ptr(draw_dev), _ptr(out),
            _ptr(sel), _ptr(heat), _ptr(labels), _ptr(labels_out), _stream(spec)), "pca_importance_points")
    res = [out, labels_out]
    if want_sel:
        res.append(sel)
    if want_heat:
        res.append(heat)
    return tuple(res)


def pack_points_2d_seq(spec: torch.Tensor, farr: torch.Tensor, idx_seq: torch.Tensor,
                       step_dev: torch.Tensor, base_dev: torch.Tensor, B: int,
                       labels: Optional[torch.Tensor] = None, frame_major: bool = False,
                       out: Optional[torch.Tensor] = None,
                       labels_out: Optional[torch.Tensor] = None):
    """pack_points_2d for batch number ``step_dev[0] - base_dev[0]`` of the pre-staged index
    sequence ``idx_seq`` ([n_steps * B] int64 on the device): the cursor lives on the device, so
    a captured step replays without any per-step index upload (pca_pack_points_2d_seq)."""
    _need_cuda(spec, farr, idx_seq, step_dev, base_dev)
    assert spec.dtype == torch.float32 and idx_seq.dtype == torch.int64
    assert step_dev.dtype == torch.int32 and base_dev.dtype == torch.int32
    if frame_major:
        T_, F = spec.shape
        sf, st = spec.stride(1), spec.stride(0)
    else:
        F, T_ = spec.shape
        sf, st = spec.stride(0), spec.stride(1)
    with torch.cuda.device(spec.device):
        if out is None:
            out = torch.empty((B, F, 2), dtype=torch.float32, device=spec.device)
        if labels is not None and labels_out is None:
            labels_out = torch.empty(B, dtype=torch.int64, device=spec.device)
        check(lib().pca_pack_points_2d_seq(_ptr(spec), sf, st, _ptr(farr), _ptr(idx_seq),
                                           _ptr(step_dev), _ptr(base_dev), B, F, _ptr(out),
                                           _ptr(labels), _ptr(labels_out), _stream(spec)),
              "pca_pack_points_2d_seq")
    return out, labels_out


def pack_points_3d_seq(spec: torch.Tensor, farr: torch.Tensor, tarr: torch.Tensor,
                       idx_seq: torch.Tensor, step_dev: torch.Tensor, base_dev: torch.Tensor,
                       B: int, labels: Optional[torch.Tensor] = None,
                       out: Optional[torch.Tensor] = None,
                       labels_out: Optional[torch.Tensor] = None,
                       nt_valid: Optional[torch.Tensor] = None,
                       lengths_out: Optional[torch.Tensor] = None):
    """3-D counterpart of pack_points_2d_seq (pca_pack_points_3d_seq); with ``nt_valid`` the
    batch is padded and the lengths are returned as a third value."""
    _need_cuda(spec, farr, tarr, idx_seq, step_dev, base_dev)
    assert spec.dtype == torch.float32 and idx_seq.dtype == torch.int64
    F, Nt, S = spec.shape
    with torch.cuda.device(spec.device):
        if out is None:
            out = torch.empty((B, F * Nt, 3), dtype=torch.float32, device=spec.device)
        if labels is not None and labels_out is None:
            labels_out = torch.empty(B, dtype=torch.int64, device=spec.device)
        if nt_valid is not None and lengths_out is None:
            lengths_out = torch.empty(B, dtype=torch.int32, device=spec.device)
        check(lib().pca_pack_points_3d_seq(
            _ptr(spec), spec.stride(0), spec.stride(1), spec.stride(2), _ptr(farr), _ptr(tarr),
            _ptr(nt_valid), _ptr(idx_seq), _ptr(step_dev), _ptr(base_dev), B, F, Nt, _ptr(out),
            _ptr(lengths_out if nt_valid is not None else None), _ptr(labels), _ptr(labels_out),
            _stream(spec)), "pca_pack_points_3d_seq")
    if nt_valid is not None:
        return out, labels_out, lengths_out
    return out, labels_out
