"""Train / eval engine for the ST classifier on MI355X.

Replaces the training loop of Code/settransformer.py:96-112 (and settransformertemp.py):
``DataLoader -> .to(device) -> nn.DataParallel(model)(imgs) -> CrossEntropyLoss -> backward ->
Adam.step -> .item()`` becomes, per step and per GPU,

    [pack batch] -> pca_st_train_fwd_bwd -> all-reduce(gradients) -> pca_adam_step

or, with ``overlap`` (large models: the exchange hides under the rest of the backward),

    [pack batch] -> pca_st_train_fwd_bwd(phase 0) -> all-reduce(bucket enc.1+dec)
                 -> pca_st_train_fwd_bwd(phase 1) -> all-reduce(bucket enc.0) -> pca_adam_step

* one process per GPU; parameters, gradients and Adam moments are flat fp32 vectors in
  state_dict order, so the all-reduce is over two contiguous buckets and the optimiser is
  one fused kernel; the nn.Module's parameters are views of the flat vector;
* the device work of a step is captured once into hipGraphs (torch.cuda.CUDAGraph) and
  replayed, so no Python / autograd / allocator work sits between kernels;
* nn.DataParallel (Code/settransformer.py:94: single process, per-step parameter broadcast,
  scatter and gather) is replaced by an RCCL all-reduce of gradients (torch.distributed
  backend 'nccl' over xGMI): one message between the backward and the optimiser, or two
  buckets with the first on a side stream under the enc.0 backward (``overlap``);
* loss / accuracy are accumulated on the device and read once per epoch instead of the
  two ``.item()`` host syncs per step of Code/settransformer.py:110-112.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib
from ._lib import StConfig, check, lib
from .ops import _need_cuda


def st_config(model, B: int, N: int, mode: int = _lib.MODE_F32) -> StConfig:
    """Read the architecture off an ``models.ST`` instance (bare, or wrapped in
    ``nn.DataParallel`` as ``runfiles.load_run`` / Code/pceval.py:46 hand it over)."""
    model = getattr(model, "module", model)
    isab0 = model.enc[0]
    if getattr(isab0.mab0, "ln0", None) is not None:
        raise _lib.PcaHipError("the whole-model engine covers ST(ln=False) (45 tensors); a model "
                               "with LayerNorms runs through the nn.Module path")
    d = isab0.mab0.dim_V
    return StConfig(B, N, isab0.mab1.fc_q.in_features, d, isab0.mab0.num_heads,
                    isab0.I.shape[1], model.dec[0].S.shape[1],
                    model.dec[1].out_features, mode)


def flatten_parameters(model) -> torch.Tensor:
    """Re-home every parameter of ``model`` as a view of ONE flat fp32 vector (state_dict
    order, the layout pca_st_* expects) and return that vector.  Idempotent: a model that
    is already flat keeps its vector (so several engines can share one model)."""
    model = getattr(model, "module", model)
    params = list(model.parameters())
    names = [n for n, _ in model.named_parameters()]
    assert names == list(model.state_dict().keys()), "unexpected parameter order"
    flat = getattr(model, "_pca_flat", None)
    if flat is not None:
        off, ok = 0, True
        for p in params:
            ok = ok and p.data_ptr() == flat.data_ptr() + 4 * off and p.is_contiguous()
            off += p.numel()
        if ok and off == flat.numel():
            return flat
    flat = torch.cat([p.detach().reshape(-1).float() for p in params]).contiguous()
    off = 0
    for p in params:
        n = p.numel()
        p.data = flat[off:off + n].view_as(p)
        off += n
    object.__setattr__(model, "_pca_flat", flat)
    return flat


class ShardedIndexStream:
    """Per-rank stream of set indices with DistributedSampler semantics: one
    seeded permutation per epoch, shared by all ranks; rank r takes elements r::world of it
    (the tail that does not divide evenly is dropped so that every rank runs the same number
    of steps).  Pure index logic: works on any device, tested on CPU with gloo."""

    def __init__(self, n: int, batch: int, rank: int = 0, world: int = 1, seed: int = 0,
                 shuffle: bool = True, device="cpu"):
        self.n, self.B, self.rank, self.world = int(n), int(batch), rank, world
        self.seed, self.shuffle, self.device = seed, shuffle, device
        self.epoch = 0
        self._perm = None
        self._cursor = 0
        if self.n // self.world < self.B:
            raise ValueError(f"dataset of {n} sets is too small for batch {batch} x {world} ranks")

    def per_rank(self) -> int:
        return self.n // self.world

    def next(self) -> torch.Tensor:
        per = self.per_rank()
        if self._perm is None or self._cursor + self.B > per:
            if self.shuffle:
                g = torch.Generator(device="cpu").manual_seed(self.seed + self.epoch)
                perm = torch.randperm(self.n, generator=g)
            else:
                perm = torch.arange(self.n)
            self._perm = perm[self.rank:per * self.world:self.world].contiguous().to(self.device)
            self._cursor = 0
            self.epoch += 1
        out = self._perm[self._cursor:self._cursor + self.B]
        self._cursor += self.B
        return out

    def steps_per_epoch(self) -> int:
        return self.per_rank() // self.B

    def next_epoch(self) -> torch.Tensor:
        """All index batches of this rank's next epoch back to back ([steps_per_epoch * B]) -
        the same sequence ``next()`` would hand out one batch at a time."""
        spe = self.steps_per_epoch()
        first = self.next()
        # ``next`` has just started an epoch (it only does so with the cursor at 0)
        assert self._cursor == self.B
        self._cursor = spe * self.B
        return torch.cat([first, self._perm[self.B:spe * self.B]]) if spe > 1 else first


def allreduce_buckets(grads: torch.Tensor, split: int, group=None, first_stream=None):
    """Sum the flat gradient vector over the ranks in the two buckets of SURVEY.md 8e:
    [split, end) (enc.1 + dec, complete after backward phase 0) and [0, split) (enc.0).
    With ``first_stream`` the first bucket is reduced on that (side) stream so that it overlaps
    the remaining backward; the caller joins the streams before the optimiser."""
    if first_stream is not None:
        with torch.cuda.stream(first_stream):
            dist.all_reduce(grads[split:], group=group)
    else:
        dist.all_reduce(grads[split:], group=group)
    return lambda: dist.all_reduce(grads[:split], group=group)


class STEngine:
    """Forward / train-step of an ``models.ST`` through the pca_st_* entry points."""

    def __init__(self, model, B: int, N: int, mode: int = _lib.MODE_F32, training: bool = True):
        self.model = model
        self.flat = flatten_parameters(model)
        _need_cuda(self.flat)
        self.dev = self.flat.device
        self.cfg = st_config(model, B, N, mode)
        L = lib()
        n = L.pca_st_param_count(C.byref(self.cfg))
        if n != self.flat.numel():
            raise _lib.PcaHipError(f"parameter count mismatch: engine {n}, model "
                                   f"{self.flat.numel()}: {L.pca_last_error()}")
        self.split = int(L.pca_st_bucket_split(C.byref(self.cfg)))
        self.training = training
        with torch.cuda.device(self.dev):
            self.ws = torch.empty(L.pca_st_ws_bytes(C.byref(self.cfg), int(training)),
                                  dtype=torch.uint8, device=self.dev)
            self.logits = torch.empty((B * self.cfg.k, self.cfg.C), dtype=torch.float32,
                                      device=self.dev)
            if training:
                self.grads = torch.zeros_like(self.flat)
                self.loss = torch.zeros(1, dtype=torch.float32, device=self.dev)
                self.stats = torch.zeros(2, dtype=torch.float32, device=self.dev)
        # the set-resident forward's bounded spin-waits (csrc/set128_fwd.hip) count their expiries in a
        # word of the workspace that only the caller clears: zero it now, look at it at every host sync
        self._handoff_word = None
        if training:
            ptr = C.c_void_p()
            check(L.pca_st_handoff_counter(C.byref(self.cfg), self.ws.data_ptr(), C.byref(ptr)),
                  "pca_st_handoff_counter")
            if ptr.value:
                off = ptr.value - self.ws.data_ptr()
                self._handoff_word = self.ws[off:off + 4].view(torch.int32)
                self._handoff_word.zero_()

    def check_handoffs(self) -> None:
        """Raise if a pair hand-off of the set-resident forward ever timed out (one host sync): a
        workgroup whose partner was not scheduled within ~1 s went on with stale data, i.e. every
        result since is garbage - e.g. another process holding compute units of this GPU."""
        if self._handoff_word is not None:
            n = int(self._handoff_word.item())
            if n:
                raise _lib.PcaHipError(
                    f"set-resident forward: {n} pair hand-off(s) timed out - the step's results are "
                    "invalid (is another process using this GPU?); PCA_SET128=0 selects the per-block "
                    "launches, which need no co-resident workgroups")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    @staticmethod
    def _len_ptr(lengths, B):
        if lengths is None:
            return None
        assert lengths.is_cuda and lengths.dtype == torch.int32 and lengths.is_contiguous()
        assert tuple(lengths.shape) == (B,), lengths.shape
        return lengths.data_ptr()

    def forward(self, X: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """lengths (optional, device int32[B]): valid points per set of a padded batch."""
        assert X.is_cuda and X.dtype == torch.float32 and X.is_contiguous()
        assert tuple(X.shape) == (self.cfg.B, self.cfg.N, self.cfg.din), X.shape
        check(lib().pca_st_forward(C.byref(self.cfg), self.flat.data_ptr(), X.data_ptr(),
                                   self._len_ptr(lengths, self.cfg.B), self.logits.data_ptr(),
                                   self.ws.data_ptr(), self._stream()),
              "pca_st_forward")
        return self.logits

    def fwd_bwd(self, X: torch.Tensor, labels: torch.Tensor, phase: int = -1,
                grad_scale: float = 1.0, lengths: Optional[torch.Tensor] = None) -> None:
        check(lib().pca_st_train_fwd_bwd(C.byref(self.cfg), self.flat.data_ptr(), X.data_ptr(),
                                         self._len_ptr(lengths, self.cfg.B),
                                         labels.data_ptr(), self.grads.data_ptr(),
                                         self.loss.data_ptr(), self.stats.data_ptr(),
                                         self.logits.data_ptr(), grad_scale, phase,
                                         self.ws.data_ptr(), self._stream()),
              "pca_st_train_fwd_bwd")


class Trainer:
    """Data-parallel trainer: one instance per process / GPU.

    dataset   object with ``batch(idx, out=, labels_out=)`` (dataset.ESC_pc / ESC_pc_temp)
    The step consumes ``batch_size`` sets per GPU; indices come from a per-epoch device
    permutation (rank r takes r::world, DistributedSampler semantics).
    """

    def __init__(self, model, dataset, batch_size: int, lr: float = 1e-3,
                 weight_decay: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 mode: int = _lib.MODE_F32, use_graph: bool = True, seed: int = 0,
                 shuffle: bool = True, process_group=None, keep_grads: bool = False,
                 overlap: Optional[bool] = None):
        self.ds = dataset
        self.keep_grads = keep_grads    # True: gradients of the last step stay readable
        self.B = int(batch_size)
        self.N = int(dataset.num_points)
        self.eng = STEngine(model, self.B, self.N, mode, training=True)
        self.dev = self.eng.dev
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.seed = seed
        self.shuffle = shuffle
        self.use_graph = use_graph
        n = self.eng.flat.numel()
        with torch.cuda.device(self.dev):
            self.m = torch.zeros(n, dtype=torch.float32, device=self.dev)
            self.v = torch.zeros(n, dtype=torch.float32, device=self.dev)
            self.step_count = torch.zeros(2, dtype=torch.int32, device=self.dev)  # [count, ticket]
            self.idx = torch.zeros(self.B, dtype=torch.int64, device=self.dev)
            self.X = torch.empty((self.B, self.N, self.eng.cfg.din), dtype=torch.float32,
                                 device=self.dev)
            self.labels = torch.zeros(self.B, dtype=torch.int64, device=self.dev)
            # padded batches of variable-size sets (dataset.variable_length): point counts
            self.lengths = (torch.zeros(self.B, dtype=torch.int32, device=self.dev)
                            if getattr(dataset, "variable_length", False) else None)
            # device cursor: the epoch's index batches live in ``seq``; the pack kernel takes
            # batch (step_count - epoch_base), so a step is a bare graph replay
            self.epoch_base = torch.zeros(1, dtype=torch.int32, device=self.dev)
            self.seq = None
            # Several GPUs, two ways to exchange the gradients:
            #   overlap=True : the step is split at the bucket boundary; bucket A (enc.1 + dec) is
            #                  reduced on a side stream under enc.0's backward, bucket B after it
            #   overlap=False: one all-reduce of the whole vector between backward and Adam
            # The split costs three graph launches instead of one and ends the launch merging of
            # the backward at the boundary: measured +82 us per step at cfg2 on one GPU
            # (PCA_FORCE_SPLIT=1: 0.370 -> 0.452 ms), more than the all-reduce of 1.2 MB it can
            # hide.  Default: overlap only from 16 MB of gradients on.
            if overlap is None:
                overlap = self.eng.flat.numel() * 4 >= (16 << 20)
            self._split = (self.world > 1 and bool(overlap)) or \
                os.environ.get("PCA_FORCE_SPLIT") == "1"
            # (PCA_FORCE_SPLIT=2: the overlap=False step shape on one GPU, minus the all-reduce)
            self._exchange = self.world > 1 or os.environ.get("PCA_FORCE_SPLIT") == "2"
            # third form (set_exchange("captured")): the all-reduce is captured INSIDE the step's one
            # graph, [pack .. backward | all-reduce | Adam] = one graph launch per step
            self._captured = False
            self.comm_stream = torch.cuda.Stream(self.dev) if self._split else None
        self._cursor_mode = callable(getattr(dataset, "batch_seq", None))
        self._k = 0                       # optimiser steps issued so far (host copy)
        self.g0 = self.g1 = self.g2 = None
        self.indices = ShardedIndexStream(len(dataset), self.B, self.rank, self.world, seed,
                                          shuffle, self.dev)
        if self.world > 1:      # identical initial weights on every rank (rank 0's)
            dist.broadcast(self.eng.flat, src=0, group=self.pg)

    # ---- how the gradients are exchanged ---------------------------------------------
    def set_exchange(self, form: str) -> None:
        """Choose the step shape of a multi-rank run and drop the captured graphs (the next step
        re-captures).  "serial": [graph: pack .. backward] -> all-reduce of the whole vector -> Adam;
        "overlap": the step split at the bucket boundary, bucket A reduced on a side stream under
        enc.0's backward; "captured": ONE graph per step with the all-reduce recorded inside it
        (needs a backend whose collectives can be stream-captured: RCCL / nccl);
        "captured_overlap": ONE graph per step as well, split at the bucket boundary INSIDE it - bucket A's
        all-reduce forks onto a side stream under enc.0's backward and joins before Adam (the overlap of
        "overlap" without its three graph launches and cross-stream events per step, which cost 68 us).
        bench.py times all of them on the first windows of a multi-GPU run and keeps the fastest."""
        if form not in ("serial", "overlap", "captured", "captured_overlap"):
            raise ValueError(form)
        torch.cuda.synchronize(self.dev)
        self.g0 = self.g1 = self.g2 = None
        multi = self.world > 1 or (dist.is_initialized() and os.environ.get("PCA_EXCHANGE_WORLD1") == "1")
        self._split = form in ("overlap", "captured_overlap") and multi
        self._captured = form in ("captured", "captured_overlap") and multi
        self._exchange = multi
        if self._split and self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(self.dev)
        self.exchange_form = form

    def _allreduce_all(self):
        if self.world > 1 or self._captured:
            dist.all_reduce(self.eng.grads, group=self.pg)

    def _captured_body(self):
        """[pack .. backward | all-reduce | Adam] as it is recorded into (or, without graphs, run as) one
        step.  Split form: bucket A (enc.1 + dec, final after phase 0) is reduced on the side stream
        while enc.0's backward runs on the main one; both join before Adam."""
        self._seg0()
        if not self._split:
            self._seg1()
            self._allreduce_all()
        else:
            main = torch.cuda.current_stream(self.dev)
            self.comm_stream.wait_stream(main)                    # fork
            second = allreduce_buckets(self.eng.grads, self.eng.split, self.pg, self.comm_stream)
            self._seg1()
            second()                                              # bucket B on the main stream
            main.wait_stream(self.comm_stream)                    # join
        self._seg2()

    # ---- index stream ------------------------------------------------------------
    def _next_indices(self) -> torch.Tensor:
        return self.indices.next()

    # ---- the three device segments of a step ---------------------------------------
    def _seg0(self):     # pack + zero grads + forward + loss + backward(dec, enc.1)
        if self._cursor_mode:
            kw = dict(lengths_out=self.lengths) if self.lengths is not None else {}
            # the pack rides in the engine's first launch (pca_pack_defer): one launch less per step
            defer = os.environ.get("PCA_PACK_DEFER", "1") != "0"
            if defer:
                check(lib().pca_pack_defer(1), "pca_pack_defer")
            try:
                self.ds.batch_seq(self.seq, self.step_count, self.epoch_base, self.B, out=self.X,
                                  labels_out=self.labels, **kw)
                if self.keep_grads:
                    self.eng.grads.zero_()
                self.eng.fwd_bwd(self.X, self.labels, phase=0 if self._split else -1,
                                 lengths=self.lengths)
            except BaseException:
                if defer:            # disarm (drops a pending job) without masking the error in flight
                    lib().pca_pack_defer(0)
                raise
            if defer:
                check(lib().pca_pack_defer(0), "pca_pack_defer")
            return
        elif self.lengths is not None:
            self.ds.batch(self.idx, out=self.X, labels_out=self.labels,
                          lengths_out=self.lengths)
        elif getattr(self.ds, "stochastic", False):
            # random sub-sampling datasets: the draw number must advance per REPLAY of a captured
            # step, so it is read on the device from the optimiser's step counter
            self.ds.batch(self.idx, out=self.X, labels_out=self.labels,
                          draw_dev=self.step_count)
        else:
            self.ds.batch(self.idx, out=self.X, labels_out=self.labels)
        if self.keep_grads:           # otherwise the Adam pass leaves them cleared
            self.eng.grads.zero_()
        # one GPU: the whole backward in one call (the shared-query gradient kernels of all three
        # blocks then share one pair of launches); several GPUs: stop at the bucket boundary
        self.eng.fwd_bwd(self.X, self.labels, phase=0 if self._split else -1,
                         lengths=self.lengths)

    def _seg1(self):     # backward(enc.0)
        if self._split:
            self.eng.fwd_bwd(self.X, self.labels, phase=1, lengths=self.lengths)

    def _seg2(self):     # Adam over the flat vector
        e = self.eng
        check(lib().pca_adam_step(e.flat.data_ptr(), e.grads.data_ptr(), self.m.data_ptr(),
                                  self.v.data_ptr(), e.flat.numel(), self.lr, self.betas[0],
                                  self.betas[1], self.eps, self.wd, 1.0 / self.world,
                                  self.step_count.data_ptr(), int(not self.keep_grads),
                                  e._stream()), "pca_adam_step")

    def _capture(self):
        """Warm up eagerly on a side stream, then capture the segments."""
        s = torch.cuda.Stream(self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        snap = (self.eng.flat.clone(), self.m.clone(), self.v.clone(),
                self.step_count.clone(), self.eng.stats.clone())
        with torch.cuda.stream(s):
            self._seg0(); self._seg1(); self._seg2()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        for dst, src in zip((self.eng.flat, self.m, self.v, self.step_count, self.eng.stats),
                            snap):
            dst.copy_(src)
        # thread-local capture mode: a HIP call from another thread (e.g. the RCCL watchdog of
        # torch.distributed) must not invalidate the capture
        mode = dict(capture_error_mode="thread_local")
        if self._captured:              # [pack .. backward | all-reduce | Adam] in ONE graph
            self.g0 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g0, **mode):
                self._captured_body()
        elif not self._split and not self._exchange:
            self.g0 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g0, **mode):
                self._seg0(); self._seg1(); self._seg2()
        elif not self._split:           # [pack, forward, backward] | all-reduce | Adam
            # (Adam is one kernel: a plain launch, not a one-node graph - every graph boundary
            #  costs tens of microseconds of GPU idle time)
            self.g0 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g0, **mode):
                self._seg0()
        else:
            self.g0, self.g1, self.g2 = (torch.cuda.CUDAGraph() for _ in range(3))
            with torch.cuda.graph(self.g0, **mode):
                self._seg0()
            with torch.cuda.graph(self.g1, pool=self.g0.pool(), **mode):
                self._seg1()
            with torch.cuda.graph(self.g2, pool=self.g0.pool(), **mode):
                self._seg2()
        for dst, src in zip((self.eng.flat, self.m, self.v, self.step_count, self.eng.stats),
                            snap):
            dst.copy_(src)

    def step(self) -> None:
        """One optimiser step; enqueues only (no host sync)."""
        if self._cursor_mode:
            spe = self.indices.steps_per_epoch()
            if self._k % spe == 0:        # once per epoch: stage its index batches
                ep = self.indices.next_epoch()
                if self.seq is None:
                    self.seq = torch.empty(spe * self.B, dtype=torch.int64, device=self.dev)
                self.seq.copy_(ep, non_blocking=True)
                self.epoch_base.fill_(self._k)
            self._k += 1
        else:
            self.idx.copy_(self._next_indices(), non_blocking=True)
        if self.use_graph and self.g0 is None:
            self._capture()
        main = torch.cuda.current_stream(self.dev)
        if self._captured:
            if self.use_graph:
                self.g0.replay()
            else:
                self._captured_body()
            return
        if not self._split and not self._exchange:
            if self.use_graph:
                self.g0.replay()
            else:
                self._seg0(); self._seg1(); self._seg2()
            return
        e = self.eng
        if not self._split:
            self.g0.replay() if self.use_graph else self._seg0()
            if self.world > 1:
                dist.all_reduce(e.grads, group=self.pg)
            self._seg2()
            return
        self.g0.replay() if self.use_graph else self._seg0()
        # bucket A (enc.1 + dec) is final: reduce it while enc.0's backward runs
        self.comm_stream.wait_stream(main)
        if self.world > 1:
            second = allreduce_buckets(e.grads, e.split, self.pg, self.comm_stream)
        self.g1.replay() if self.use_graph else self._seg1()
        if self.world > 1:
            second()
        main.wait_stream(self.comm_stream)
        self.g2.replay() if self.use_graph else self._seg2()

    # ---- epoch statistics ------------------------------------------------------------
    def read_stats(self, reset: bool = True) -> Tuple[float, float]:
        """(sum of per-sample losses, number of correct predictions) since the last reset,
        summed over ranks.  One host sync."""
        st = self.eng.stats.clone()
        if self.world > 1:
            dist.all_reduce(st, group=self.pg)
        out = st.cpu().tolist()
        self.eng.check_handoffs()
        if reset:
            self.eng.stats.zero_()
        return float(out[0]), float(out[1])


@torch.no_grad()
def evaluate(model, dataset, batch_size: int, mode: int = _lib.MODE_F32
             ) -> Tuple[float, int]:
    """Accuracy over ``dataset`` in order (full batches through the engine, the tail through
    a second engine sized for it).  Returns (accuracy, n)."""
    n = len(dataset)
    dev = next(model.parameters()).device
    correct = torch.zeros((), dtype=torch.int64, device=dev)
    done = 0
    while done < n:
        b = min(batch_size, n - done)
        eng = STEngine(model, b, dataset.num_points, mode, training=False)
        while done + b <= n:
            idx = torch.arange(done, done + b, device=dev)
            res = dataset.batch(idx)
            X, lab = res[0], res[1]
            logits = eng.forward(X, res[2] if len(res) > 2 else None)
            correct += (logits.argmax(1) == lab).sum()
            done += b
    return float(correct) / max(n, 1), n
