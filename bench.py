#!/usr/bin/env python3
"""Headline benchmark: train clips/sec of the ESC-50-shaped Set Transformer on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input per GPU:
  gather B point sets from the HBM-resident log-magnitude spectrogram (pack kernel) ->
  ST forward -> mean cross-entropy -> backward -> [all-reduce of the flat gradient vector
  over RCCL] -> fused Adam (coupled weight decay).
The spectrogram itself is produced once, before the timed region, by the STFT kernel from
synthetic class-conditional clips (5 s @ 44.1 kHz) -- the reference also runs its STFT as
a one-off pre-pass (Code/settransformer.py:43-53).  value = sets/s of all ranks divided by
the sets one clip yields at this framing (cfg2: 431).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "point-cloud-audio_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# name: (din, n_fft, ntemp, d, h, m, C, B per GPU, sets per 5-s clip, description)
CONFIGS = {
    "cfg2": dict(din=2, n_fft=1024, ntemp=1, d=128, h=4, m=16, C=50, B=128, sets_per_clip=431,
                 desc="BASELINE configs[1]: ESC-50-shaped 2-D point sets N=512, ST d=128 h=4 "
                      "m=16 C=50"),
    "cfg3": dict(din=3, n_fft=1024, ntemp=4, d=128, h=4, m=16, C=50, B=32, sets_per_clip=107,
                 desc="BASELINE configs[2]: 3-D temporal point sets N=2048, ST d=128 h=4 m=16"),
    "cfg4": dict(din=3, n_fft=1024, ntemp=8, d=256, h=8, m=32, C=50, B=128, sets_per_clip=53,
                 desc="BASELINE configs[3] shape: N=4096, ST d=256 h=8 m=32"),
    "cfg5": dict(din=3, n_fft=1024, ntemp=8, d=256, h=8, m=32, C=10, B=128, sets_per_clip=26,
                 varlen=True,
                 desc="BASELINE configs[4] shape: UrbanSound8K-shaped clips of 1-4 s (padded "
                      "variable-size sets, N <= 4096), ST d=256 h=8 m=32, 10 classes"),
    "fst": dict(din=2, n_fft=2048, ntemp=1, d=64, h=8, m=64, C=10, B=128, sets_per_clip=216,
                desc="shipped FST shape: N=1025, ST d=64 h=8 m=64"),
    "3st": dict(din=3, n_fft=1024, ntemp=10, d=64, h=8, m=64, C=10, B=16, sets_per_clip=43,
                desc="shipped 3ST shape: N=5120, ST d=64 h=8 m=64"),
}
FS = 44100
MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (non-scaled fp8 MFMA
                                   # issues at the same rate: same peak for the fp8 mode here)
FP32_VALU_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0


def st_fwd_macs(N, din, d, m, k, C):
    """SURVEY.md 8d closed form, reference formulation (MACs per set, forward)."""
    return N * (3 * din * d + 7 * d * d + 8 * m * d + 2 * k * d) + 6 * m * d * d + k * d * d + k * d * C


def synth_clip(clip_id: int, cls: int, seconds: float = 5.0, fs: int = FS) -> np.ndarray:
    """Synthetic class-conditional clip of SURVEY.md 8d: 3 harmonics of f0(c) = 110*2^(c/12) Hz
    with seeded phases + low-passed noise, PCG64(seed = 1000 + clip_id); float32 in [-1, 1].
    (The bench's own generator: nothing under oracle/ is touched outside cpu_baseline().)"""
    rng = np.random.Generator(np.random.PCG64(1000 + clip_id))
    L = int(round(seconds * fs))
    t = np.arange(L) / fs
    f0 = 110.0 * 2.0 ** (cls / 12.0)
    x = np.zeros(L)
    for k in range(1, 4):
        x += (0.5 / k) * np.sin(2 * np.pi * f0 * k * t + rng.uniform(0, 2 * np.pi))
    noise = np.convolve(rng.standard_normal(L), np.ones(8) / 8.0, mode="same")
    x = x + 0.1 * noise
    x = x / (np.max(np.abs(x)) + 1e-9) * 0.9
    return x.astype(np.float32)


def build_dataset(cfg, n_clips, dev, seed):
    """Synthetic class-conditional clips -> STFT kernel -> device-resident dataset."""
    import dataset
    import pca_hip
    n_fft, ntemp, C_ = cfg["n_fft"], cfg["ntemp"], cfg["C"]
    hop = n_fft // 2
    drop = not (cfg["din"] == 2 and n_fft == 2048)      # FST keeps the Nyquist bin (N=1025)
    F = n_fft // 2 if drop else n_fft // 2 + 1
    varlen = bool(cfg.get("varlen"))
    nt_valid = []
    waves, classes = [], []
    for i in range(n_clips):
        cls = (seed * 7919 + i) % C_
        # UrbanSound8K-shaped: durations uniform in [1, 4] s
        secs = 5.0 if not varlen else float(
            np.random.Generator(np.random.PCG64(77 + i)).uniform(1.0, 4.0))
        waves.append(torch.from_numpy(synth_clip(seed * 100000 + i, cls, seconds=secs)).to(dev))
        classes.append(cls)
    # the whole corpus through ONE launch of the STFT kernel (round 2 launched it per clip); timed
    # on the second call (the first pays the one-time costs of the kernel's first launch)
    for _ in range(2):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        spec_all, foff = pca_hip.stft_logmag_batch(waves, n_fft, n_fft, hop, drop_nyquist=drop,
                                                   frame_major=True)
        torch.cuda.synchronize(dev)
        t_stft = time.perf_counter() - t0
    specs, labels = [], []
    for i, cls in enumerate(classes):
        s = spec_all[foff[i]:foff[i + 1]]
        if ntemp > 1 and varlen:
            # keep the short last chunk as a padded set with fewer valid frames
            T_ = s.shape[0]
            S = -(-T_ // ntemp)
            pad = torch.zeros((S * ntemp - T_, F), dtype=s.dtype, device=dev)
            s = torch.cat([s, pad]).reshape(S, ntemp, F)
            nt_valid += [ntemp] * (S - 1) + [T_ - (S - 1) * ntemp]
        elif ntemp > 1:
            S = s.shape[0] // ntemp
            s = s[:S * ntemp].reshape(S, ntemp, F)       # chunks of ntemp frames, tail dropped
        specs.append(s)
        labels.append(torch.full((s.shape[0],), cls, dtype=torch.int64, device=dev))
    spec = torch.cat(specs).contiguous()
    lab = torch.cat(labels)
    farr = np.linspace(0, FS / 2, F) / FS if drop else np.linspace(0, FS / 2, n_fft // 2 + 1) / FS
    if ntemp == 1:
        ds = dataset.ESC_pc.from_device(spec, lab, farr)
    else:
        tarr = np.linspace(0, (hop / FS) * ntemp, ntemp)
        ds = dataset.ESC_pc_temp.from_device(spec, lab, farr, tarr)
        if varlen:
            ds.nt_valid = np.asarray(nt_valid, dtype=np.int32)
    return ds, t_stft / max(n_clips, 1)


def cpu_baseline(cfg, N, budget_s=20.0):
    """Oracle ('port') train step on the host cores: same B, N, architecture, fp32, Adam."""
    from oracle import st_oracle as orc
    ncore = min(16, os.cpu_count() or 1)
    torch.set_num_threads(ncore)
    B = cfg["B"]
    p = orc.st_init_params(cfg["din"], 1, cfg["C"], cfg["m"], cfg["d"], seed=0)
    opt = orc.AdamState(p)
    g = torch.Generator().manual_seed(0)
    X = torch.randn(B, N, cfg["din"], generator=g)
    y = torch.randint(0, cfg["C"], (B,), generator=g)
    orc.train_step(X, y, p, opt, cfg["h"])           # warm-up
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 50):
        t0 = time.perf_counter()
        orc.train_step(X, y, p, opt, cfg["h"])
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    # the reference's per-item host packing (Code/dataset.py:50-54 / 160-166), one core: it caps
    # the reference's loader independently of the model (SURVEY.md 8d asks for it as its own line)
    rng = np.random.Generator(np.random.PCG64(0))
    if cfg["din"] == 2:
        xs = rng.standard_normal((N, 512)).astype(np.float32)
        farr = np.linspace(0, 0.5, N)
        pack = lambda i: orc.pack_points_2d(xs, farr, i % 512)                  # noqa: E731
    else:
        F = N // cfg["ntemp"]
        xs = rng.standard_normal((F, cfg["ntemp"], 64)).astype(np.float32)
        farr, tarr = np.linspace(0, 0.5, F), np.linspace(0, 0.1, cfg["ntemp"])
        pack = lambda i: orc.pack_points_3d(xs, farr, tarr, i % 64)              # noqa: E731
    t0 = time.perf_counter()
    n_items = 0
    while time.perf_counter() - t0 < 1.0:
        pack(n_items)
        n_items += 1
    pack_rate = n_items / (time.perf_counter() - t0)
    return dict(value=round(B / med / cfg["sets_per_clip"], 4), unit="clips/s", cores=ncore,
                kind="port",
                sample=f"{len(times)} train steps of B={B} sets (median {med * 1e3:.1f} ms/step, "
                       f"{B / med:.1f} sets/s), oracle/st_oracle.py on torch CPU fp32",
                pack_sets_per_s_1core=round(pack_rate, 1))


def event_pair_overhead_us(dev, n=200):
    """What two back-to-back event records on an idle stream measure (their own cost): the HIP
    event clock of the roofline block includes it once per launch."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(n)]
    torch.cuda.synchronize(dev)
    for a, b in evs:
        a.record()
        b.record()
    torch.cuda.synchronize(dev)
    return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3


def committed_traffic(key):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary
    (profiles/r*_hbm_traffic.json) - or None when that file predates the library it would
    describe (a stale constant is worse than none)."""
    import glob
    if key is None:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    so = os.path.join(ROOT, "point-cloud-audio_amd", "pca_hip", "libpca_hip.so")
    for path in reversed(files):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if key not in d:
            continue
        import hashlib
        cur = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]
        return d[key] if d.get("library_sha16") == cur else None
    return None


def measure_kernel(L, _lib, tr, dev, kid, ksteps, peak):
    """HIP-event time, algorithmic FLOPs and bytes of every launch of kernel family ``kid``
    over ``ksteps`` eager steps -> roofline dictionary (None when it never ran)."""
    _lib.check(L.pca_prof_start(kid, 4096 * ksteps), "prof_start")
    for _ in range(ksteps):
        tr.step()
    torch.cuda.synchronize(dev)
    ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    _lib.check(L.pca_prof_stop(C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "prof_stop")
    if n.value <= 0 or ms.value <= 0:
        return None
    tflops = fl.value / (ms.value * 1e-3) / 1e12
    gbs = by.value / (ms.value * 1e-3) / 1e9
    # roofline model: HBM-bound when the algorithmic intensity lies below the ridge
    intensity = fl.value / max(by.value, 1.0)
    hbm_bound = intensity < peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    return {
        "kernel": "", "bound": "hbm" if hbm_bound else "mfma",
        "achieved": round(gbs, 1) if hbm_bound else round(tflops, 3),
        "peak": HBM_PEAK_GBS if hbm_bound else peak,
        "unit": "GB/s" if hbm_bound else "TFLOP/s",
        "frac": round(gbs / HBM_PEAK_GBS, 5) if hbm_bound else round(tflops / peak, 5),
        "traffic": None,
        "alg_intensity_flop_per_byte": round(intensity, 1),
        "ridge_flop_per_byte": round(peak * 1e12 / (HBM_PEAK_GBS * 1e9), 1),
        "achieved_tflops": round(tflops, 3), "frac_of_mfma_peak": round(tflops / peak, 5),
        "achieved_gbs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 5),
        "launches": int(n.value), "avg_us": round(ms.value * 1e3 / n.value, 3),
        "alg_flops_per_launch": round(fl.value / n.value),
        "alg_bytes_per_launch": round(by.value / n.value),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="sets per GPU per step (0 = config)")
    ap.add_argument("--mode", default="bf16", choices=["f32", "bf16", "fp8"],
                    help="bf16 = fused MFMA kernels (bf16 operands, fp32 accumulate); fp8 = the "
                         "same with fp8 e4m3 operands in the d x d projections of the forward; "
                         "f32 = exact parity path")
    ap.add_argument("--clips", type=int, default=48, help="synthetic clips in the corpus")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--windows", type=int, default=50,
                    help="extra timed windows of --steps steps each after the headline window "
                         "(spread estimate: median / p10 / p90 in 'windows'; 0 = off)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    # PCA_FORCE_DEVICE: rehearse the multi-rank path on a one-GPU box (all ranks on one card,
    # PCA_DIST_BACKEND=gloo); never set in a real run
    local_rank = int(os.environ.get("PCA_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend, rccl_ranks = None, None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PCA_DIST_BACKEND", "nccl")
        # a multi-GPU line is an RCCL line: another backend is accepted only for the one-GPU rehearsal
        # (PCA_FORCE_DEVICE), so that a gloo fall-back can never produce a SCALE record
        if backend != "nccl" and "PCA_FORCE_DEVICE" not in os.environ:
            raise SystemExit(f"bench: --gpus {world} needs backend nccl (RCCL), got PCA_DIST_BACKEND="
                             f"{backend}; set PCA_FORCE_DEVICE to rehearse on one GPU")
        dist.init_process_group(backend, device_id=dev if backend == "nccl" else None)
        # self-check of the exchange the step relies on: a sum of ones over the ranks, on the
        # device, through the same backend (nccl = RCCL over xGMI) - reported in the JSON line
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(round(float(ones)))
        assert rccl_ranks == world, f"all-reduce of ones over {world} ranks gave {rccl_ranks}"

    import models
    import pca_hip
    from pca_hip import _lib, trainer
    L = pca_hip.lib()

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    mode = {"f32": _lib.MODE_F32, "bf16": _lib.MODE_BF16, "fp8": _lib.MODE_FP8}[args.mode]
    # the same corpus on every rank: the index sharding of the Trainer (rank r takes elements
    # r::world of ONE permutation) assumes identical datasets, as DistributedSampler does
    ds, stft_s_per_clip = build_dataset(cfg, args.clips, dev, seed=0)
    N = ds.num_points
    torch.manual_seed(1)
    net = models.ST(dim_input=cfg["din"], num_outputs=1, dim_output=cfg["C"],
                    num_inds=cfg["m"], dim_hidden=cfg["d"], num_heads=cfg["h"]).to(dev)
    tr = trainer.Trainer(net, ds, cfg["B"], lr=1e-3, weight_decay=1e-3, mode=mode,
                         use_graph=not args.no_graph, seed=1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- several GPUs: which step shape exchanges the gradients fastest is measured, not assumed ---
    # (round-3 verdict: the overlap / serial choice had never met a real all-reduce.)  A short window
    # of each form after a warm-up of its own; every rank takes the same decision (max over ranks);
    # the timed region below runs the winner.
    exchange = None
    if world > 1:
        forms = ["serial", "overlap"] + (["captured", "captured_overlap"] if backend == "nccl" else [])
        k_probe = max(10, min(50, args.steps))
        exchange = {}
        for form in forms:
            # a collective that never completes inside a replayed graph is invisible to RCCL's watchdog:
            # give every probe a deadline of its own, so that a hung form ends the run in minutes with a
            # message instead of holding eight GPUs until the caller's limit
            dog = threading.Timer(300.0, lambda f=form: (sys.stderr.write(
                f"bench: exchange form '{f}' made no progress for 300 s on rank {rank}, aborting\n"),
                sys.stderr.flush(), os._exit(3)))
            dog.daemon = True
            dog.start()
            try:
                tr.set_exchange(form)
                for _ in range(max(5, args.warmup // 2)):
                    tr.step()
                barrier()
                t0 = time.perf_counter()
                for _ in range(k_probe):
                    tr.step()
                barrier()
                t = torch.tensor([(time.perf_counter() - t0) / k_probe * 1e3], dtype=torch.float64,
                                 device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                exchange[form + "_ms"] = round(float(t), 4)
            except Exception as e:          # e.g. a backend that refuses stream capture
                exchange[form + "_error"] = repr(e)[:300]
            finally:
                dog.cancel()
        timed = {f: exchange[f + "_ms"] for f in forms if f + "_ms" in exchange}
        # (a form that failed on ANY rank is out: agree on it)
        for f in list(timed):
            bad = torch.tensor([0.0 if f + "_ms" in exchange else 1.0], device=dev)
            dist.all_reduce(bad)
            if float(bad) > 0:
                timed.pop(f)
        chosen = min(timed, key=timed.get)
        exchange["chosen"] = chosen
        tr.set_exchange(chosen)
        # the all-reduce by itself (HIP events on the stream it is issued from), eager
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        g = tr.eng.grads
        for _ in range(3):
            dist.all_reduce(g)
        torch.cuda.synchronize(dev)
        ev[0].record()
        for _ in range(20):
            dist.all_reduce(g)
        ev[1].record()
        torch.cuda.synchronize(dev)
        exchange["allreduce_alone_ms"] = round(ev[0].elapsed_time(ev[1]) / 20, 4)
        g.zero_()

    for _ in range(args.warmup):
        tr.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    # spread: more windows of the same K steps, each bracketed like the headline window (bounded
    # to ~10 s so that slow configurations still finish in minutes)
    win_ms = []
    n_win = args.windows if elapsed * args.windows <= 10.0 else max(0, int(10.0 / max(elapsed, 1e-9)))
    for _ in range(n_win):
        barrier()
        w0 = time.perf_counter()
        for _ in range(args.steps):
            tr.step()
        barrier()
        win_ms.append((time.perf_counter() - w0) / args.steps * 1e3)
    if world > 1 and win_ms:
        t = torch.tensor(win_ms, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        win_ms = t.cpu().tolist()
    loss_sum, correct = tr.read_stats()
    n_seen = (args.warmup + args.steps * (1 + len(win_ms))) * cfg["B"] * world
    if not np.isfinite(loss_sum) or not bool(torch.isfinite(tr.eng.flat).all()):
        raise SystemExit(f"bench: non-finite training state (loss sum {loss_sum}): the timed "
                         "steps are invalid")

    sets_per_s = args.steps * cfg["B"] * world / elapsed
    out = {
        "metric": "train clips/sec, ESC-50 SetTransformer",
        "value": round(sets_per_s / cfg["sets_per_clip"], 3),
        "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.mode, "data": "synthetic",
        "config": {"workload": f"{args.config}: {cfg['desc']}; N={N} points/set, "
                               f"B={cfg['B']} sets/GPU/step, CE + Adam(lr 1e-3, wd 1e-3)",
                   "global_batch": cfg["B"] * world, "points_per_set": N,
                   "sets_per_clip": cfg["sets_per_clip"], "parallelism": f"dp{world}",
                   "hipgraph": not args.no_graph},
        "sets_per_s": round(sets_per_s, 1),
        "train_loss_mean": round(loss_sum / max(n_seen, 1), 4),
        "stft_ms_per_clip": round(stft_s_per_clip * 1e3, 3),
        "dist_backend": backend, "rccl_ranks": rccl_ranks,
        "grad_allreduce_bytes": int(tr.eng.grads.numel()) * 4 if world > 1 else 0,
    }
    if exchange is not None:
        out["exchange"] = exchange
    if win_ms:
        q = np.percentile(np.asarray(win_ms), [10, 50, 90])
        out["windows"] = {"n": len(win_ms), "steps_each": args.steps,
                          "ms_per_step_p10": round(float(q[0]), 4),
                          "ms_per_step_median": round(float(q[1]), 4),
                          "ms_per_step_p90": round(float(q[2]), 4),
                          "value_median": round(cfg["B"] * world / (float(q[1]) * 1e-3)
                                                / cfg["sets_per_clip"], 3)}

    # ---- roofline of the dominant kernel(s): HIP events on their launch stream, live ------
    if not args.no_roofline:
        # un-captured steps of the same workload so that the events bracket real launches
        tr_use_graph = tr.use_graph
        tr.use_graph = False
        for _ in range(3):
            tr.step()
        torch.cuda.synchronize(dev)
        peak = FP32_VALU_PEAK_TFLOPS if args.mode == "f32" else MFMA_BF16_PEAK_TFLOPS
        overhead_us = event_pair_overhead_us(dev)
        if args.mode == "f32":
            kernels = [(_lib.K_GEMM_F32, "k_gemm_f32", None,
                        "exact-fp32 parity path: strided VALU GEMMs, priced against the fp32 "
                        "vector/matrix peak (157.3 TFLOP/s)")]
        elif cfg["d"] == 256:
            # configs[3] / [4]: the dominant kernel is the weight-gradient reduction k_wgrad256
            # (17 % of the step, profiles/r0*_bf16_cfg4_kernel_stats.csv), then k_attn1_bwd3
            kernels = [
                (_lib.K_WGRAD, "k_wgrad256 (dW = G^T A over the B*N rows, all jobs of the step; "
                               "the [B*m]-row fp32 jobs of the deferred launch included)",
                 None,
                 "2*rows*256*256 FLOPs per job; algorithmic bytes = both operands of every job "
                 "once (an operand shared by the jobs of one launch once)"),
                (_lib.K_MAB1_BWD, "k_attn1_bwd3 (+ k_sum_parts256): fc_o adjoint + attention "
                                  "adjoint of the many-queries block, both layers",
                 None,
                 "reference-formulation FLOPs of the launches inside the scope only: "
                 "M*(2*d^2 + 8*m*d) (dX runs in k_rowstream, the weight gradients in k_wgrad256); "
                 "algorithmic bytes = dY + Qp in (layer 1: the points), dZ + dQp out"),
            ]
        else:
            kernels = [
                # round 4: the set-resident forward (one launch: both ISABs + the PMA attention of a
                # set, a pair of workgroups per set) is the longest launch of the cfg2 step; on shapes
                # it does not take (cfg3, B > #CUs / 2) it never runs and the next entries report
                (_lib.K_SET_FWD, "k_set128_fwd (set-resident forward: ISAB, ISAB, PMA attention of a set "
                                 "in ONE launch, a pair of workgroups per set)",
                 "k_set128_fwd_bytes_per_launch",
                 "reference-formulation FLOPs of the three blocks (SURVEY.md 8d: 2 * B * [N (3 din d + "
                 "7 d^2 + 8 m d + 2 d) + 6 m d^2]); algorithmic bytes = 4 N din + 4 (2 N d) per set "
                 "(SURVEY.md 8d 'algorithmic bytes per set'): the saved tensors the training forward "
                 "also writes (O, Qp, masks) are traffic, not algorithmic bytes"),
                (_lib.K_MAB1_BWD, "k_mab1_bwd (fused ISAB mab1 backward chain, both layers)",
                 "k_mab1_bwd_bytes_per_launch",
                 "reference-formulation FLOPs of what the launch computes: M*(2*d^2 + 8*m*d "
                 "+ 2*dq*d [dX, or layer 1's in-kernel dWq]) (dWo / dWq of the d -> d layer run in "
                 "k_wgrad128 and are not charged here); algorithmic bytes = dY in + X in + dX out, "
                 "bf16 activations"),
                (_lib.K_MAB0_BWD, "k_mab0_bwd (fused ISAB mab0 / PMA backward, few queries)",
                 "k_mab0_bwd_bytes_per_launch",
                 "reference-formulation FLOPs 4*M*(2*dk*d + 2*m*d) per launch; algorithmic bytes "
                 "= X in + dX read-modify-write"),
            ]
        names = ["roofline", "roofline2", "roofline3"]
        ni = 0
        for (kid, label, tkey, note) in kernels:
            if ni >= 2 and len(kernels) <= 2:
                break
            r = measure_kernel(L, _lib, tr, dev, kid, min(args.steps, 20), peak)
            if r is None:
                continue
            key = names[ni]
            ni += 1
            r["kernel"] = label
            r["traffic"] = committed_traffic(tkey) if args.config == "cfg2" else None
            r["clock"] = ("HIP events recorded by the library around every launch of this kernel on "
                          "its launch stream (pca_prof_start/stop), eager steps of the same "
                          "workload; an event pair around nothing measures "
                          f"{overhead_us:.2f} us here, which is included in avg_us (the rocprofv3 "
                          "kernel-trace average under profiles/ is the kernel alone)")
            r["event_pair_overhead_us"] = round(overhead_us, 3)
            r["note"] = note + ("; traffic = HBM bytes per launch from rocprofv3 PMC "
                                "(profiles/, FETCH_SIZE doubled per MI355X_MICROARCH.md), null when "
                                "no committed measurement matches the current library"
                                if tkey else "")
            out[key] = r
        tr.use_graph = tr_use_graph
        fwd = st_fwd_macs(N, cfg["din"], cfg["d"], cfg["m"], 1, cfg["C"]) * 2
        out["model_tflops_ref_formulation"] = round(
            3 * fwd * cfg["B"] * world * args.steps / elapsed / 1e12, 3)

    if world > 1:
        dist.barrier()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, N)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
