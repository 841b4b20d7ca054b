"""Can the gradient all-reduce be captured INSIDE the step's hipGraph?  One rank, backend nccl (= RCCL),
on the one GPU of the test box (a world of one is enough to exercise RCCL's stream capture; the
all-reduce over one rank is the identity, so the parameters must equal the plain one-graph step's bit
for bit).  Prints CAPTURE_OK / CAPTURE_REFUSED <error text>.  Launched by tests/test_gpu_ddp.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-audio_amd")]
import numpy as np, torch, torch.distributed as dist
import dataset, models
from pca_hip import _lib, trainer

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ["PCA_EXCHANGE_WORLD1"] = "1"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
    probe = torch.ones(1, device=dev)
    dist.all_reduce(probe)
    torch.cuda.synchronize()
except Exception as e:
    print("BACKEND_REFUSED", repr(e)[:600], flush=True)
    sys.exit(0)
rng = np.random.Generator(np.random.PCG64(5))
F, T, C, B = 256, 640, 10, 32
x = rng.normal(-9, 3, size=(F, T)).astype(np.float32)
y = rng.integers(0, C, size=(T,))
farr = np.linspace(0, 0.5, F)
mode = _lib.MODE_BF16 if os.environ.get("PCA_MODE", "bf16") == "bf16" else _lib.MODE_F32


def run(form):
    torch.manual_seed(3)
    net = models.ST(dim_input=2, dim_output=C, num_inds=16, dim_hidden=128, num_heads=4).to(dev)
    ds = dataset.ESC_pc(x, y, farr, device=dev)
    tr = trainer.Trainer(net, ds, B, mode=mode, use_graph=True, seed=11, shuffle=True,
                         process_group=dist.group.WORLD)
    if form is not None:
        tr.set_exchange(form)
    for _ in range(6):
        tr.step()
    torch.cuda.synchronize()
    return tr.eng.flat.detach().clone(), tr


plain, _ = run(None)
try:
    cap, tr = run("captured")
    assert tr._captured and tr.g0 is not None and tr.g1 is None
    print("CAPTURE_OK", bool(torch.equal(plain, cap)), float((plain - cap).abs().max()), flush=True)
except Exception as e:
    print("CAPTURE_REFUSED", repr(e)[:1500].replace("\n", " | "), flush=True)
# the split form of the same graph: bucket A's all-reduce forked onto a side stream under enc.0's backward.
# The phase-split backward sums some gradients in another order than the one-call backward: compare with
# the un-captured split step ("overlap"), whose arithmetic it shares, bit for bit.
try:
    ovl, _ = run("overlap")
    cov, tr = run("captured_overlap")
    assert tr._captured and tr._split and tr.g0 is not None and tr.g1 is None
    print("CAPTURE_OVERLAP_OK", bool(torch.equal(ovl, cov)), float((ovl - cov).abs().max()),
          float((plain - cov).abs().max()), flush=True)
except Exception as e:
    print("CAPTURE_OVERLAP_REFUSED", repr(e)[:1500].replace("\n", " | "), flush=True)
dist.destroy_process_group()
