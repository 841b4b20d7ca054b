"""Launched by tests/test_gpu_ddp.py under torch.distributed.run with 2 ranks: on ONE GPU with
backend gloo (the rehearsal a one-GPU box allows), or with PCA_DIST_BACKEND=nccl (RCCL) - one
rank per GPU when the box has two, both on GPU 0 otherwise (RCCL refuses that: the test records
the refusal and skips).  The multi-rank step (phase-0 graph, bucket-A all-reduce on the side stream,
phase-1 graph, bucket-B all-reduce, Adam) must reproduce a single-rank run on the union batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-audio_amd")]
import numpy as np, torch, torch.distributed as dist
import dataset, models
from pca_hip import _lib, trainer

mode = _lib.MODE_BF16 if os.environ.get("PCA_MODE", "f32") == "bf16" else _lib.MODE_F32
use_graph = os.environ.get("PCA_GRAPH", "1") == "1"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("PCA_DIST_BACKEND", "gloo")
local = int(os.environ.get("LOCAL_RANK", "0")) if (backend == "nccl" and torch.cuda.device_count() >= 2) else 0
dev = torch.device("cuda", local)
torch.cuda.set_device(local)
try:
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)              # the first collective creates the communicator
        torch.cuda.synchronize()
        print(f"NCCL_RANKS {world} probe {float(probe)}", flush=True)
    else:
        dist.init_process_group("gloo")
except Exception as e:                      # e.g. two ranks on one device
    print("BACKEND_REFUSED", backend, repr(e)[:600], flush=True)
    sys.exit(0)
rng = np.random.Generator(np.random.PCG64(5))
F, T, C, B = 256, 640, 10, 32
x = rng.normal(-9, 3, size=(F, T)).astype(np.float32)
y = rng.integers(0, C, size=(T,))
farr = np.linspace(0, 0.5, F)

def run(world_sim, rank_sim, batch, pg):
    torch.manual_seed(3)
    net = models.ST(dim_input=2, dim_output=C, num_inds=16, dim_hidden=128, num_heads=4).to(dev)
    ds = dataset.ESC_pc(x, y, farr, device=dev)
    tr = trainer.Trainer(net, ds, batch, mode=mode, use_graph=use_graph, seed=11, shuffle=True,
                         process_group=pg, overlap=os.environ.get("PCA_OVERLAP", "1") == "1")
    if pg is None:                      # force a single-rank trainer inside the 2-rank job
        tr.world, tr.rank = 1, 0
        tr.indices = trainer.ShardedIndexStream(len(ds), batch, 0, 1, 11, True, dev)
        tr.comm_stream = None
        tr._split = tr._exchange = False
    for _ in range(2 if (pg is not None and os.environ.get("PCA_SWITCH_TO")) else 4):
        tr.step()
    if pg is not None and os.environ.get("PCA_SWITCH_TO"):
        # bench.py's dual measurement: the other step shape on the same Trainer, mid-run
        tr.set_exchange(os.environ["PCA_SWITCH_TO"])
        for _ in range(2):
            tr.step()
        if rank == 0:
            print("SWITCHED_TO", tr.exchange_form, flush=True)
    torch.cuda.synchronize()
    return tr.eng.flat.detach().cpu().clone(), tr.read_stats() if pg is not None else None

flat2, stats2 = run(world, rank, B, dist.group.WORLD)
dist.barrier()
# every rank holds identical parameters after the steps
mine = flat2.to(dev) if backend == "nccl" else flat2
gathered = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(gathered, mine)
same = all(torch.equal(gathered[0], g) for g in gathered)
if rank == 0:
    print("RANKS_IDENTICAL", same)
    print("FLAT_NORM", float(flat2.norm()))
    torch.save(flat2, os.environ["PCA_OUT"])
dist.destroy_process_group()
