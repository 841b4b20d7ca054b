import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-audio_amd")]
import numpy as np, torch
import bench, models
from pca_hip import trainer
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["cfg2"])
ds, _ = bench.build_dataset(cfg, 12, dev, seed=0)
torch.manual_seed(1)
net = models.ST(dim_input=2, dim_output=50, num_inds=16, dim_hidden=128, num_heads=4).to(dev)
tr = trainer.Trainer(net, ds, 128, use_graph=True, seed=1)
sync = len(sys.argv) > 1
for s in range(12):
    tr.step()
    if sync:
        torch.cuda.synchronize()
        print(s, float(tr.eng.loss), tr.eng.stats.tolist(), int(tr.step_count[0]), float(tr.eng.flat.abs().max()), float(tr.eng.grads.abs().max()))
torch.cuda.synchronize()
print("final", float(tr.eng.loss), tr.eng.stats.tolist(), int(tr.step_count[0]))
