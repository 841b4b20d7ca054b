import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-audio_amd")]
import numpy as np, torch
import bench, models
from pca_hip import trainer
from oracle import st_oracle as orc
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["cfg2"])
ds, _ = bench.build_dataset(cfg, 48, dev, seed=0)
print("frames", len(ds), "N", ds.num_points)
B = 128
torch.manual_seed(1)
net = models.ST(dim_input=2, dim_output=50, num_inds=16, dim_hidden=128, num_heads=4).to(dev)
p = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
opt = orc.AdamState(p)
torch.set_num_threads(16)
tr = trainer.Trainer(net, ds, B, use_graph=True, seed=1)
ls, lo = [], []
for s in range(40):
    tr.step()
    ls.append(round(float(tr.eng.loss), 3))
    X = tr.X.cpu(); y = tr.labels.cpu()
    l, _ = orc.train_step(X, y, p, opt, 4)
    lo.append(round(l, 3))
print("gpu   ", ls)
print("oracle", lo)
