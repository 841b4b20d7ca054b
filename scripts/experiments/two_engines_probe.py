import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "point-cloud-audio_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, inputs as gi, models
from pca_hip import _lib, trainer
B, N, din, d, h, m, C = 128, 512, 2, 128, 4, 16, 50
dev = torch.device("cuda", 0)
torch.manual_seed(1)
net = models.ST(dim_input=din, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d, num_heads=h).to(dev)
X = torch.from_numpy(gi.pc_input(2, B, N, din)).to(dev); y = torch.from_numpy(gi.labels(3, B, C)).to(dev)
outs = []
keep = []
for e in range(3):
    eng = trainer.STEngine(net, B, N, _lib.MODE_BF16, training=True)
    keep.append(eng)
    junk = torch.randn(1 << 22, device=dev)   # perturb allocator state between engines
    keep.append(junk)
    eng.grads.zero_(); eng.fwd_bwd(X, y, phase=-1); torch.cuda.synchronize()
    outs.append(eng.grads.clone())
off = 0; bad = 0
for k, prm in net.named_parameters():
    n = prm.numel()
    dd = max(float((outs[0][off:off+n] - outs[i][off:off+n]).abs().max()) for i in (1, 2))
    if dd > 0: bad += 1; print(k, dd)
    off += n
print("tensors that differ across engines:", bad)
