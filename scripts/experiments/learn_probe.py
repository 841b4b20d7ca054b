"""Scratch: how fast does the reference ST learn the synthetic corpus? (container only)"""
import os, sys, time
import numpy as np, torch
ROOT = "/root/repo"; REF = "/root/reference"
sys.path[:0] = [REF + "/Code", REF + "/set_transformer-master", ROOT + "/tests/golden", ROOT,
                ROOT + "/point-cloud-audio_amd"]
os.chdir(REF + "/Code")
import inputs as gi, models as ref_models
from oracle import st_oracle as so
C = int(os.environ.get("C", 10)); step_cls = int(os.environ.get("STEP", 5))
steps = int(os.environ.get("STEPS", 1500)); B = int(os.environ.get("B", 128))
d = int(os.environ.get("D", 128)); m = int(os.environ.get("M", 16)); h = int(os.environ.get("H", 4))
cpc = int(os.environ.get("CPC", 5))
xs, ys = [], []
for c in range(C):
    for j in range(cpc):
        w = so.synth_clip(cpc * c + j, c * step_cls, seconds=0.5)
        s = so.stft_logmag(w, 1024, drop_nyquist=True)
        xs.append(s); ys.append(np.full(s.shape[1], c))
x = np.concatenate(xs, 1); y = np.concatenate(ys)
F = 512; farr = (np.linspace(0, 22050, F + 1) / 44100)[:F].astype(np.float32)
torch.manual_seed(77); torch.set_num_threads(8)
net = ref_models.ST(dim_input=2, num_outputs=1, dim_output=C, num_inds=m, dim_hidden=d, num_heads=h)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-3)
crit = torch.nn.CrossEntropyLoss()
rng = np.random.default_rng(0); t0 = time.time()
for s in range(steps):
    idx = rng.integers(0, x.shape[1], B)
    X = np.empty((B, F, 2), np.float32); X[:, :, 0] = farr; X[:, :, 1] = x[:, idx].T
    preds = net(torch.from_numpy(X)); loss = crit(preds, torch.from_numpy(y[idx]))
    opt.zero_grad(); loss.backward(); opt.step()
    if s % 25 == 0:
        acc = (preds.argmax(1).numpy() == y[idx]).mean()
        print(s, f"{loss.item():.4f} acc {acc:.3f} {time.time()-t0:.0f}s", flush=True)
