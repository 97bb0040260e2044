"""Does one hipGraph launch per train step pay for the ST-GCN engines?  (experiment; SpectrogramTrainer(graph=True) is the product form
for Path B)  Usage: python tools/graph_probe.py [fp32|f32_split|bf16] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from sar_amd.stgcn import STGCN  # noqa: E402
from sar_amd.train import synthetic_clips  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
eng = STGCN(num_classes=60, device=dev, mfma=mode)
x, y = synthetic_clips(bs, dev, seed=0)


def step():
    out = eng.loss_and_grad(x, y, bs)
    eng.sgd_step(0.1, 0.9)
    return out


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3


for _ in range(5):
    step()
torch.cuda.synchronize()
n = 60 if mode == "bf16" else 20
e = timed(step, n)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
r = timed(g.replay, n)
print("%s bs %d: eager %.3f ms/step (%.0f clips/s), graph replay %.3f ms/step (%.0f clips/s)" % (mode, bs, e, bs / e * 1e3, r, bs / r * 1e3))
