"""Per-region table of one leg's train step (sar_amd.profiler regions: calls and ms per step, in-step timings with the weight-gradient
stream on): which arithmetic each GEMM family of the step actually took.   python tools/step_regions.py [fp32|bf16|f32_split]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "skeleton-action-recognition_amd"))
sys.path.insert(0, ROOT)
import torch
from sar_amd import profiler
from sar_amd.stgcn import STGCN
from sar_amd.train import synthetic_clips
dev = torch.device("cuda:0")
eng = STGCN(num_classes=60, device=dev, mfma=sys.argv[1] if len(sys.argv) > 1 else "f32_split")
x, y = synthetic_clips(64, dev, seed=0, num_classes=60)
for _ in range(4):
    eng.loss_and_grad(x, y); eng.sgd_step(0.1)
t = profiler.KernelTimer(); profiler.install(t)
for _ in range(5):
    eng.loss_and_grad(x, y); eng.sgd_step(0.1)
torch.cuda.synchronize(); profiler.install(None)
for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1]["ms"]):
    print("%-40s calls %3d  %.3f ms/step" % (k, v["calls"] // 5, v["ms"] / 5))
