"""How much of a train step is host time?  Enqueue time per step (Python + ctypes + allocator, no synchronisation) against the
GPU time per step, for both ST-GCN configurations.  python tools/host_overhead.py"""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/skeleton-action-recognition_amd")
from sar_amd.stgcn import STGCN
from sar_amd.train import Trainer, synthetic_clips
dev = torch.device("cuda:0")
for mode in ("bf16", "fp32"):
    eng = STGCN(num_classes=60, device=dev, seed=0, mfma=mode)
    tr = Trainer(eng, batch_size=64)
    x, y = synthetic_clips(64, dev, seed=1)
    for _ in range(3):
        tr.step(x, y)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(x, y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: enqueue %.2f ms/step, wall %.2f ms/step (GPU-bound if enqueue << wall)" % (mode, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))

# Path B (VirtualRadar -> resnet18, bs = 32)
sys.path.insert(0, ROOT + "/skeleton-action-recognition_amd")
from models.resnet import Model  # noqa: E402
from sar_amd.train import SpectrogramTrainer  # noqa: E402
model = Model(num_classes=60, device=dev)
trainer = SpectrogramTrainer(model, 1e-3, 1)
x, y = synthetic_clips(32, dev, seed=1)
for _ in range(5):
    trainer.step(x, y, 1e-3)
torch.cuda.synchronize()
for rep in range(3):
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        trainer.step(x, y, 1e-3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("spectrogram: enqueue %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
