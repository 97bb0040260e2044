"""Repeatability soak of the VirtualRadar -> spectrogram stage (with and without the fused x250 up-sampling): same clips, many runs, bitwise."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/skeleton-action-recognition_amd")
from utils import import_class
from sar_amd.train import synthetic_clips
dev = torch.device("cuda:0")
Model = import_class("models.resnet.Model")
for pad in (0, 250):
    model = Model(num_classes=60, num_filters=64, device=dev, num_pad_frames=pad, sigma=3)
    x, y = synthetic_clips(8 if pad else 32, dev, seed=3, num_classes=60)
    ref, bad = None, 0
    for r in range(60 if pad else 150):
        with torch.no_grad():
            img = model.spectrogram(x)
        torch.cuda.synchronize()
        if ref is None: ref = img.clone()
        elif not torch.equal(ref, img): bad += 1
    print("num_pad_frames %d: spectrogram %s, %d mismatches" % (pad, tuple(ref.shape), bad))
