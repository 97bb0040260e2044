#!/usr/bin/env python3
"""Run-to-run determinism of the whole train step at the bench size: the kernels use no atomics and every reduction has
a fixed order, so logits, loss and every gradient must be BITWISE identical between repetitions -- a mismatch means a
race (LDS hazard, missing barrier).  Usage: python tools/determinism_check.py [--reps 5] [--batch 64]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from sar_amd.stgcn import STGCN  # noqa: E402
from sar_amd.train import synthetic_clips  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--modes", default="fp32,bf16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    total_bad = 0
    for mode in a.modes.split(","):
        bad = 0
        if mode == "resnet":      # Path B's classifier at its bench shape (bs 32, 256 x 256 spectrograms)
            from sar_amd.resnet import ResNet18
            eng = ResNet18(num_classes=60, num_filters=64, device=dev, seed=0)
            g = torch.Generator(device=dev).manual_seed(5)
            x, y = torch.randn((32, 1, 256, 256), generator=g, device=dev), torch.randint(0, 60, (32,), generator=g, device=dev)
        elif mode == "stgin":
            from sar_amd.stgin import STGIN
            eng = STGIN(num_classes=60, device=dev, seed=0)
            x, y = synthetic_clips(a.batch, dev, seed=3, num_classes=60)
        elif mode in ("dense", "bone_motion"):     # trainable adjacency (dense contraction kernels) / bone + motion input streams
            from sar_amd.bone import NTU_BONE_PAIRS
            kw = dict(trainable_adjacency=True) if mode == "dense" else dict(bone_pairs=NTU_BONE_PAIRS, motion=True)
            eng = STGCN(num_classes=60, device=dev, seed=0, mfma="fp32", **kw)
            x, y = synthetic_clips(a.batch, dev, seed=3, num_classes=60)
        else:
            eng = STGCN(num_classes=60, device=dev, seed=0, mfma=mode)
            x, y = synthetic_clips(a.batch, dev, seed=3, num_classes=60)
        state = {k: v.clone() for k, v in eng.state_dict().items()}
        ref = None
        for r in range(a.reps):
            eng.load_params(state)
            logits, loss = eng.loss_and_grad(x, y)[:2]
            torch.cuda.synchronize()
            cur = (logits.clone(), loss.clone(), eng.grad.clone())
            if ref is None:
                ref = cur
            else:
                same = all(torch.equal(p, q) for p, q in zip(ref, cur))
                if not same:
                    bad += 1
                    d = (ref[2] - cur[2]).abs().max().item()
                    names = []
                    for k in eng.shapes:      # which tensors differ
                        o, n = eng.offsets[k], 1
                        for v in eng.shapes[k]:
                            n *= v
                        dk = (ref[2][o:o + n] - cur[2][o:o + n]).abs().max().item()
                        if dk > 0:
                            names.append("%s %.2e" % (k, dk))
                    print("%s rep %d: MISMATCH (max |grad diff| %.3e; logits equal %s) %s" %
                          (mode, r, d, torch.equal(ref[0], cur[0]), ", ".join(names[:12]) + (" ... %d tensors" % len(names) if len(names) > 12 else "")))
        print("%s: %d repetitions, loss %.6f, %s" % (mode, a.reps, ref[1].item(), "bitwise identical" if bad == 0 else "NOT deterministic (%d)" % bad))
        total_bad += bad
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
