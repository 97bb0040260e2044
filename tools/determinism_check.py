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
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    x, y = synthetic_clips(a.batch, dev, seed=3, num_classes=60)
    bad = 0
    for mode in ("fp32", "bf16"):
        eng = STGCN(num_classes=60, device=dev, seed=0, mfma=mode)
        state = {k: v.clone() for k, v in eng.state_dict().items()}
        ref = None
        for r in range(a.reps):
            eng.load_params(state)
            logits, loss = eng.loss_and_grad(x, y)
            torch.cuda.synchronize()
            cur = (logits.clone(), loss.clone(), eng.grad.clone())
            if ref is None:
                ref = cur
            else:
                same = all(torch.equal(p, q) for p, q in zip(ref, cur))
                if not same:
                    bad += 1
                    d = (ref[2] - cur[2]).abs().max().item()
                    print("%s rep %d: MISMATCH (max |grad diff| %.3e)" % (mode, r, d))
        print("%s: %d repetitions, loss %.6f, %s" % (mode, a.reps, ref[1].item(), "bitwise identical" if bad == 0 else "NOT deterministic"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
