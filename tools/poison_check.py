#!/usr/bin/env python3
"""Uninitialised-read detector: every torch.empty / empty_like / new_empty of the engines is filled with NaN (floats) or 0xFF
(bytes) before use; a train step whose results then differ from the un-poisoned step (or contain NaN) read memory it never
wrote.  Usage: python tools/poison_check.py [--modes fp32,bf16] [--batch 8]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

_empty, _empty_like = torch.empty, torch.empty_like
POISON = [False]


def _fill(t):
    if POISON[0] and t.is_cuda:
        if t.dtype in (torch.float32, torch.float64, torch.bfloat16, torch.float16):
            t.fill_(float("nan"))
        elif t.dtype in (torch.uint8, torch.int32, torch.int64):
            t.fill_(0x7f)
    return t


torch.empty = lambda *a, **k: _fill(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _fill(_empty_like(*a, **k))

from sar_amd.stgcn import STGCN  # noqa: E402
from sar_amd.train import synthetic_clips  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="fp32,bf16")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    x, y = synthetic_clips(a.batch, dev, seed=3, num_classes=60)
    rc = 0
    for mode in a.modes.split(","):
        eng = STGCN(num_classes=60, device=dev, seed=0, mfma=mode)
        state = {k: v.clone() for k, v in eng.state_dict().items()}
        res = []
        for poison in (False, True, True):
            POISON[0] = poison
            eng.load_params(state)
            eng.grad.zero_()
            logits, loss = eng.loss_and_grad(x, y)
            torch.cuda.synchronize()
            POISON[0] = False
            res.append((logits.clone(), eng.grad.clone()))
        for i in (1, 2):
            bad = []
            if not torch.equal(res[0][0], res[i][0]):
                bad.append("logits (nan: %s)" % bool(torch.isnan(res[i][0]).any()))
            for k in eng.shapes:
                o, n = eng.offsets[k], 1
                for v in eng.shapes[k]:
                    n *= v
                if not torch.equal(res[0][1][o:o + n], res[i][1][o:o + n]):
                    bad.append("%s (nan: %s)" % (k, bool(torch.isnan(res[i][1][o:o + n]).any())))
            print("%s poisoned run %d: %s" % (mode, i, "identical to the clean run" if not bad else "DIFFERS: " + ", ".join(bad[:10]) + (" ... %d" % len(bad) if len(bad) > 10 else "")))
            rc |= 1 if bad else 0
    sys.exit(rc)


if __name__ == "__main__":
    main()
