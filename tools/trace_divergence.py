#!/usr/bin/env python3
"""Which launch of the fp32 train step is not repeatable?  Every sar_amd.ops call is wrapped: after it, a 64-bit checksum of each
tensor argument is enqueued on the stream (no synchronisation inside the step); the checksum sequences of the repetitions are
compared with those of the first one and the first call whose arguments differ is reported.
Usage: python tools/trace_divergence.py [--reps 100] [--batch 64] [--mode fp32]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from sar_amd import ops, ops8  # noqa: E402
from sar_amd.stgcn import STGCN  # noqa: E402
from sar_amd.train import synthetic_clips  # noqa: E402

LOG = []


def checksum(t):
    if t.dtype == torch.float32:
        v = t.contiguous().view(torch.int32) if t.is_contiguous() else t.contiguous().view(torch.int32)
    elif t.dtype == torch.bfloat16:
        v = t.contiguous().view(torch.int16)
    else:
        v = t.contiguous()
    return v.to(torch.int64).sum() if v.numel() < (1 << 20) else v.sum(dtype=torch.int64)


def wrap(mod, name):
    fn = getattr(mod, name)

    def inner(*a, **k):
        r = fn(*a, **k)
        ts = [x for x in list(a) + list(k.values()) if isinstance(x, torch.Tensor) and x.is_cuda and x.numel() > 0]
        for x in list(a) + list(k.values()):
            if isinstance(x, (tuple, list)):
                ts += [y for y in x if isinstance(y, torch.Tensor) and y.is_cuda and y.numel() > 0]
        if isinstance(r, tuple):
            ts += [y for y in r if isinstance(y, torch.Tensor) and y.is_cuda and y.numel() > 0]
        ints = [x for x in a if isinstance(x, int)] + ["%s=%s" % (kk, vv) for kk, vv in k.items() if isinstance(vv, (int, bool))]
        LOG.append(("%s.%s %s" % (mod.__name__.split(".")[-1], name, ints), [(tuple(x.shape), checksum(x)) for x in ts]))
        return r
    setattr(mod, name, inner)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--mode", default="fp32")
    a = ap.parse_args()
    for mod in (ops, ops8):
        for name in dir(mod):
            f = getattr(mod, name)
            if callable(f) and not name.startswith("_") and getattr(f, "__module__", "") == mod.__name__ and not isinstance(f, type) \
                    and name not in ("check", "ptr", "stream_ptr", "make_bn_tail", "bn_tail_tickets", "relu_mask", "empty", "from_cn", "to_cn"):
                wrap(mod, name)
    dev = torch.device("cuda:0")
    x, y = synthetic_clips(a.batch, dev, seed=3, num_classes=60)
    eng = STGCN(num_classes=60, device=dev, seed=0, mfma=a.mode)
    state = {k: v.clone() for k, v in eng.state_dict().items()}
    ref = None
    nbad = 0
    for r in range(a.reps):
        eng.load_params(state)
        LOG.clear()
        eng.loss_and_grad(x, y)
        torch.cuda.synchronize()
        cur = [(n, [(s, int(c.item())) for s, c in ts]) for n, ts in LOG]
        if ref is None:
            ref = cur
            print("%d wrapped calls per step" % len(cur))
            continue
        for i, (p, q) in enumerate(zip(ref, cur)):
            if p != q:
                nbad += 1
                diffs = [j for j, (u, v) in enumerate(zip(p[1], q[1])) if u != v]
                print("rep %d: first divergence at call %d %s, tensor arguments %s differ (shapes %s); previous call: %s" %
                      (r, i, q[0], diffs, [q[1][j][0] for j in diffs], cur[i - 1][0] if i else "-"))
                break
    print("%d of %d repetitions diverged" % (nbad, a.reps - 1))


if __name__ == "__main__":
    main()
