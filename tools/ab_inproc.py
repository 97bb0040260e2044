"""Interleaved A/B of a construction-time switch of the engines inside ONE process (same box, same clocks):

  python tools/ab_inproc.py --set ops.SLAB_BATCH=True,False [--rounds 4] [--steps 30] pathB_f32_split pathB bf16 f32_split fp32

One engine per value of the switch (module attribute, read when the engine is built), timed round-robin; every step is a valid
train step.  Legs as tools/skip_probe.py."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "skeleton-action-recognition_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from skip_probe import make_step, timed  # noqa: E402


def main():
    argv = sys.argv[1:]
    opt = {"--rounds": "4", "--steps": "30", "--set": "ops.SLAB_BATCH=True,False"}
    legs, i = [], 0
    while i < len(argv):
        if argv[i] in opt:
            opt[argv[i]] = argv[i + 1]
            i += 2
        else:
            legs.append(argv[i])
            i += 1
    name, vals = opt["--set"].split("=", 1)
    names = name.split("+")                   # several switches: --set "ops.SLAB_BATCH+ops.SLAB_FLUSH=True,'end';True,'block';False,'end'"
    targets = [(importlib.import_module("sar_amd." + n.rsplit(".", 1)[0]), n.rsplit(".", 1)[1]) for n in names]
    values = [eval(v) for v in vals.split(";" if len(names) > 1 else ",")]
    rounds, steps = int(opt["--rounds"]), int(opt["--steps"])
    dev = torch.device("cuda:0")
    for leg in legs or ["pathB_f32_split"]:
        engines = []
        for v in values:
            vs = v if len(names) > 1 else (v,)
            old = [getattr(m, a) for m, a in targets]
            for (m, a), x in zip(targets, vs):
                setattr(m, a, x)
            try:                      # the switch holds while the engine is built, warmed and (below) timed: construction-time and
                step, bs = make_step(leg, dev)    # call-time switches alike
                for k in range(6):
                    step(k)
            finally:
                for (m, a), x in zip(targets, old):
                    setattr(m, a, x)
            engines.append(step)
        res = [[] for _ in values]
        for _ in range(rounds):
            for j, step in enumerate(engines):
                vs = values[j] if len(names) > 1 else (values[j],)
                old = [getattr(m, a) for m, a in targets]
                for (m, a), x in zip(targets, vs):
                    setattr(m, a, x)
                try:
                    timed(step, 3)
                    res[j].append(timed(step, steps))
                finally:
                    for (m, a), x in zip(targets, old):
                        setattr(m, a, x)
        base = min(res[-1])
        for v, r in zip(values, res):
            print("%-16s %s=%-16s %s ms/step  (best %.3f, %+.2f %% vs last, %.0f clips/s)" % (
                leg, name, str(v), " ".join("%.3f" % t for t in r), min(r), (min(r) / base - 1) * 100, bs / min(r) * 1e3), flush=True)
        del engines
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
