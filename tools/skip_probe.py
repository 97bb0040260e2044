"""Upper bound of what folding the small launches of a train step into their producers could buy (diagnostic; results of the probed
steps are WRONG -- the skipped launches leave their outputs at the previous step's values -- only the step time is read).

  python tools/skip_probe.py pathB_f32_split pathB bf16 f32_split [--rounds 3]

For each leg: interleaved rounds of (nothing skipped | one group skipped | all groups skipped), 30 steps each after a warm-up of
valid steps.  Groups: finalize (sar_bn_finalize_f32), bwdfinalize (sar_bn_bwd_finalize_f32), bound (sar_bn_bound_f32 /
sar_affine_bound_f32), slab (sar_slab_reduce_f32)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "skeleton-action-recognition_amd"))
import torch  # noqa: E402
from sar_amd import _lib as L  # noqa: E402
from sar_amd.train import synthetic_clips  # noqa: E402

GROUPS = {
    "finalize": ["sar_bn_finalize_f32"],
    "bwdfinalize": ["sar_bn_bwd_finalize_f32"],
    "bound": ["sar_bn_bound_f32", "sar_affine_bound_f32"],
    "slab": ["sar_slab_reduce_f32"],
}


class SkipLib:
    """proxy of the ctypes library: a symbol in `skip` returns 0 without launching"""

    def __init__(self, lib):
        self._lib, self.skip = lib, set()

    def __getattr__(self, name):
        if name in self.skip:
            return lambda *a: 0
        return getattr(self._lib, name)


def make_step(leg, dev):
    if leg.startswith("pathB"):
        from sar_amd.train import SpectrogramTrainer
        from models.resnet import Model
        mfma = "f32_split" if leg.endswith("f32_split") else "fp32"
        model = Model(num_classes=60, num_filters=64, device=dev, num_pad_frames=0, mfma=mfma)
        trainer = SpectrogramTrainer(model, 1e-3)
        batches = [synthetic_clips(32, dev, seed=i, num_classes=60) for i in range(4)]
        return (lambda i: trainer.step(*batches[i % 4], 1e-3)), 32
    from sar_amd.stgcn import STGCN
    eng = STGCN(num_classes=60, device=dev, mfma={"bf16": "bf16", "f32_split": "f32_split", "fp32": "fp32"}[leg])
    batches = [synthetic_clips(64, dev, seed=i, num_classes=60) for i in range(4)]

    def step(i):
        eng.loss_and_grad(*batches[i % 4])
        eng.sgd_step(0.1)
    return step, 64


def timed(step, n):
    torch.cuda.synchronize()
    t = time.time()
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3


def main():
    legs = [a for a in sys.argv[1:] if not a.startswith("--")] or ["pathB_f32_split"]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
    dev = torch.device("cuda:0")
    real = L.load()
    proxy = SkipLib(real)
    L.load = lambda: proxy          # every ops.* wrapper asks L.load() per call
    for leg in legs:
        step, bs = make_step(leg, dev)
        for i in range(8):
            step(i)
        settings = [("none", [])] + [(g, GROUPS[g]) for g in GROUPS] + [("all", sum(GROUPS.values(), []))]
        res = {name: [] for name, _ in settings}
        for _ in range(rounds):
            for name, syms in settings:
                proxy.skip = set(syms)
                timed(step, 3)
                res[name].append(timed(step, 30))
        proxy.skip = set()
        base = min(res["none"])
        for name, _ in settings:
            best = min(res[name])
            print("%-16s skip %-12s %s ms/step  (best %.3f, %+.1f %% vs none, %.0f clips/s)" % (
                leg, name, " ".join("%.3f" % v for v in res[name]), best, (best / base - 1) * 100, bs / best * 1e3), flush=True)
        del step
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
