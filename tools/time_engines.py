"""Train-step time of the engines at a given batch size (diagnostic): python tools/time_engines.py stgcn stgin [--bs 64]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "skeleton-action-recognition_amd"))
import torch  # noqa: E402
from sar_amd.train import synthetic_clips  # noqa: E402


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["stgcn", "stgin"]
    bs = int(sys.argv[sys.argv.index("--bs") + 1]) if "--bs" in sys.argv else 64
    dev = torch.device("cuda:0")
    for name in names:
        if name == "stgin":
            from sar_amd.stgin import STGIN as cls
        else:
            from sar_amd.stgcn import STGCN as cls
        eng = cls(num_classes=60, device=dev)
        x, y = synthetic_clips(bs, dev, seed=0, num_classes=60)
        for _ in range(5):
            eng.loss_and_grad(x, y)
            eng.sgd_step(0.1)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(10):
            eng.loss_and_grad(x, y)
            eng.sgd_step(0.1)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 10
        print("%s bs %d: %.1f ms/step, %.0f clips/s, peak memory %.1f GB" % (name, bs, dt * 1e3, bs / dt,
                                                                          torch.cuda.max_memory_allocated() / 2**30), flush=True)
        del eng
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
