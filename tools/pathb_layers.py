"""Per-layer kernel time of the Path B step (diagnostic): SAR_PROFILE_SHAPES=1 SAR_WGRAD_STREAM=0 python tools/pathb_layers.py [batch [mfma]]
Prints, per conv geometry, launches per step, ms per step and TFLOP/s (HIP events, side stream off so that durations are clean)."""
import os
import sys

os.environ.setdefault("SAR_PROFILE_SHAPES", "1")
os.environ.setdefault("SAR_WGRAD_STREAM", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "skeleton-action-recognition_amd"))
import torch  # noqa: E402
from sar_amd import profiler  # noqa: E402
from sar_amd.train import SpectrogramTrainer, synthetic_clips  # noqa: E402
from models.resnet import Model  # noqa: E402


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    model = Model(num_classes=60, device=dev, mfma=sys.argv[2] if len(sys.argv) > 2 else "fp32")
    trainer = SpectrogramTrainer(model, 1e-3, 1)
    x, y = synthetic_clips(bs, dev, seed=0)
    for _ in range(3):
        trainer.step(x, y, 1e-3)
    timer = profiler.KernelTimer()
    profiler.install(timer)
    torch.cuda.synchronize()
    steps = 5
    for _ in range(steps):
        trainer.step(x, y, 1e-3)
    torch.cuda.synchronize()
    profiler.install(None)
    tot = 0.0
    for tag, d in sorted(timer.summary().items(), key=lambda kv: -kv[1]["ms"]):
        ms = d["ms"] / steps
        tot += ms
        print("%-52s launches/step %4.1f  %7.3f ms/step  %6.1f TFLOP/s" % (tag, d["calls"] / steps, ms, d["flops"] / steps / ms / 1e9 if ms else 0))
    print("sum of the timed regions: %.3f ms/step" % tot)


if __name__ == "__main__":
    main()
