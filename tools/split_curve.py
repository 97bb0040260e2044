"""Long-horizon evidence for the f32_split engine (VERDICT r05 next #1d): the r03 bf16 protocol (tools/bf16_curve.py,
tests/test_gpu_bf16_training.py: the full 10-block ST-GCN on a LEARNABLE synthetic task with an irreducible error, a fresh seeded
batch every step, Nesterov SGD 0.02 -> 0.002 for the last quarter) over CURVE_STEPS (default 1 200) steps, from the same initial
weights on the same data stream, for
    fp32             the native fp32 MFMA engine (the reference's precision; the control)
    f32_split        fp32 storage, every GEMM product = three exact products of fp16 terms (f16x3a)
    f32_split_bf16x6 the six-product bf16 form
    fp32+1ulp        the fp32 engine again from weights perturbed by ONE unit in the last place of every parameter: how far two
                     runs of the SAME arithmetic drift apart on this task (SGD is chaotic: the yardstick for the rows above)
Prints the mean loss per 50 steps, the last-50 mean loss / top-1, and per mode the largest |loss - loss_fp32| over the first 20 steps
(where trajectories still coincide: the arithmetic's own error) and over all steps (trajectory drift).
    python tools/split_curve.py > profiles/r06_f32split_training_curve.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import stgcn as O                     # noqa: E402  (initial weights only)
from sar_amd.stgcn import STGCN                   # noqa: E402
import test_gpu_bf16_training as TT               # noqa: E402


def main():
    dev = torch.device("cuda:0")
    classes, steps, bs, lr = 10, int(os.environ.get("CURVE_STEPS", "1200")), 32, 0.02
    batch = TT._task(dev, classes)
    p = O.init_params(classes, seed=7, dtype=torch.float64)
    p_ulp = {k: (torch.nextafter(v.float(), torch.full_like(v.float(), float("inf"))).double() if v.is_floating_point() else v)
             for k, v in p.items()}
    curves = {}
    for name, mode, params in (("fp32", "fp32", p), ("f32_split", "f32_split", p), ("f32_split_bf16x6", "f32_split_bf16x6", p),
                               ("fp32+1ulp", "fp32", p_ulp)):
        eng = STGCN(num_classes=classes, device=dev, mfma=mode)
        eng.load_params(params)
        losses, correct = [], []
        for s in range(steps):
            x, y = batch(bs, s)
            logits, loss = eng.loss_and_grad(x, y)
            eng.sgd_step(lr if s < steps * 3 // 4 else lr / 10)
            losses.append(loss.reshape(()))
            correct.append((logits.argmax(1) == y).float().mean())
        L, C = torch.stack(losses).cpu().double(), torch.stack(correct).cpu().double()
        assert torch.isfinite(L).all(), name
        curves[name] = (L, C)
        print("%-17s loss per 50: %s" % (name, " ".join("%.3f" % L[i:i + 50].mean().item() for i in range(0, steps, 50))))
        print("%-17s last 50: loss %.4f top-1 %.4f" % (name, L[-50:].mean().item(), C[-50:].mean().item()), flush=True)
        del eng
    L0, C0 = curves["fp32"]
    print()
    print("|loss - loss_fp32| at steps 0 1 2 3 5 8 (before the trajectories separate: the arithmetic's own error, then its growth) and")
    print("over windows (SGD at this rate is chaotic from the first few steps on: the fp32+1ulp row is the yardstick for every other row)")
    print("%-17s %-62s %-12s %-12s %-12s %-10s %-10s" % ("vs fp32", "steps 0 1 2 3 5 8", "max 0-49", "max 50-599", "max 600-", "dloss@end", "dtop1@end"))
    for name, (L, C) in curves.items():
        if name == "fp32":
            continue
        d = (L - L0).abs()
        early = " ".join("%.2e" % d[i].item() for i in (0, 1, 2, 3, 5, 8) if i < steps)
        print("%-17s %-62s %-12.3e %-12.3e %-12.3e %-+10.4f %-+10.4f" % (name, early, d[:50].max().item(), d[50:600].max().item() if steps > 50 else 0.0,
                                                                      d[600:].max().item() if steps > 600 else 0.0,
                                                                      (L[-50:].mean() - L0[-50:].mean()).item(), (C[-50:].mean() - C0[-50:].mean()).item()))


if __name__ == "__main__":
    main()
