import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/skeleton-action-recognition_amd"); sys.path.insert(0, "/root/repo/tests")
from oracle import stgcn as O
from sar_amd.stgcn import STGCN
import test_gpu_bf16_training as TT
dev = torch.device("cuda:0")
classes, steps, bs = 10, int(os.environ.get("CURVE_STEPS", "400")), 32
batch = TT._task(dev, classes)
p = O.init_params(classes, seed=7, dtype=torch.float64)
for lr in (0.02,):
  for mode in ("fp32", "bf16", "fp32", "bf16_operands"):
    eng = STGCN(num_classes=classes, device=dev, mfma=mode)
    eng.load_params(p)
    losses, correct = [], []
    for s in range(steps):
        x, y = batch(bs, s)
        logits, loss = eng.loss_and_grad(x, y)
        eng.sgd_step(lr if s < steps * 3 // 4 else lr / 10)
        losses.append(loss.reshape(())); correct.append((logits.argmax(1) == y).float().mean())
    L = torch.stack(losses).cpu(); C = torch.stack(correct).cpu()
    print(lr, mode, "loss per 50:", [round(L[i:i+50].mean().item(), 3) for i in range(0, steps, 50)], "acc last50 %.3f" % C[-50:].mean().item())
