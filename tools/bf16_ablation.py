#!/usr/bin/env python3
"""Which stored tensor owns the angle between the bf16 network's gradient and the float64 gradient?

The float64 oracle (oracle/stgcn.py) is run at the full NTU shape with bfloat16 STORAGE emulated at chosen sites (value
rounded in the forward pass, gradient rounded in the backward pass, straight-through): all sites = the bf16 configuration;
all but one = that tensor kept in fp32.  Printed: the gradient cosine against the plain float64 gradient for block 0's
kernels (the worst, furthest from the loss) and for blocks 8-9.  CPU only (test infrastructure, not the product path).
    python tools/bf16_ablation.py [N clips, default 2]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import stgcn as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
torch.set_num_threads(os.cpu_count() or 4)
blocks = list(O.BLOCKS)
p = O.randomize_affine(O.init_params(60, seed=3, dtype=torch.float64, blocks=blocks), seed=4)
x, y = O.synthetic_batch(N, seed=3, T=T, num_classes=60)
t0 = time.time()
_, loss_ref, g_ref, _, _ = O.loss_and_grads(p, x.double(), y, blocks=blocks)
print("float64 reference: loss %.6f (%.0f s)" % (loss_ref.item(), time.time() - t0), flush=True)
ALL = {"x0", "g", "h", "u", "r", "y", "w"}


def cosines(quant):
    _, loss, g, _, _ = O.loss_and_grads(p, x.double(), y, blocks=blocks, quant=quant)
    out = {}
    for grp, keys in (("block0", [k for k in g if k.startswith("l0.") and k.endswith("kernel")]),
                      ("blocks8-9", [k for k in g if k.startswith(("l8.", "l9.")) and k.endswith("kernel")]),
                      ("all kernels", [k for k in g if k.endswith("kernel")])):
        out[grp] = min(((g[k] * g_ref[k]).sum() / (g[k].norm() * g_ref[k].norm())).item() for k in keys)
    return loss.item(), out


rows = [("bf16 configuration (all sites)", ALL)]
rows += [("all but %-2s (kept fp32)" % s, ALL - {s}) for s in ("g", "u", "y", "h", "w", "x0", "r")]
rows += [("only %-2s rounded" % s, {s}) for s in ("g", "u", "y", "h", "w")]
rows += [("values only (gradients fp32)", {s + ":fwd" for s in ALL})]
for name, q in rows:
    t0 = time.time()
    loss, c = cosines(q)
    print("%-34s loss %.6f  cos block0 %.4f  blocks8-9 %.4f  worst kernel %.4f  (%.0f s)"
          % (name, loss, c["block0"], c["blocks8-9"], c["all kernels"], time.time() - t0), flush=True)
