"""A/B of the engine's arithmetic modes: every gradient of one full-shape train step (N = 2) against the fp32 engine's.
SAR_SPLIT_KINDS=tfwd,tdgrad,twgrad,gfwd,gdgrad,gwgrad selects the kernel families a split engine converts.
Usage: python tools/mode_cmp.py f32_split [f32_split_bf16x6 ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skeleton-action-recognition_amd"))
from oracle import stgcn as O
from sar_amd.stgcn import STGCN
dev = torch.device("cuda:0")
blocks = list(O.BLOCKS)
p = O.randomize_affine(O.init_params(60, seed=3, dtype=torch.float64, blocks=blocks), seed=4)
x, y = O.synthetic_batch(2, seed=3, T=300, num_classes=60)
def run(mode):
    eng = STGCN(num_classes=60, device=dev, blocks=blocks, mfma=mode)
    eng.load_params(p)
    eng.loss_and_grad(x.to(dev), y.to(dev))
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in eng.g.items()}
ref = run("fp32")
for mode in sys.argv[1:]:
    g = run(mode)
    errs = {k: ((g[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-30)).item() for k in ref if ref[k].abs().max() > 1e-9 and not k.endswith(('tcn.bias', 'res.bias'))}
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print(mode, os.environ.get("SAR_SPLIT_KINDS"), " ".join("%s=%.2e" % kv for kv in top))
