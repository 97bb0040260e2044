import sys, os, torch
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/skeleton-action-recognition_amd"); sys.path.insert(0, ROOT+"/tests")
from oracle import stgcn as O
from sar_amd.stgcn import STGCN
dev=torch.device("cuda:0")
blocks = list(O.BLOCKS) if len(sys.argv) > 1 else [(64, 1, False), (64, 1, True), (128, 2, True), (128, 1, True)]
p = O.randomize_affine(O.init_params(10, seed=0, dtype=torch.float64, blocks=blocks))
x, y = O.synthetic_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 4, seed=0, T=300 if len(sys.argv) > 1 else 40, num_classes=10)
out={}
for mode in ("fp32","bf16","bf16_operands"):
    eng = STGCN(num_classes=10, device=dev, blocks=blocks, mfma=mode)
    eng.load_params(p)
    logits, loss = eng.loss_and_grad(x.to(dev), y.to(dev))
    torch.cuda.synchronize()
    out[mode]=(logits.cpu().double(), loss.item(), {k: v.cpu().double().clone() for k, v in eng.g.items()})
for k,g32 in out["fp32"][2].items():
    a=out["bf16"][2][k]; b=out["bf16_operands"][2][k]
    cos=lambda u,v: ((u*v).sum()/(u.norm()*v.norm()+1e-300)).item()
    print("%-22s n=%6d |g|=%.3e cos(cn8)=%.4f cos(operands)=%.4f  relmax cn8 %.2e" % (k, g32.numel(), g32.abs().max().item(), cos(g32,a), cos(g32,b), ((a-g32).abs().max()/(g32.abs().max()+1e-300)).item()))
print("logits", (out["bf16"][0]-out["fp32"][0]).abs().max().item()/out["fp32"][0].abs().max().item(), out["bf16"][1], out["fp32"][1])
