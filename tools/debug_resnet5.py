import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch, torch.nn.functional as F
from oracle import resnet as RN
from sar_amd import resnet as RS, ops
dev = torch.device("cuda:0")
eng = RS.ResNet18(num_classes=60, num_filters=64, device=dev, seed=3)
p = {k: v.double() for k, v in eng.state_dict().items()}
g = torch.Generator().manual_seed(5)
x = torch.randn(2, 1, 256, 256, generator=g) * 3 - 4
y = torch.tensor([7, 33])
taps = {}
pp = dict(p)
lg = RN.forward(pp, x.double(), True, None, taps)
calls = []
orig = ops.bn_add_relu_bwd_reduce
def patched(dy, yy, u, r, mu=None, mr=None):
    part, nparts = orig(dy, yy, u, r, mu, mr)
    torch.cuda.synchronize()
    calls.append(dict(dy=dy.clone(), y=yy.clone(), u=u.clone(), mu=mu.clone(), part=part.clone(), nparts=nparts))
    return part, nparts
ops.bn_add_relu_bwd_reduce = patched
RS.ops.bn_add_relu_bwd_reduce = patched
eng.loss_and_grad(x.to(dev), y.to(dev)); torch.cuda.synchronize()
names = [b[0] for b in reversed(eng.blocks)]
for name, c in zip(names, calls):
    dy, yy, u = c["dy"].cpu().double(), c["y"].cpu().double(), c["u"].cpu().double()
    dz = dy * (yy > 0)
    s1 = dz.sum(1); s2 = (dz * (u - c["mu"].cpu().double()[:, None])).sum(1)
    ps = c["part"].cpu().double().sum(1)
    H = int((yy.shape[1] // 2) ** 0.5)
    yo = taps[name + "out"].permute(1, 0, 2, 3).reshape(yy.shape[0], -1)
    print("%-10s n=%6d nparts=%d  s1 err %.2e (|s1|max %.2e)  s2 err %.2e   y vs oracle %.2e   frac(y>0) %.3f  mask mismatches vs oracle %d" % (
        name, yy.shape[1], c["nparts"], (ps[:, 0] - s1).abs().max() / s1.abs().max(), s1.abs().max(), (ps[:, 1] - s2).abs().max() / s2.abs().max(),
        (yy - yo).abs().max() / yo.abs().max(), (yy > 0).double().mean(), ((yy > 0) != (yo > 0)).sum()))
