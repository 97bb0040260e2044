import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch, torch.nn.functional as F
from oracle import resnet as RN
from sar_amd import resnet as RS, ops
from util import rel_err
dev = torch.device("cuda:0")
eng = RS.ResNet18(num_classes=60, num_filters=64, device=dev, seed=3)
p = {k: v.double() for k, v in eng.state_dict().items()}
g = torch.Generator().manual_seed(5)
x = torch.randn(2, 1, 256, 256, generator=g) * 3 - 4
y = torch.tensor([7, 33])
taps = {}
logits = RN.forward(p, x.double().requires_grad_(False), True, None, taps)
for v in taps.values(): v.retain_grad() if v.requires_grad else None
# oracle intermediate gradients: rebuild with leaves at the taps
names = ["layer2.0.out", "layer2.1.out", "layer3.0.out"]
taps2 = {}
pp = {k: v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("running_mean") and not k.endswith("running_var") else v for k, v in p.items()}
lg = RN.forward(pp, x.double(), True, None, taps2)
loss = F.cross_entropy(lg, y)
gr = torch.autograd.grad(loss, [taps2[n] for n in names])
ref = dict(zip(names, gr))
# engine: monkeypatch _conv_dgrad to capture outputs
captured = {}
orig = RS.ResNet18._conv_dgrad
def patched(self, name, dout, B, H, W, Ho, Wo, **epi):
    dx, r = orig(self, name, dout, B, H, W, Ho, Wo, **epi)
    captured[name] = (dx, epi.get("epi"))
    return dx, r
RS.ResNet18._conv_dgrad = patched
eng.loss_and_grad(x.to(dev), y.to(dev)); torch.cuda.synchronize()
def nchw(t, H): return t.cpu().view(t.shape[0], 2, H, H).permute(1, 0, 2, 3)
for name, tapname, H in [("layer3.0.conv1", "layer2.1.out", 32), ("layer3.1.conv1", "layer3.0.out", 16), ("layer2.1.conv1", "layer2.0.out", 32)]:
    dx = captured[name][0]
    r = ref[tapname]
    e = (nchw(dx, H).double() - r)
    print(name, "-> d", tapname, "rel err %.3e" % (e.abs().max() / r.abs().max()), " mean err per channel (first 4):", e.mean(dim=(0,2,3))[:4].tolist(), " ref mean abs", r.abs().mean().item())
    # where is the error? per-row / per-col profile
    print("   err by row h (max over others):", [round(v, 9) for v in e.abs().amax(dim=(0,1,3))[:8].tolist()], "...", [round(v,9) for v in e.abs().amax(dim=(0,1,3))[-4:].tolist()])
    print("   err by col w:", [round(v, 9) for v in e.abs().amax(dim=(0,1,2))[:8].tolist()], "...", [round(v,9) for v in e.abs().amax(dim=(0,1,2))[-4:].tolist()])
