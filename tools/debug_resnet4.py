import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch
from oracle import resnet as RN
from sar_amd import resnet as RS, ops
from util import rel_err
dev = torch.device("cuda:0")
eng = RS.ResNet18(num_classes=60, num_filters=64, device=dev, seed=3)
p = {k: v.double() for k, v in eng.state_dict().items()}
g = torch.Generator().manual_seed(5)
x = torch.randn(2, 1, 256, 256, generator=g) * 3 - 4
y = torch.tensor([7, 33])
_, _, gref, _, _ = RN.loss_and_grads(p, x.double(), y)
def run(tag):
    eng.loss_and_grad(x.to(dev), y.to(dev)); torch.cuda.synchronize()
    print("%-28s bn2.bias(l2.1) %.2e  conv2.w(l2.1) %.2e  l1.0.conv1.w %.2e  l3.0.conv1.w %.2e" % (tag, rel_err(eng.g["layer2.1.bn2.bias"].cpu(), gref["layer2.1.bn2.bias"]), rel_err(eng.g["layer2.1.conv2.weight"].cpu(), gref["layer2.1.conv2.weight"]), rel_err(eng.g["layer1.0.conv1.weight"].cpu(), gref["layer1.0.conv1.weight"]), rel_err(eng.g["layer3.0.conv1.weight"].cpu(), gref["layer3.0.conv1.weight"])))
run("plain #1"); run("plain #2")
orig_d, orig_w = RS.ResNet18._conv_dgrad, RS.ResNet18._conv_wgrad
def d_sync(self, *a, **k):
    r = orig_d(self, *a, **k); torch.cuda.synchronize(); return r
RS.ResNet18._conv_dgrad = d_sync; run("sync after dgrad")
RS.ResNet18._conv_dgrad = orig_d
def w_sync(self, *a, **k):
    r = orig_w(self, *a, **k); torch.cuda.synchronize(); return r
RS.ResNet18._conv_wgrad = w_sync; run("sync after wgrad")
RS.ResNet18._conv_wgrad = orig_w
keepalive = []
def d_keep(self, *a, **k):
    r = orig_d(self, *a, **k); keepalive.append(r); return r
RS.ResNet18._conv_dgrad = d_keep; run("keep dgrad outputs alive"); keepalive.clear()
RS.ResNet18._conv_dgrad = orig_d
orig_cw = ops.conv2d_wgrad
slabs = []
def cw_keep(src, dout, dW, **geo):
    import torch as T
    r = orig_cw(src, dout, dW, **geo); return r
# keep wgrad slabs alive by monkeypatching torch.empty inside ops? simpler: disable caching reuse via sync+empty_cache
run("plain #3")
