import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch, torch.nn.functional as F
from sar_amd import resnet as RS, ops, _lib as L
from util import rel_err
dev = torch.device("cuda:0")
eng = RS.ResNet18(num_classes=60, num_filters=64, device=dev, seed=3)
g = torch.Generator().manual_seed(5)
x = torch.randn(2, 1, 256, 256, generator=g) * 3 - 4
y = torch.tensor([7, 33])
cap = {}
orig = RS.ResNet18._conv_dgrad
def patched(self, name, dout, B, H, W, Ho, Wo, **epi):
    dx, r = orig(self, name, dout, B, H, W, Ho, Wo, **epi)
    torch.cuda.synchronize()
    cap[name] = dict(dout=dout.clone(), dx=dx.clone(), aux=None if epi.get("aux") is None else epi["aux"].clone(), geo=(B, H, W, Ho, Wo), epi=epi.get("epi"))
    return dx, r
RS.ResNet18._conv_dgrad = patched
eng.loss_and_grad(x.to(dev), y.to(dev)); torch.cuda.synchronize()
for name in ["layer4.0.conv1", "layer3.0.downsample.0", "layer3.0.conv1", "layer2.1.conv1", "layer2.0.conv1"]:
    c = cap[name]; cv = eng.convs[name]; B, H, W, Ho, Wo = c["geo"]
    w = eng.p[name + ".weight"].cpu().double()
    dout = c["dout"].cpu().double().view(cv.cout, B, Ho, Wo).permute(1, 0, 2, 3)
    xx = torch.zeros(B, cv.cin, H, W, dtype=torch.float64, requires_grad=True)
    yy = F.conv2d(xx, w, None, stride=cv.stride, padding=cv.pad)
    (gx,) = torch.autograd.grad(yy, xx, dout)
    exp = gx.permute(1, 0, 2, 3).reshape(cv.cin, -1)
    if c["epi"] == L.SAR_EPI_ADD: exp = exp + c["aux"].cpu().double()
    got = c["dx"].cpu().double()
    e = (got - exp).abs()
    print("%-24s epi %s  rel err %.3e   (max|exp| %.3e)  bad elems %d / %d" % (name, c["epi"], e.max() / exp.abs().max(), exp.abs().max(), (e > 1e-3 * exp.abs().max()).sum(), e.numel()))
    if e.max() / exp.abs().max() > 1e-3:
        idx = (e > 1e-3 * exp.abs().max()).nonzero()
        ch, pos = idx[:, 0], idx[:, 1]
        b = pos // (H * W); hh = (pos % (H * W)) // W; ww = pos % W
        print("   bad channels", ch.unique()[:20].tolist(), " rows", hh.unique()[:40].tolist(), " cols", ww.unique()[:40].tolist(), " images", b.unique().tolist())
        # rerun the same call in isolation
        dx2 = torch.empty_like(c["dx"])
        kw = dict(epi=c["epi"], aux=c["aux"]) if c["epi"] == L.SAR_EPI_ADD else {}
        ops.conv2d_gemm(c["dout"], dx2, eng._packed[name][1], cv.cout * cv.cin, cv.cin, B=B, Kc=cv.cout, M=cv.cin, H_src=Ho, W_src=Wo, H_out=H, W_out=W, KH=cv.k, KW=cv.k, stride=cv.stride, pad=cv.pad, transposed=True, **kw)
        torch.cuda.synchronize()
        print("   isolated rerun rel err %.3e" % ((dx2.cpu().double() - exp).abs().max() / exp.abs().max()))
