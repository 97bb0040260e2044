import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")): sys.path.insert(0, p)
import torch
from sar_amd import _lib as L, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def run(B, C, F, T, s):
    V = 25
    To = -(-T // s)
    pad_total = max((To - 1) * s + 9 - T, 0); pad = pad_total // 2
    x = torch.randn(C, B * T * V, device=dev)
    dy = torch.randn(F, B * To * V, device=dev)
    out = torch.zeros(9 * C * F + F, device=dev)
    ops.conv_wgrad(L.SAR_CONV_TEMPORAL, x, dy, out, B=B, V=V, T_src=T, T_out=To, Kc=C, M=F, taps=9, stride=s, pad=pad,
                   w_stride_tap=C * F, w_stride_c=F, wsize=9 * C * F, bsize=F)
    torch.cuda.synchronize()
    xx = x.reshape(C, B, T, V).permute(1, 0, 2, 3).double().cpu()
    dd = dy.reshape(F, B, To, V).permute(1, 0, 2, 3).double().cpu()
    xp = torch.nn.functional.pad(xx, (0, 0, pad, pad_total - pad))
    ref = torch.zeros(9, C, F, dtype=torch.float64)
    for tp in range(9):
        xs = xp[:, :, tp:tp + (To - 1) * s + 1:s, :]
        ref[tp] = torch.einsum("bctv,bftv->cf", xs, dd)
    got = out[:9 * C * F].reshape(9, C, F).double().cpu()
    err = (got - ref).abs().amax(dim=(1, 2)) / ref.abs().max()
    print("B=%d C=%d F=%d T=%d s=%d  per-tap rel err:" % (B, C, F, T, s), " ".join("%.1e" % e for e in err.tolist()))
run(2, 64, 64, 14, 2); run(2, 64, 64, 16, 2); run(2, 64, 64, 13, 1); run(3, 128, 128, 150, 2); run(1, 64, 64, 300, 2)
