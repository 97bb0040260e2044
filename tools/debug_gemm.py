import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")): sys.path.insert(0, p)
import torch
from sar_amd import _lib as L, ops
dev = torch.device("cuda:0")
B, T, V, C = 1, 8, 25, 64
x = torch.arange(C * B * T * V, device=dev, dtype=torch.float32).reshape(C, -1) * 1e-3
W = torch.zeros(9, C, C, device=dev)
W[4] = torch.eye(C, device=dev)
out = torch.full((C, B * T * V), -7.0, device=dev)
ops.conv_gemm(L.SAR_CONV_TEMPORAL, x, out, W, C * C, C, B=B, V=V, T_src=T, T_out=T, Kc=C, M=C, taps=9, stride=1, pad=4)
torch.cuda.synchronize()
print("max err identity", (out - x).abs().max().item())
print(out[:3, :6]); print(x[:3, :6])
d = (out - x).abs()
bad = (d > 1e-4).nonzero()
print("bad count", bad.shape[0], bad[:10].tolist())
W[4] = 0; W[0] = torch.eye(C, device=dev)   # out[t] = x[t-4]
ops.conv_gemm(L.SAR_CONV_TEMPORAL, x, out, W, C * C, C, B=B, V=V, T_src=T, T_out=T, Kc=C, M=C, taps=9, stride=1, pad=4)
ref = torch.zeros_like(x); ref[:, 4 * V:] = x[:, :-4 * V]
print("shift err", (out - ref).abs().max().item())
