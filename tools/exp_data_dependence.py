import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")): sys.path.insert(0, p)
import torch
from sar_amd import _lib as L, ops
from sar_amd.stgcn import STGCN
from sar_amd.train import synthetic_clips
from tools.kernel_bench import timeit
dev = torch.device("cuda:0")
eng = STGCN(60, device=dev)
x, y = synthetic_clips(64, dev, seed=0)
keep = {}
eng.forward(x, True, keep)
B, V = 128, 25
def run(name, G, f, T, s, sc, sh, W, b):
    To = -(-T // s); pad = 4 if s == 1 else 3
    out = torch.empty((f, B * To * V), device=dev)
    ms = timeit(lambda: ops.conv_gemm(L.SAR_CONV_TEMPORAL, G, out, W, f * f, f, B=B, V=V, T_src=T, T_out=To, Kc=f, M=f, taps=9,
                stride=s, pad=pad, bias=b, pro=(sc, sh), pro_relu=True, epi=L.SAR_EPI_STATS), 5)
    print("%-40s %.3f ms" % (name, ms), flush=True)
i = 8; f = 256; T = 75
g_real = keep["l%d.g" % i]
bn = eng.bn["l%d.bn1" % i]
W, b = eng.p["l%d.tcn.kernel" % i], eng.p["l%d.tcn.bias" % i]
gen = torch.Generator(device=dev).manual_seed(0)
g_rand = torch.randn(g_real.shape, device=dev, generator=gen)
W_rand = torch.randn(W.shape, device=dev, generator=gen) * 0.05
one, zero = torch.ones(f, device=dev), torch.zeros(f, device=dev)
print("g_real stats: mean %.3g std %.3g absmin-nonzero %.3g frac|x|<1e-30 %.3g" % (g_real.mean().item(), g_real.std().item(), g_real[g_real != 0].abs().min().item(), (g_real.abs() < 1e-30).float().mean().item()))
print("scale/shift", bn.scale.abs().min().item(), bn.scale.abs().max().item(), bn.shift.abs().max().item())
run("real g, real bn, real W", g_real, f, T, 1, bn.scale, bn.shift, W, b)
run("rand g, real bn, real W", g_rand, f, T, 1, bn.scale, bn.shift, W, b)
run("real g, unit bn, real W", g_real, f, T, 1, one, zero, W, b)
run("real g, real bn, rand W", g_real, f, T, 1, bn.scale, bn.shift, W_rand, b)
run("rand g, unit bn, rand W", g_rand, f, T, 1, one, zero, W_rand, b)
run("zero g, unit bn, rand W", torch.zeros_like(g_rand), f, T, 1, one, zero, W_rand, b)
# same data, copied into a fresh contiguous buffer
run("real g (clone), real bn, real W", g_real.clone(), f, T, 1, bn.scale.clone(), bn.shift.clone(), W.clone(), b.clone())
print("W is view of flat: storage_offset", W.storage_offset(), "ptr%16 =", W.data_ptr() % 16, "g ptr%256 =", g_real.data_ptr() % 256)
