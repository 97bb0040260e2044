"""Sweep of the number of reduction splits (slabs) of the weight-gradient kernels at the bench shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")): sys.path.insert(0, p)
import torch
from sar_amd import _lib as L, ops
from sar_amd.stgcn import ntu_adjacency
from tools.kernel_bench import timeit
dev = torch.device("cuda:0")
B, V = 128, 25
tabs = ops.GraphTables(ntu_adjacency().astype("float32"), dev)
for (cin, f, T, s) in [(64, 64, 300, 1), (128, 128, 150, 1), (256, 256, 75, 1)]:
    To = T // s
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn((cin, B * T * V), device=dev, generator=g)
    G = torch.randn((f, B * T * V), device=dev, generator=g)
    dU = torch.randn((f, B * To * V), device=dev, generator=g)
    sc, sh = torch.rand(f, device=dev) + 0.5, torch.randn(f, device=dev) * 0.1
    outt = torch.empty(9 * f * f + f, device=dev)
    outg = torch.empty(cin * 3 * f + 3 * f, device=dev)
    wgs_t = ((f + 63) // 64) * ((f + 31) // 32)
    wgs_g = ((f + (127 if f > 64 else 63)) // (128 if f > 64 else 64)) * ((cin + 63) // 64)
    for total in (256, 512, 768, 1024, 1536, 2048):
        nt, ng = max(1, total // wgs_t), max(1, total // wgs_g)
        mt = timeit(lambda: ops.conv_wgrad(L.SAR_CONV_TEMPORAL, G, dU, outt, B=B, V=V, T_src=T, T_out=To, Kc=f, M=f, taps=9, stride=s, pad=4,
                    pro=(sc, sh), pro_relu=True, w_stride_tap=f * f, w_stride_c=f, wsize=9 * f * f, bsize=f, nsplit=nt), 5)
        mg = timeit(lambda: ops.conv_wgrad(L.SAR_CONV_GRAPH, X, G, outg, B=B, V=V, T_src=T, T_out=T, Kc=cin, M=f, taps=3, tables=tabs,
                    w_stride_tap=f, w_stride_c=3 * f, wsize=cin * 3 * f, bsize=3 * f, nsplit=ng), 5)
        print("f=%3d workgroups %4d: temporal nsplit %4d %.3f ms | graph nsplit %4d %.3f ms" % (f, total, nt, mt, ng, mg), flush=True)
