"""Isolated time of sar_bn_bwd_finalize_f32 at the partial counts the fp32 ST-GCN step produces (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")): sys.path.insert(0, p)
import torch
from sar_amd import ops
from tools.kernel_bench import timeit
dev = torch.device("cuda:0")
for C, nparts, stride in [(64, 15360, 2), (128, 7680, 2), (256, 3840, 2), (64, 118, 4), (256, 30, 4), (64, 3750 * 4, 2)]:
    part = torch.randn((C, nparts, stride), device=dev)
    gamma, mean, rstd = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dg, db, k1, k2, k3 = (torch.empty(C, device=dev) for _ in range(5))
    ms = timeit(lambda: ops.bn_bwd_finalize(part, nparts, nparts * stride, stride, 0, 1, C, 1e6, gamma, mean, rstd, dg, db, k1, k2, k3), 20)
    print("C=%3d nparts=%5d stride %d: %.1f us" % (C, nparts, stride, ms * 1e3))
