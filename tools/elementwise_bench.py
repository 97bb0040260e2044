"""Bandwidth of the row-wise element-wise kernels at the bench shapes, in isolation (vs their in-step rocprof time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")): sys.path.insert(0, p)
import torch
from sar_amd import ops
from tools.kernel_bench import timeit
dev = torch.device("cuda:0")
for C, n in [(64, 960000), (128, 480000), (256, 240000)]:
    g = torch.Generator(device=dev).manual_seed(0)
    a, b, y, u = (torch.randn((C, n), device=dev, generator=g) for _ in range(4))
    out, out2 = torch.empty_like(a), torch.empty_like(a)
    k = tuple(torch.randn(C, device=dev, generator=g) for _ in range(3))
    gb = C * n * 4 / 1e9
    ms = timeit(lambda: ops.affine2(a, b, k, out), 5); print("C=%3d affine2 out-of-place   %.3f ms %.2f TB/s" % (C, ms, 3 * gb / ms))
    ms = timeit(lambda: ops.affine2(a, b, k, a), 5); print("C=%3d affine2 in place       %.3f ms %.2f TB/s" % (C, ms, 3 * gb / ms))
    ms = timeit(lambda: ops.bn_add_relu_fwd(u, k[0], k[1], 1, b, None, None, out), 5); print("C=%3d bn_add_relu_fwd(ident) %.3f ms %.2f TB/s" % (C, ms, 3 * gb / ms))
    mean = k[2]
    ms = timeit(lambda: ops.bn_add_relu_bwd_reduce(a, y, u, None, mean, None), 5); print("C=%3d bwd_reduce             %.3f ms %.2f TB/s" % (C, ms, 3 * gb / ms))
    ms = timeit(lambda: ops.bn_add_relu_bwd_apply(a, y, u, None, k, None, out, None, out2), 5); print("C=%3d bwd_apply (3 in, 2 out)  %.3f ms %.2f TB/s" % (C, ms, 5 * gb / ms))

# CN8 (bf16 storage) versions
from sar_amd import ops8
for C, n in [(64, 960000), (128, 480000), (256, 240000)]:
    g = torch.Generator(device=dev).manual_seed(0)
    f = [torch.randn((C, n), device=dev, generator=g) for _ in range(3)]
    a8, y8, u8 = (ops8.from_cn(t) for t in f)
    mean = torch.randn(C, device=dev, generator=g)
    gb = C * n * 2 / 1e9
    ms = timeit(lambda: ops8.bn_add_relu_bwd_reduce(a8, y8, u8, None, C, mean, None), 5); print("C=%3d cn8 bwd_reduce         %.3f ms %.2f TB/s" % (C, ms, 3 * gb / ms))
