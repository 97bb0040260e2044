import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "skeleton-action-recognition_amd"))
from sar_amd import ops, _lib as L
dev = torch.device("cuda:0")
def run(cin, H, B=32):
    cout = cin
    geo = dict(B=B, Kc=cin, M=cout, H_src=H, W_src=H, H_out=H, W_out=H, KH=3, KW=3, stride=1, pad=1)
    x = torch.randn(cin, B * H * H, device=dev)
    w = torch.randn(9 * cin * cout, device=dev) * 0.02
    out = torch.empty(cout, B * H * H, device=dev)
    pk, wb = ops._pack_split_conv2d(w, cin * cout, cout, cin, cout, "f16x3a", False)
    cell = torch.zeros(1, dtype=torch.int32, device=dev); ops.amax(x, cell)
    f = lambda: ops.conv2d_gemm(x, out, w, cin * cout, cout, epi=L.SAR_EPI_STATS, split="f16x3a", packed=pk, bounds=(cell, wb), **geo)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for cin, H in ((512, 8), (256, 16), (128, 32), (64, 64)):
    print("Kc%d %dx%d: %.1f us" % (cin, H, H, run(cin, H)), flush=True)
