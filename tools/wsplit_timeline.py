"""Per-workgroup timeline of conv_wgrad_split_kernel (-DSAR_SPLIT_TL build of conv_wgrad_split.hip): mean cycles per phase, summed
over the workgroup's tiles.  Usage: SAR_HIP_LIB=tools/bin/libsar_wsplit_tl.so python tools/wsplit_timeline.py"""
import sys, os, ctypes, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/skeleton-action-recognition_amd")
from sar_amd import ops, _lib as L
dev = torch.device("cuda:0")
lib = L.load()
B, V = 128, 25
NW = 16384
buf = np.zeros((NW, 16), dtype=np.uint32)
ptr = buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint))
lib.sar_debug_wsplit_timeline.argtypes = [ctypes.POINTER(ctypes.c_uint), ctypes.c_int, ctypes.c_int]
for (f, T) in [(64, 300), (128, 150), (256, 75)]:
    n = B * T * V
    g = torch.Generator(device=dev).manual_seed(9)
    rn = lambda *sh: torch.randn(sh, device=dev, generator=g)
    G, U = rn(f, n), rn(f, n) * 1e-5
    sc, sh = 1 + 0.1 * rn(f), 0.1 * rn(f)
    flat = torch.empty(9 * f * f + f, device=dev)
    bG, bU = ops._src_bound_single(G, (sc, sh)), ops._src_bound_single(U, None)
    fn = lambda: ops.conv_wgrad(L.SAR_CONV_TEMPORAL, G, U, flat, B=B, V=V, T_src=T, T_out=T, Kc=f, M=f, taps=9, stride=1, pad=4,
                                pro=(sc, sh), pro_relu=True, w_stride_tap=f * f, w_stride_c=f, wsize=9 * f * f, bsize=f,
                                split="f16x3a", bounds=(bG, bU))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.sar_debug_wsplit_timeline(None, 0, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    lib.sar_debug_wsplit_timeline(ptr, NW, 0)
    rows = buf[buf[:, 1] != 0].astype(np.int64)
    life = (rows[:, 1] - rows[:, 0]) * 10.0
    ph = rows[:, 4:14].mean(axis=0)
    tiles = rows[:, 14].mean()
    ghz = rows[:, 4:14].sum() / max(life.sum(), 1)
    print("[%3d T%3d] t_wgrad %7.1f us (incl. slab reduce) | %5d wgs, %.1f tiles each, lifetime %.1f us, %.2f GHz | cycles per TILE: overhead %.0f, "
          "close-barrier %.0f, stager %.0f, open-barrier %.0f, k-steps %.0f | slab stores %.0f per workgroup"
          % (f, T, e0.elapsed_time(e1) * 1e3, len(rows), tiles, life.mean() / 1e3, ghz, ph[0] / tiles, ph[1] / tiles, ph[2] / tiles,
             ph[3] / tiles, ph[4] / tiles, ph[5]))
