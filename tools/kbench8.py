"""Per-kernel timing of the CN8 conv kernels at the NTU layer shapes (bs = 64): python tools/kbench8.py"""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/skeleton-action-recognition_amd")
from sar_amd import ops, ops8, _lib as L
from graph.ntu_rgb_d import Graph
dev = torch.device("cuda:0")
B, V = 128, 25
A = Graph().A.astype(np.float32)
tab, tabT = ops.GraphTables(A, dev), ops.GraphTables(A, dev, True)
def rnd(C, n, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return ops8.from_cn(torch.randn((C, n), generator=g, device=dev))
def pack(W, st, sc, sm, taps, Kc, M):
    pk = ops.PackedWeights(); pk.add("w", 0, st, sc, sm, taps, Kc, M); pk.finalize(dev); pk.refresh(W.reshape(-1)); return pk.image("w")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
NOEPI = os.environ.get("KB_NOEPI", "0") == "1"      # forward / data-gradient kernels with the plain store epilogue
tot = {}
for (cin, f, s, T, cnt) in [(64, 64, 1, 300, 3), (64, 128, 2, 300, 1), (128, 128, 1, 150, 2), (128, 256, 2, 150, 1), (256, 256, 1, 75, 2)]:
    To = -(-T // s); pad = max((To - 1) * s + 9 - T, 0) // 2
    n_in, n_out = B * T * V, B * To * V
    X, G, dU, dG = rnd(cin, n_in, 1), rnd(f, n_in, 2), rnd(f, n_out, 3), rnd(f, n_in, 4)
    g = torch.Generator(device=dev).manual_seed(9)
    Wt = torch.randn((9, f, f), generator=g, device=dev) * 0.05
    Wg = torch.randn((cin, 3 * f), generator=g, device=dev) * 0.1
    sc, sh, mean = 1 + 0.2 * torch.randn(f, generator=g, device=dev), 0.3 * torch.randn(f, generator=g, device=dev), 0.1 * torch.randn(f, generator=g, device=dev)
    pw_tb, pw_gb = pack(Wt, f * f, 1, f, 9, f, f), pack(Wg, f, 1, 3 * f, 3, f, cin)
    pw_tf, pw_gf = pack(Wt, f * f, f, 1, 9, f, f), pack(Wg, f, 3 * f, 1, 3, cin, f)
    u, g_, dz, dx = ops8.empty(f, n_out, dev), ops8.empty(f, n_in, dev), ops8.empty(f, n_in, dev), ops8.empty(cin, n_in, dev)
    flat_t, flat_g = torch.zeros(9 * f * f + f, device=dev), torch.zeros(cin * 3 * f + 3 * f, device=dev)
    X2 = rnd(cin, n_in, 5)
    gmask = torch.randint(0, 256, ((cin + 7) // 8, X2.shape[1]), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
    gmean = 0.1 * torch.randn(cin, generator=g, device=dev)
    K = {
        "t_fwd": lambda: ops8.conv_gemm(L.SAR_CONV_TEMPORAL, G, u, pw_tf, B=B, V=V, T_src=T, T_out=To, Kc=f, M=f, taps=9, stride=s, pad=pad, pro=(sc, sh), pro_relu=True, epi=L.SAR_EPI_NONE if NOEPI else L.SAR_EPI_STATS),
        "t_dgrad": lambda: ops8.conv_gemm(L.SAR_CONV_TEMPORAL, dU, dz, pw_tb, B=B, V=V, T_src=To, T_out=T, Kc=f, M=f, taps=9, stride=s, pad=pad, transposed=True, epi=L.SAR_EPI_NONE if NOEPI else L.SAR_EPI_MASK, aux=None if NOEPI else G, aux_affine=None if NOEPI else (sc, sh), aux_mean=None if NOEPI else mean),
        "g_fwd": lambda: ops8.conv_gemm(L.SAR_CONV_GRAPH, X, g_, pw_gf, B=B, V=V, T_src=T, T_out=T, Kc=cin, M=f, taps=3, tables=tab, epi=L.SAR_EPI_STATS),
        "g_dgrad": lambda: ops8.conv_gemm(L.SAR_CONV_GRAPH, dG, dx, pw_gb, B=B, V=V, T_src=T, T_out=T, Kc=f, M=cin, taps=3, tables=tabT, epi=L.SAR_EPI_ADD, aux=X),
        "g_dgate": lambda: ops8.conv_gemm(L.SAR_CONV_GRAPH, dG, dx, pw_gb, B=B, V=V, T_src=T, T_out=T, Kc=f, M=cin, taps=3, tables=tabT, epi=L.SAR_EPI_ADD_GATE, aux=X, aux2=X2, aux_mask=gmask, aux_mean=gmean),
        "t_wgrad": lambda: ops8.conv_wgrad(L.SAR_CONV_TEMPORAL, G, dU, flat_t, B=B, V=V, T_src=T, T_out=To, Kc=f, M=f, taps=9, stride=s, pad=pad, pro=(sc, sh), pro_relu=True, w_stride_tap=f * f, w_stride_c=f, wsize=9 * f * f, bsize=f),
        "g_wgrad": lambda: ops8.conv_wgrad(L.SAR_CONV_GRAPH, X, dG, flat_g, B=B, V=V, T_src=T, T_out=T, Kc=cin, M=f, taps=3, tables=tab, w_stride_tap=f, w_stride_c=3 * f, wsize=cin * 3 * f, bsize=3 * f),
    }
    line = "[%3d->%3d s%d T%3d x%d]" % (cin, f, s, T, cnt)
    stamps = []
    for name, fn in K.items():
        if only and name not in only: continue
        us = timeit(fn)
        tot[name] = tot.get(name, 0.0) + us * cnt
        line += "  %s %6.1f us" % (name, us)
        lib = L.load()
        if hasattr(lib, "sar_debug_wgrad8_stamps") and name in ("t_wgrad", "g_wgrad"):
            import ctypes
            buf = (ctypes.c_ulonglong * 10)()
            torch.cuda.synchronize(); lib.sar_debug_wgrad8_stamps(buf, 1)
            fn(); torch.cuda.synchronize(); lib.sar_debug_wgrad8_stamps(buf, 1)
            v = [float(x) for x in buf]
            nwg, ghz, nt = v[6], v[8] / max(v[7], 1) * 0.1, max(v[9], 1)
            stamps.append("    %-8s %5d workgroups x %.1f tiles, lifetime %6.1f us at %.2f GHz | per tile (cycles): store %5.0f  barrier %5.0f  "
                          "loads+MFMA %5.0f  barrier %5.0f | prologue %5.0f  epilogue %5.0f" %
                          (name, nwg, nt / nwg, v[7] / nwg / 100, ghz, v[0] / nt, v[1] / nt, v[2] / nt, v[3] / nt, v[4] / nwg, v[5] / nwg))
        if hasattr(lib, "sar_debug_cn8_stamps") and name in ("t_fwd", "t_dgrad", "g_fwd", "g_dgrad"):   # diagnostic build (tools/stamps8.sh)
            import ctypes
            buf = (ctypes.c_ulonglong * 10)()
            torch.cuda.synchronize(); lib.sar_debug_cn8_stamps(buf, 1)
            fn(); torch.cuda.synchronize(); lib.sar_debug_cn8_stamps(buf, 1)
            v = [float(x) for x in buf]
            nwg, ghz = v[6], v[8] / max(v[7], 1) * 0.1
            if nwg == 0: continue   # the launch went to a kernel of another translation unit (LDS-DMA data gradient, read-gather graph kernel)
            kc = f if name in ("t_fwd", "t_dgrad", "g_dgrad") else cin
            nst = -(-kc // 16)
            if name[0] == "t":
                stamps.append("    %-8s %5d workgroups, lifetime %6.1f us at %.2f GHz | per stage (cycles): store %5.0f  barrier %5.0f  "
                              "loads+MFMA %5.0f  barrier %5.0f | prologue %5.0f  epilogue %5.0f" %
                              (name, nwg, v[7] / nwg / 100, ghz, v[0] / nwg / nst, v[1] / nwg / nst, v[2] / nwg / nst, v[3] / nwg / nst,
                               v[4] / nwg, v[5] / nwg))
            else:
                stamps.append("    %-8s %5d workgroups, lifetime %6.1f us at %.2f GHz | per stage (cycles): store raw %5.0f  barrier %5.0f  "
                              "gather %5.0f  barrier %5.0f  loads+MFMA %5.0f | prologue %5.0f  epilogue %5.0f" %
                              (name, nwg, v[7] / nwg / 100, ghz, v[0] / nwg / nst, v[1] / nwg / nst, v[2] / nwg / nst, v[3] / nwg / nst,
                               v[9] / nwg / nst, v[4] / nwg, v[5] / nwg))
    print(line)
    for st in stamps: print(st)
print("per step (9 of 10 blocks):", {k: round(v / 1e3, 2) for k, v in tot.items()}, "sum %.2f ms" % (sum(tot.values()) / 1e3))
