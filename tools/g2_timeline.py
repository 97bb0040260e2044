"""Reads the per-workgroup timeline rows of a -DSAR_G2_TIMELINE build (tools/g2_timeline.sh) for the graph forward / data
gradient at the NTU layer shapes and prints: kernel span, workgroup lifetime, resident workgroups per CU, phase cycles."""
import sys, os, ctypes, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/skeleton-action-recognition_amd")
from sar_amd import ops, ops8, _lib as L
from graph.ntu_rgb_d import Graph
dev = torch.device("cuda:0")
lib = L.load()
B, V = 128, 25
A = Graph().A.astype(np.float32)
tab, tabT = ops.GraphTables(A, dev), ops.GraphTables(A, dev, True)
def rnd(C, n, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return ops8.from_cn(torch.randn((C, n), generator=g, device=dev))
def pack(W, st, sc, sm, taps, Kc, M):
    pk = ops.PackedWeights(); pk.add("w", 0, st, sc, sm, taps, Kc, M); pk.finalize(dev); pk.refresh(W.reshape(-1)); return pk.image("w")
NW = 16384
buf = np.zeros((NW, 12), dtype=np.uint32)
ptr = buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint))
lib.sar_debug_g2_timeline.argtypes = [ctypes.POINTER(ctypes.c_uint), ctypes.c_int, ctypes.c_int]
for (cin, f, T) in [(64, 64, 300), (128, 128, 150), (256, 256, 75)]:
    n = B * T * V
    X, dG = rnd(cin, n, 1), rnd(f, n, 4)
    g = torch.Generator(device=dev).manual_seed(9)
    Wg = torch.randn((cin, 3 * f), generator=g, device=dev) * 0.1
    pw_gb, pw_gf = pack(Wg, f, 1, 3 * f, 3, f, cin), pack(Wg, f, 3 * f, 1, 3, cin, f)
    g_, dx = ops8.empty(f, n, dev), ops8.empty(cin, n, dev)
    X2 = rnd(cin, n, 5)
    gmask = torch.randint(0, 256, ((cin + 7) // 8, X2.shape[1]), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
    gmean = 0.1 * torch.randn(cin, generator=g, device=dev)
    K = {"g_fwd": lambda: ops8.conv_gemm(L.SAR_CONV_GRAPH, X, g_, pw_gf, B=B, V=V, T_src=T, T_out=T, Kc=cin, M=f, taps=3, tables=tab, epi=L.SAR_EPI_STATS),
         "g_dgate": lambda: ops8.conv_gemm(L.SAR_CONV_GRAPH, dG, dx, pw_gb, B=B, V=V, T_src=T, T_out=T, Kc=f, M=cin, taps=3, tables=tabT, epi=L.SAR_EPI_ADD_GATE, aux=X, aux2=X2, aux_mask=gmask, aux_mean=gmean),
         "g_dgrad": lambda: ops8.conv_gemm(L.SAR_CONV_GRAPH, dG, dx, pw_gb, B=B, V=V, T_src=T, T_out=T, Kc=f, M=cin, taps=3, tables=tabT, epi=L.SAR_EPI_ADD, aux=X)}
    for name, fn in K.items():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        lib.sar_debug_g2_timeline(None, 0, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        lib.sar_debug_g2_timeline(ptr, NW, 0)
        rows = buf[buf[:, 1] != 0].astype(np.int64)
        st, en = rows[:, 0], rows[:, 1]
        t0 = st.min()
        st, en = (st - t0) * 10.0, (en - t0) * 10.0            # ns
        life = en - st
        hw, xcc = rows[:, 2], rows[:, 3] & 0xf
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)   # (xcc, se, sh, cu)
        ncu = len(np.unique(cu))
        # resident workgroups per CU, time-averaged over the kernel span
        span = en.max()
        resid = life.sum() / (ncu * span)
        # peak concurrency on the busiest CU
        peak = 0
        for c in np.unique(cu)[:16]:
            m = cu == c
            ev = sorted([(a, 1) for a in st[m]] + [(b_, -1) for b_ in en[m]])
            cur = 0
            for _, dlt in ev:
                cur += dlt; peak = max(peak, cur)
        per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
        ghz = rows[:, 4:9].sum() / max(life.sum(), 1)     # cycles per ns
        last_start = st.max()
        first_wave = np.sort(st)[min(len(st) - 1, 4 * ncu - 1)]
        print("[%3d->%3d T%3d] %-7s event %.1f us | %d workgroups on %d CUs (%d..%d per CU), span %.1f us, lifetime mean %.2f us "
              "(p10 %.2f p90 %.2f), resident per CU %.2f (peak %d), %.2f GHz | cycles: prologue %.0f + barrier %.0f + %.0f, stages %.0f, epilogue %.0f | "
              "first %d workgroups started within %.2f us, last start at %.1f us"
              % (cin, f, T, name, e0.elapsed_time(e1) * 1e3, len(rows), ncu, per_cu.min(), per_cu.max(), span / 1e3, life.mean() / 1e3,
                 np.percentile(life, 10) / 1e3, np.percentile(life, 90) / 1e3, resid, peak, ghz, rows[:, 4].mean(), rows[:, 5].mean(),
                 rows[:, 6].mean(), rows[:, 7].mean(), rows[:, 8].mean(), 4 * ncu, first_wave / 1e3, last_start / 1e3))
