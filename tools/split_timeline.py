"""Reads the per-workgroup timeline rows of a -DSAR_SPLIT_TL build (tools/split_timeline.sh) for the split-arithmetic kernels at
the NTU layer shapes and prints: launch time, workgroup lifetime, resident workgroups per CU, clock, mean cycles per phase
(0 prologue / tables, 7 geometry + first requests, 1 store phase, 2 wait for the W DMA, 3 opening barrier, 4 load issue + MFMA
phase, 5 closing barrier + DMA issue, 6 epilogue)."""
import sys, os, ctypes, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/skeleton-action-recognition_amd")
from sar_amd import ops, _lib as L
from sar_amd.stgcn import same_pad
from graph.ntu_rgb_d import Graph
dev = torch.device("cuda:0")
lib = L.load()
AR = os.environ.get("SPLIT_ARITH", "f16x3a")
B, V = 128, 25
A = Graph().A.astype(np.float32)
tab, tabT = ops.GraphTables(A, dev), ops.GraphTables(A, dev, True)
NW = 16384
buf = np.zeros((NW, 16), dtype=np.uint32)
ptr = buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint))
lib.sar_debug_split_timeline.argtypes = [ctypes.POINTER(ctypes.c_uint), ctypes.c_int, ctypes.c_int]
shapes = [(64, 64, 300, 1), (128, 128, 150, 1), (256, 256, 75, 1)]
only = set(sys.argv[1:])
for (cin, f, T, s) in shapes:
    n = B * T * V
    g = torch.Generator(device=dev).manual_seed(9)
    rn = lambda *sh: torch.randn(sh, device=dev, generator=g)
    X, G, dG = torch.relu(rn(cin, n)), rn(f, n), rn(f, n) * 1e-5
    Wg, bg = rn(cin, 3 * f) * 0.1, rn(3 * f) * 0.1
    Wt, bt = rn(9, f, f) * 0.05, rn(f) * 0.1
    sc, sh = 1 + 0.1 * rn(f), 0.1 * rn(f)
    gT, wT = Wg.t().contiguous(), Wt.transpose(1, 2).contiguous()
    out_f, dX = torch.empty((f, n), device=dev), torch.empty((cin, n), device=dev)
    mask = torch.randint(0, 256, (cin, n // 4), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
    mean = 0.1 * rn(cin)
    img = {k: ops._pack_split_single(*a, AR) for k, a in {"gf": (Wg, f, 3 * f, 3, cin, f), "gb": (gT, f * cin, cin, 3, f, cin),
                                                            "tf": (Wt, f * f, f, 9, f, f), "tb": (wT, f * f, f, 9, f, f)}.items()}
    bX, bG, bdG = ops._src_bound_single(X, None), ops._src_bound_single(G, (sc, sh)), ops._src_bound_single(dG, None)
    K = {"g_fwd": lambda: ops.conv_gemm(L.SAR_CONV_GRAPH, X, out_f, Wg, f, 3 * f, B=B, V=V, T_src=T, T_out=T, Kc=cin, M=f, taps=3, bias=bg,
                                        tables=tab, epi=L.SAR_EPI_STATS, split=AR, packed=img["gf"][0], bounds=(bX, img["gf"][1])),
         "g_dgate": lambda: ops.conv_gemm(L.SAR_CONV_GRAPH, dG, dX, gT, f * cin, cin, B=B, V=V, T_src=T, T_out=T, Kc=f, M=cin, taps=3,
                                          tables=tabT, epi=L.SAR_EPI_ADD_GATE, aux=X, aux2=X, aux_mask=mask, aux_mean=mean, split=AR,
                                          packed=img["gb"][0], bounds=(bdG, img["gb"][1])),
         "t_fwd": lambda: ops.conv_gemm(L.SAR_CONV_TEMPORAL, G, out_f, Wt, f * f, f, B=B, V=V, T_src=T, T_out=T, Kc=f, M=f, taps=9, stride=1,
                                        pad=4, bias=bt, pro=(sc, sh), pro_relu=True, epi=L.SAR_EPI_STATS, split=AR, packed=img["tf"][0],
                                        bounds=(bG, img["tf"][1])),
         "t_dgrad": lambda: ops.conv_gemm(L.SAR_CONV_TEMPORAL, dG, out_f, wT, f * f, f, B=B, V=V, T_src=T, T_out=T, Kc=f, M=f, taps=9,
                                          stride=1, pad=4, transposed=True, epi=L.SAR_EPI_MASK, aux=G, aux_affine=(sc, sh), split=AR,
                                          packed=img["tb"][0], bounds=(bdG, img["tb"][1]))}
    for name, fn in K.items():
        if only and name not in only:
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.sar_debug_split_timeline(None, 0, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        lib.sar_debug_split_timeline(ptr, NW, 0)
        rows = buf[buf[:, 1] != 0].astype(np.int64)
        st, en = rows[:, 0], rows[:, 1]
        t0 = st.min()
        st, en = (st - t0) * 10.0, (en - t0) * 10.0            # ns
        life = en - st
        hw, xcc = rows[:, 2], rows[:, 3] & 0xf
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
        ncu = len(np.unique(cu))
        span = en.max()
        resid = life.sum() / (ncu * span)
        ph = rows[:, 4:14].mean(axis=0)
        if os.environ.get("SAR_GRAPH_SPLIT3", "0") == "1" and name.startswith("g_"):
            # producer / consumer kernel (conv_graph_split3_kernel): two rows per workgroup (row[14] = role)
            tiles = B * (-(-T // 10))
            ny = max(1, (cin if name == "g_dgate" else f) // 64)
            for role, label in ((0, "PRODUCERS"), (1, "CONSUMERS")):
                rr = rows[rows[:, 14] == role]
                per_wg = tiles * ny / len(rr)
                q = rr[:, 4:14].mean(axis=0) / per_wg
                lf = (rr[:, 1] - rr[:, 0]) * 10.0
                ghz_r = rr[:, 4:14].sum() / max(lf.sum(), 1)
                if role == 0:
                    print("[%3d->%3d T%3d] %-7s %7.1f us | %s %d wgs, %.2f tiles each, %.2f GHz | per TILE: barrier wait %.0f, DMA issue %.0f, convert %.0f, "
                          "DMA wait %.0f, epilogue-slot wait %.0f (sum %.0f) | prologue per wg %.0f"
                          % (cin, f, T, name, e0.elapsed_time(e1) * 1e3, label, len(rr), per_wg, ghz_r, q[3], q[8], q[1], q[2], q[6], q[1:].sum(), rr[:, 4].mean()))
                else:
                    print("[%3d->%3d T%3d] %-7s %7.1f us | %s %d wgs, %.2f tiles each, %.2f GHz | per TILE: set-up %.0f, barrier wait %.0f, k-steps %.0f, "
                          "epilogue-slot barrier %.0f, epilogue %.0f (sum %.0f)"
                          % (cin, f, T, name, e0.elapsed_time(e1) * 1e3, label, len(rr), per_wg, ghz_r, q[7], q[3], q[4], q[5], q[6], q[1:].sum()))
            continue
        if os.environ.get("SAR_GRAPH_SPLIT2", "1") != "0" and name.startswith("g_"):
            # the persistent kernel (conv_graph_split2_kernel): a row = one workgroup = several tiles; phases per TILE
            ntile_total = {"g_fwd": None}.get(name)
            tiles = B * (-(-T // 10))                      # column tiles of the launch
            ny = max(1, (cin if name == "g_dgate" else f) // 64)
            per_wg = tiles * ny / len(rows)
            q = ph / per_wg
            print("[%3d->%3d T%3d] %-7s %7.1f us | PERSISTENT %d wgs on %d CUs, %.2f tiles each, lifetime %.1f us, %.2f GHz | cycles per workgroup: "
                  "prologue %.0f | per TILE: set-up %.0f, convert %.0f, dma-wait %.0f, barrier-B %.0f, raw-issue %.0f, mfma %.0f, barrier-C+W-issue %.0f, "
                  "epilogue %.0f, barrier-E %.0f (sum %.0f)"
                  % (cin, f, T, name, e0.elapsed_time(e1) * 1e3, len(rows), ncu, per_wg, life.mean() / 1e3, rows[:, 4:14].sum() / max(life.sum(), 1),
                     ph[0], q[7], q[1], q[2], q[3], q[8], q[4], q[5], q[6], q[9], q[1:].sum()))
            continue
        ghz = rows[:, 4:14].sum() / max(life.sum(), 1)
        print("[%3d->%3d T%3d] %-7s %7.1f us | %5d wgs on %d CUs, lifetime %.1f us (p10 %.1f p90 %.1f), resident/CU %.2f, %.2f GHz | cycles: "
              "tables %.0f, geometry+init %.0f, store %.0f, dma-wait %.0f, open-barrier %.0f, load-issue %.0f, mfma %.0f, close-barrier %.0f, epilogue %.0f (sum %.0f)"
              % (cin, f, T, name, e0.elapsed_time(e1) * 1e3, len(rows), ncu, life.mean() / 1e3, np.percentile(life, 10) / 1e3,
                 np.percentile(life, 90) / 1e3, resid, ghz, ph[0], ph[7], ph[1], ph[2], ph[3], ph[8], ph[4], ph[5], ph[6], ph.sum()))
