"""The fp32 launches that stay on the fp32 kernels in every fp32-storage engine ("leftovers": the 3-channel first layer, the
strided 1x1 residual convolutions of blocks 5 and 8), ALONE at bs = 64: time, bytes moved by the algorithm, GB/s.  These shapes
are memory-bound (<= 8 GFLOP per launch), so the yardstick is the HBM time of their algorithmic bytes.

  python tools/leftover_bench.py [--reps 7]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from kernel_bench import timeit  # noqa: E402
from sar_amd import _lib as L, ops  # noqa: E402
from sar_amd.stgcn import ntu_adjacency  # noqa: E402


def main():
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 7
    dev = torch.device("cuda:0")
    B, V = 128, 25
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *sh: torch.randn(sh, device=dev, generator=g)
    A = ntu_adjacency().astype("float32")
    tf_ = ops.GraphTables(A, dev)
    rows = []

    def case(name, nbytes, fn):
        ms = timeit(fn, reps)
        rows.append((name, ms, nbytes))
        print("%-44s %8.1f us  %7.1f MB  %6.2f TB/s  (HBM floor at 6 TB/s: %6.1f us)" % (
            name, ms * 1e3, nbytes / 1e6, nbytes / (ms * 1e-3) / 1e12, nbytes / 6e12 * 1e6), flush=True)

    # ---- first layer: 3 -> 64 channels, T = 300
    T, cin, f = 300, 3, 64
    n = B * T * V
    X, G = rn(cin, n), rn(f, n)
    Wg, bg = rn(cin, 3 * f) * 0.1, rn(3 * f) * 0.1
    out = torch.empty((f, n), device=dev)
    flat = torch.empty(cin * 3 * f + 3 * f, device=dev)
    case("l0 graph forward 3 -> 64 (STATS)", 4.0 * (cin + f) * n, lambda: ops.conv_gemm(
        L.SAR_CONV_GRAPH, X, out, Wg, f, 3 * f, B=B, V=V, T_src=T, T_out=T, Kc=cin, M=f, taps=3, bias=bg, tables=tf_,
        epi=L.SAR_EPI_STATS, split=None))
    case("l0 graph weight gradient 3 -> 64", 4.0 * (cin + f) * n, lambda: ops.conv_wgrad(
        L.SAR_CONV_GRAPH, X, G, flat, B=B, V=V, T_src=T, T_out=T, Kc=cin, M=f, taps=3, tables=tf_, w_stride_tap=f,
        w_stride_c=3 * f, wsize=cin * 3 * f, bsize=3 * f, split=None))
    # ---- strided 1x1 residual convolutions
    for cin, f, T in ((64, 128, 300), (128, 256, 150)):
        To = T // 2
        n_in, n_out = B * T * V, B * To * V
        X, dr = rn(cin, n_in), rn(f, n_out) * 1e-3
        W, bias = rn(cin, f) * 0.1, rn(f) * 0.1
        rT = W.t().contiguous()
        r = torch.empty((f, n_out), device=dev)
        dXres = torch.empty((cin, n_in), device=dev)
        dXc = torch.empty((cin, n_out), device=dev)
        flat = torch.empty(cin * f + f, device=dev)
        tag = "%d -> %d T %d" % (cin, f, T)
        # forward: reads the even frames of X (every other 100-byte frame: the bus moves whole sectors), writes r
        case("res forward " + tag + " (STATS)", 4.0 * (cin * n_in + f * n_out), lambda: ops.conv_gemm(
            L.SAR_CONV_TEMPORAL, X, r, W, 0, f, B=B, V=V, T_src=T, T_out=To, Kc=cin, M=f, taps=1, stride=2, pad=0, bias=bias,
            epi=L.SAR_EPI_STATS, split=None))
        case("res data gradient " + tag + " (strided)", 4.0 * (f * n_out + cin * n_in), lambda: ops.conv_gemm(
            L.SAR_CONV_TEMPORAL, dr, dXres, rT, 0, cin, B=B, V=V, T_src=To, T_out=T, Kc=f, M=cin, taps=1, stride=2, pad=0,
            transposed=True, split=None))
        case("res data gradient " + tag + " (compact)", 4.0 * (f * n_out + cin * n_out), lambda: ops.conv_gemm(
            L.SAR_CONV_TEMPORAL, dr, dXc, rT, 0, cin, B=B, V=V, T_src=To, T_out=To, Kc=f, M=cin, taps=1, stride=1, pad=0,
            split=None))
        # the same launches on conv_tap1_split_kernel (f16x3a; term images and bounds prepared outside the timed launches)
        imf = ops._pack_split_single(W, 0, f, 1, cin, f, "f16x3a")
        imb = ops._pack_split_single(rT, 0, cin, 1, f, cin, "f16x3a")
        bX, bD = ops._src_bound_single(X, None), ops._src_bound_single(dr, None)
        case("res forward " + tag + " (STATS) f16x3a", 4.0 * (cin * n_in + f * n_out), lambda: ops.conv_gemm(
            L.SAR_CONV_TEMPORAL, X, r, W, 0, f, B=B, V=V, T_src=T, T_out=To, Kc=cin, M=f, taps=1, stride=2, pad=0, bias=bias,
            epi=L.SAR_EPI_STATS, split="f16x3a", packed=imf[0], bounds=(bX, imf[1])))
        case("res data gradient " + tag + " (compact) f16x3a", 4.0 * (f * n_out + cin * n_out), lambda: ops.conv_gemm(
            L.SAR_CONV_TEMPORAL, dr, dXc, rT, 0, cin, B=B, V=V, T_src=To, T_out=To, Kc=f, M=cin, taps=1, stride=1, pad=0,
            split="f16x3a", packed=imb[0], bounds=(bD, imb[1])))
        case("res weight gradient " + tag + " f16x3a", 4.0 * (cin * n_in + f * n_out), lambda: ops.conv_wgrad(
            L.SAR_CONV_TEMPORAL, X, dr, flat, B=B, V=V, T_src=T, T_out=To, Kc=cin, M=f, taps=1, stride=2, pad=0, w_stride_tap=0,
            w_stride_c=f, wsize=cin * f, bsize=f, split="f16x3a", bounds=(bX, bD)))
        case("res weight gradient " + tag, 4.0 * (cin * n_in + f * n_out), lambda: ops.conv_wgrad(
            L.SAR_CONV_TEMPORAL, X, dr, flat, B=B, V=V, T_src=T, T_out=To, Kc=cin, M=f, taps=1, stride=2, pad=0, w_stride_tap=0,
            w_stride_c=f, wsize=cin * f, bsize=f, split=None))
    tot = sum(ms for _, ms, _ in rows)
    floor = sum(nb for _, _, nb in rows) / 6e12 * 1e3
    print("TOTAL %.3f ms alone (both data-gradient forms counted); HBM floor of the same bytes %.3f ms" % (tot, floor))


if __name__ == "__main__":
    main()
