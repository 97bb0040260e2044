#!/usr/bin/env python3
"""Per-shape timing of the conv-GEMM / wgrad kernels at the bench shapes (bs=64 => B=128 sequences).
Usage: python tools/kernel_bench.py [--reps 5] [--only tconv_fwd,gcn_fwd,...] [--layers 2,6,9]
Prints one line per (kernel, layer): ms, algorithmic TFLOP/s, fraction of the 157.3 TF fp32 MFMA peak."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from sar_amd import _lib as L, ops  # noqa: E402
from sar_amd.stgcn import BLOCKS, same_pad, ntu_adjacency  # noqa: E402

PEAK = 157.3
NOPRO = os.environ.get("KB_NOPRO", "0") == "1"      # diagnostic: temporal forward / weight gradient without the folded BN + ReLU prologue


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--only", default="")
    ap.add_argument("--layers", default="")
    ap.add_argument("--bf16", action="store_true", help="temporal fwd/dgrad with bf16 MFMA operands (sar_conv_gemm_bf16)")
    ap.add_argument("--split", default=None, help="split arithmetic of csrc/conv_*_split.hip (f16x3a / bf16x6): term images and operand "
                                                  "bounds are prepared once, outside the timed launches")
    a = ap.parse_args()
    only = set(a.only.split(",")) if a.only else None
    layers = set(int(x) for x in a.layers.split(",")) if a.layers else None
    dev = torch.device("cuda:0")
    B, V = a.batch * 2, 25
    A = ntu_adjacency().astype("float32")
    tf_, tb_ = ops.GraphTables(A, dev), ops.GraphTables(A, dev, True)
    T, cin = 300, 3
    tot = {}
    for i, (f, s, res) in enumerate(BLOCKS):
        To, pad, _ = same_pad(T, 9, s)
        n_in, n_out = B * T * V, B * To * V
        if layers is None or i in layers:
            g = torch.Generator(device=dev).manual_seed(i)
            rn = lambda *sh: torch.randn(sh, device=dev, generator=g)
            X, G, U = rn(cin, n_in), rn(f, n_in), rn(f, n_out)
            Wg, bg = rn(cin, 3 * f) * 0.1, rn(3 * f) * 0.1
            Wt, bt = rn(9, f, f) * 0.05, rn(f) * 0.1
            sc, sh = 1 + 0.1 * rn(f), 0.1 * rn(f)
            out_in, out_out = torch.empty((f, n_in), device=dev), torch.empty((f, n_out), device=dev)
            dX = torch.empty((cin, n_in), device=dev)
            gT = Wg.t().contiguous()
            wT = Wt.transpose(1, 2).contiguous()
            flat_g = torch.empty(cin * 3 * f + 3 * f, device=dev)
            flat_t = torch.empty(9 * f * f + f, device=dev)
            U = U * 1e-5      # gradient-like magnitudes
            gmask = (torch.randint(0, 256, (max(cin, 8), n_in // 4), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
                     if n_in % 4 == 0 else None)
            gmean = 0.1 * rn(cin)
            sp = a.split
            kw = {k_: dict(split=None) for k_ in ("gf", "gb", "tf", "tb", "tw", "gw")}
            if sp:
                bX, bG, bGp, bU = (ops._src_bound_single(X, None), ops._src_bound_single(G, None), ops._src_bound_single(G, (sc, sh)),
                                   ops._src_bound_single(U, None))
                if ops.split_applicable(L.SAR_CONV_GRAPH, V, cin, f, 3, 1, tf_):
                    im = ops._pack_split_single(Wg, f, 3 * f, 3, cin, f, sp)
                    kw["gf"] = dict(split=sp, packed=im[0], bounds=(bX, im[1]))
                    kw["gw"] = dict(split=sp, bounds=(bX, bG))
                if ops.split_applicable(L.SAR_CONV_GRAPH, V, f, cin, 3, 1, tb_):
                    im = ops._pack_split_single(gT, f * cin, cin, 3, f, cin, sp)
                    kw["gb"] = dict(split=sp, packed=im[0], bounds=(bG, im[1]))
                im = ops._pack_split_single(Wt, f * f, f, 9, f, f, sp)
                kw["tf"] = dict(split=sp, packed=im[0], bounds=(bGp, im[1]))
                im = ops._pack_split_single(wT, f * f, f, 9, f, f, sp)
                kw["tb"] = dict(split=sp, packed=im[0], bounds=(bU, im[1]))
                kw["tw"] = dict(split=sp, bounds=(bGp, bU))
            cases = {
                "gcn_fwd": (2.0 * f * cin * 3 * n_in, lambda: ops.conv_gemm(
                    L.SAR_CONV_GRAPH, X, out_in, Wg, f, 3 * f, B=B, V=V, T_src=T, T_out=T, Kc=cin, M=f, taps=3, bias=bg,
                    tables=tf_, epi=L.SAR_EPI_STATS, bf16=a.bf16, **kw["gf"])),
                "tconv_fwd": (2.0 * f * f * 9 * n_out, lambda: ops.conv_gemm(
                    L.SAR_CONV_TEMPORAL, G, out_out, Wt, f * f, f, B=B, V=V, T_src=T, T_out=To, Kc=f, M=f, taps=9, stride=s,
                    pad=pad, bias=bt, pro=None if NOPRO else (sc, sh), pro_relu=not NOPRO, epi=L.SAR_EPI_STATS, bf16=a.bf16, **kw["tf"])),
                "tconv_dgrad": (2.0 * f * f * 9 * n_out, lambda: ops.conv_gemm(
                    L.SAR_CONV_TEMPORAL, U, out_in, wT, f * f, f, B=B, V=V, T_src=To, T_out=T, Kc=f, M=f, taps=9, stride=s,
                    pad=pad, transposed=True, epi=L.SAR_EPI_MASK, aux=G, aux_affine=(sc, sh), bf16=a.bf16, **kw["tb"])),
                "gcn_dgrad": (2.0 * f * cin * 3 * n_in, lambda: ops.conv_gemm(
                    L.SAR_CONV_GRAPH, G, dX, gT, f * cin, cin, B=B, V=V, T_src=T, T_out=T, Kc=f, M=cin, taps=3, tables=tb_,
                    epi=L.SAR_EPI_ADD, aux=X, bf16=a.bf16, **kw["gb"])),
                # the gated data gradient (SAR_EPI_ADD_GATE): what 7 of the 9 graph data gradients of a train step are
                "gcn_dgate": (2.0 * f * cin * 3 * n_in, lambda: ops.conv_gemm(
                    L.SAR_CONV_GRAPH, G, dX, gT, f * cin, cin, B=B, V=V, T_src=T, T_out=T, Kc=f, M=cin, taps=3, tables=tb_,
                    epi=L.SAR_EPI_ADD_GATE, aux=X, aux2=X, aux_mask=gmask, aux_mean=gmean, bf16=a.bf16, **kw["gb"])),
                "tconv_wgrad": (2.0 * f * f * 9 * n_out, lambda: ops.conv_wgrad(
                    L.SAR_CONV_TEMPORAL, G, U, flat_t, B=B, V=V, T_src=T, T_out=To, Kc=f, M=f, taps=9, stride=s, pad=pad,
                    pro=None if NOPRO else (sc, sh), pro_relu=not NOPRO, w_stride_tap=f * f, w_stride_c=f, wsize=9 * f * f, bsize=f, bf16=a.bf16,
                    **kw["tw"])),
                "gcn_wgrad": (2.0 * f * cin * 3 * n_in, lambda: ops.conv_wgrad(
                    L.SAR_CONV_GRAPH, X, G, flat_g, B=B, V=V, T_src=T, T_out=T, Kc=cin, M=f, taps=3, tables=tf_,
                    w_stride_tap=f, w_stride_c=3 * f, wsize=cin * 3 * f, bsize=3 * f, bf16=a.bf16, **kw["gw"])),
            }
            for name, (flops, fn) in cases.items():
                if only and name not in only:
                    continue
                if name == "gcn_dgate" and (not only or cin % 8 or gmask is None):      # on request only; M % 8 == 0
                    continue
                ms = timeit(fn, a.reps)
                tf = flops / (ms * 1e-3) / 1e12
                tot.setdefault(name, [0.0, 0.0])
                tot[name][0] += ms; tot[name][1] += flops
                print("L%-2d %-12s cin=%3d f=%3d T=%3d->%3d s=%d  %8.3f ms  %7.2f TF  %5.1f%%" %
                      (i + 1, name, cin, f, T, To, s, ms, tf, 100 * tf / PEAK), flush=True)
        T, cin = To, f
    for name, (ms, fl) in tot.items():
        print("TOTAL %-12s %8.3f ms  %7.2f TF" % (name, ms, fl / (ms * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
