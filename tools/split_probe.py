#!/usr/bin/env python3
"""Split-arithmetic probe: the 9-tap temporal forward convolution 256 -> 256 (and the other layer
shapes with --all) computed by the native fp32 MFMA kernel and by csrc/conv_gemm_split.hip with x1 / x3 / x6 / x9 bf16 terms and
the fp16 x3 probe: error against a float64 evaluation of the same fp32 inputs, and microseconds per launch at bs = 64.
Usage: python tools/split_probe.py [--reps 20] [--all] [--out gpurun_out/split_probe.txt]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from sar_amd import _lib as L, ops  # noqa: E402
from sar_amd.stgcn import same_pad  # noqa: E402

MODES = [None, "bf16x1", "bf16x3", "bf16x6", "bf16x9", "f16x3", "f16x3s", "f16x3a"]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3


def ref_fwd(G, Wt, bt, sc, sh, B, T, To, V, s, pad):
    """float64: out[m,(b,to,v)] = sum_tap sum_c W[tap,c,m] relu(bn(g))[c,(b, to*s + tap - pad, v)] + bias"""
    f = G.shape[0]
    a = torch.relu(G.double() * sc.double()[:, None] + sh.double()[:, None]).view(f, B, T, V)
    ap = torch.zeros((f, B, T + 16, V), dtype=torch.float64, device=G.device)
    ap[:, :, 8:8 + T] = a
    out = bt.double()[:, None].repeat(1, B * To * V)
    for tap in range(9):
        lo = 8 + tap - pad
        x = ap[:, :, lo:lo + (To - 1) * s + 1:s].reshape(f, B * To * V)
        out += Wt[tap].double().t() @ x
    return out


def ref_dgrad(U, Wt, B, T, To, V, s, pad):
    """float64 data gradient: dsrc[c,(b,t,v)] = sum_tap sum_m W[tap,c,m] dout[m,(b,(t + pad - tap)/s,v)] when divisible"""
    f = U.shape[0]
    u = U.double().view(f, B, To, V)
    out = torch.zeros((Wt.shape[1], B, T + 16, V), dtype=torch.float64, device=U.device)
    for tap in range(9):
        y = (Wt[tap].double() @ u.reshape(f, -1)).view(-1, B, To, V)      # [c][b][to][v]
        lo = 8 + tap - pad
        out[:, :, lo:lo + (To - 1) * s + 1:s] += y
    return out[:, :, 8:8 + T].reshape(-1, B * T * V)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--all", action="store_true", help="every 9-tap layer shape, forward and data gradient")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    V = 25
    lines = []

    def emit(sx):
        print(sx, flush=True)
        lines.append(sx)

    shapes = [(256, 1, 75)]
    if a.all:
        shapes = [(64, 1, 300), (128, 2, 300), (128, 1, 150), (256, 2, 150), (256, 1, 75)]
    emit("# split-arithmetic probe: error vs float64 of the same fp32 inputs (B = 8 sequences), us per launch at B = 128 (bs = 64)")
    emit("# %-9s %-5s %-8s %12s %12s %10s %10s %8s" % ("shape", "kind", "arith", "rel_l2", "max/absmax", "us", "TF(fp32eq)", "x native"))
    for (f, s, T) in shapes:
        To, pad, _ = same_pad(T, 9, s)
        for kind in (("fwd", "dgrad") if a.all else ("fwd",)):
            g = torch.Generator(device=dev).manual_seed(f + s)
            rn = lambda *sh: torch.randn(sh, device=dev, generator=g)
            Wt, bt = rn(9, f, f) * (2.0 / (9 * f)) ** 0.5, rn(f) * 0.1
            sc, sh = 1 + 0.1 * rn(f), 0.1 * rn(f)
            wT = Wt.transpose(1, 2).contiguous()
            res = {}
            for B in (8, 128):
                n_in, n_out = B * T * V, B * To * V
                G, U = rn(f, n_in), rn(f, n_out) * 1e-4
                if kind == "fwd":
                    out = torch.empty((f, n_out), device=dev)
                    ref = ref_fwd(G, Wt, bt, sc, sh, B, T, To, V, s, pad) if B == 8 else None
                    call = lambda m: ops.conv_gemm(L.SAR_CONV_TEMPORAL, G, out, Wt, f * f, f, B=B, V=V, T_src=T, T_out=To, Kc=f, M=f,
                                                   taps=9, stride=s, pad=pad, bias=bt, pro=(sc, sh), pro_relu=True,
                                                   epi=L.SAR_EPI_STATS, split=m)
                else:
                    out = torch.empty((f, n_in), device=dev)
                    ref = ref_dgrad(U, Wt, B, T, To, V, s, pad) if B == 8 else None
                    call = lambda m: ops.conv_gemm(L.SAR_CONV_TEMPORAL, U, out, wT, f * f, f, B=B, V=V, T_src=To, T_out=T, Kc=f, M=f,
                                                   taps=9, stride=s, pad=pad, transposed=True, epi=L.SAR_EPI_MASK, aux=G,
                                                   aux_affine=(sc * 0 + 1, sh * 0 + 1e30), split=m)   # mask always open
                for m in MODES:
                    if B == 8:
                        out.fill_(float("nan"))
                        call(m)
                        torch.cuda.synchronize()
                        e = out.double() - ref
                        res[m] = [float(e.norm() / ref.norm()), float(e.abs().max() / ref.abs().max())]
                    else:
                        if m is not None:      # pack and bound once, outside the timed launches (the engines do so once per step)
                            pk, wb = ops._pack_split_single(Wt if kind == "fwd" else wT, f * f, f, 9, f, f, m)
                            sb = ops._src_bound_single(G, (sc, sh)) if kind == "fwd" else ops._src_bound_single(U, None)
                            d_call = (lambda m=m, pk=pk: ops.conv_gemm(
                                L.SAR_CONV_TEMPORAL, G, out, Wt, f * f, f, B=B, V=V, T_src=T, T_out=To, Kc=f, M=f, taps=9, stride=s,
                                pad=pad, bias=bt, pro=(sc, sh), pro_relu=True, epi=L.SAR_EPI_STATS, split=m, packed=pk, bounds=(sb, wb))) if kind == "fwd" else (
                                lambda m=m, pk=pk: ops.conv_gemm(
                                    L.SAR_CONV_TEMPORAL, U, out, wT, f * f, f, B=B, V=V, T_src=To, T_out=T, Kc=f, M=f, taps=9, stride=s,
                                    pad=pad, transposed=True, epi=L.SAR_EPI_MASK, aux=G, aux_affine=(sc, sh), split=m, packed=pk, bounds=(sb, wb)))
                        else:
                            d_call = lambda: call(None)
                        res[m].append(timeit(d_call, a.reps))
            flops = 2.0 * f * f * 9 * 128 * To * V
            for m in MODES:
                r = res[m]
                emit("  %-9s %-5s %-8s %12.3e %12.3e %10.1f %10.1f %8.2f" % (
                    "%d/s%d/T%d" % (f, s, T), kind, m or "fp32", r[0], r[1], r[2], flops / r[2] * 1e-6, res[None][2] / r[2]))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
