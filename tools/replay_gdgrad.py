"""Debug: capture the graph data-gradient launches of one fp32 train step and replay them in the split arithmetics."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skeleton-action-recognition_amd"))
from oracle import stgcn as O
from sar_amd import ops, _lib as L
from sar_amd.stgcn import STGCN
dev = torch.device("cuda:0")
blocks = list(O.BLOCKS)
p = O.randomize_affine(O.init_params(60, seed=3, dtype=torch.float64, blocks=blocks), seed=4)
x, y = O.synthetic_batch(2, seed=3, T=300, num_classes=60)
eng = STGCN(num_classes=60, device=dev, blocks=blocks, mfma="fp32")
eng.load_params(p)
calls = []
orig = ops.conv_gemm
def rec(mode, src, out, W, *a, **kw):
    r = orig(mode, src, out, W, *a, **kw)
    if mode == L.SAR_CONV_GRAPH and kw.get("tables") is eng.tab_bwd:
        torch.cuda.synchronize()
        calls.append((src.clone(), out.clone(), W.clone(), a, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()}))
    return r
ops.conv_gemm = rec
import sar_amd.stgcn as S
os.environ["SAR_F32_FUSE_TAIL"] = "0"
S._FUSE_TAIL_F32 = False
eng.loss_and_grad(x.to(dev), y.to(dev))
ops.conv_gemm = orig
for (src, out_ref, W, a, kw) in [calls[1], calls[0], calls[0], calls[1]]:
    kw = dict(kw)
    aux = kw.get("aux")
    print("call Kc=%d M=%d: src amax %.3e rms %.3e  |src| quantiles (of amax) %s  aux amax %s" % (
        kw["Kc"], kw["M"], src.abs().max().item(), src.pow(2).mean().sqrt().item(),
        ["%.1e" % (torch.quantile(src.abs().flatten()[::97].float(), q).item() / src.abs().max().item()) for q in (0.1, 0.5, 0.9, 0.99)],
        None if aux is None else "%.3e" % aux.abs().max().item()))
    # fp64 reference through the fp32 kernel's formula is not available here: compare to the fp32 kernel's output
    for m in ("bf16x6", "f16x3a"):
        out = torch.empty_like(out_ref)
        kw2 = dict(kw); kw2["split"] = m; kw2["packed"] = None; kw2.pop("bf16", None)
        orig(L.SAR_CONV_GRAPH, src, out, W, *a, **kw2)
        torch.cuda.synchronize()
        e = (out - out_ref).abs().max().item() / out_ref.abs().max().item()
        e2 = ((out - out_ref).norm() / out_ref.norm()).item()
        # without the aux term
        if aux is not None:
            d1, d0 = out - aux, out_ref - aux
            e3 = ((d1 - d0).norm() / d0.norm()).item()
        else:
            e3 = float("nan")
        print("   %-7s vs fp32 kernel: max %.3e  l2 %.3e  l2 of the GEMM part alone %.3e" % (m, e, e2, e3))
        if e > 1e-4:
            bad = ((out - out_ref).abs() > 1e-4 * out_ref.abs().max()).nonzero()
            rows, cols = bad[:, 0], bad[:, 1]
            print("      bad elements %d: rows %s..%s cols min %d max %d; cols mod 25: %s; frames: %s" % (
                bad.shape[0], rows.min().item(), rows.max().item(), cols.min().item(), cols.max().item(),
                sorted(set((cols % 25).tolist()))[:30], sorted(set(((cols // 25) % 75).tolist()))[:40]))
            print("      src stats at joint columns: amax per joint", [("%.1e" % src[:, j::25].abs().max().item()) for j in range(25)])
