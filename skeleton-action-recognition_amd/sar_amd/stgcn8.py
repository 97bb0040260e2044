"""The ST-GCN train step in the bf16 configuration (engine option mfma="bf16"): every activation
and activation gradient lives in HBM as a bfloat16 CN8 tensor (csrc/cn8.h), the convolutions multiply bf16 operands on
v_mfma_f32_32x32x16_bf16 with fp32 accumulation, BatchNorm statistics / backward sums are reduced in fp32 from the
accumulators, parameters, gradients, optimizer state and the loss head are fp32.  Same schedule as the fp32 engine
(sar_amd/stgcn.py): per block only the BN-barrier tensors g, u, (r,) y are materialised, BN + ReLU is folded into the
consumer's operand staging, the 3F-channel GraphConvTD intermediate never exists.

These functions are the bodies of STGCN.forward / STGCN.backward when engine.cn8 is set."""
import torch

from . import _lib as L
from . import ops, ops8
from .stgcn import BN_EPS, BN_MOMENTUM, KS, KT, same_pad  # noqa: F401

RELU_MASK = __import__("os").environ.get("SAR_CN8_RELU_MASK", "1") == "1"
# Round 4: the graph data gradient of block i gates its result (= the output gradient of block i - 1) with block i - 1's ReLU
# mask and reduces block i - 1's BatchNorm-backward sums in its epilogue (SAR_EPI_ADD_GATE): block i - 1 then needs neither the
# bn_add_relu_bwd_reduce pass nor the masked-gradient write of its apply pass.  Not for a block whose residual branch has its own
# BatchNorm (a third sum over r) and not for the last block (its output gradient comes from the pooling).  SAR_CN8_FUSE_TAIL=0: off.
FUSE_TAIL = __import__("os").environ.get("SAR_CN8_FUSE_TAIL", "1") == "1"


def forward(eng, x, training=True, keep=None):
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 5
    x = x.contiguous()
    N, Cin, T, V, M = x.shape
    assert Cin == eng.C_in and V == eng.V
    dev, B = x.device, N * M
    saved = {"x": x, "N": N, "M": M, "T": T, "blocks": [], "training": training}
    eng.packed.refresh(eng.flat)           # bf16 operand images of every conv weight, one launch
    nch = V * Cin
    if training:
        part = torch.empty((nch, N, 2), dtype=torch.float32, device=dev)
        ops.data_bn_stats(x, eng.bone_parent, part, eng.motion)
        eng._bn_forward_stats("data_bn", part, N, N * M * T, True, False)
    else:
        eng._bn_eval("data_bn")
    dbn = eng.bn["data_bn"]
    h = ops8.empty(Cin, B * T * V, dev)
    ops8.data_bn_apply(x, eng.bone_parent, dbn.scale, dbn.shift, h, eng.motion)
    if keep is not None:
        keep["x0"] = h
    Tc, cin = T, Cin
    for i, (f, s, res) in enumerate(eng.blocks):
        h, Tc = _block_forward(eng, i, h, cin, f, s, B, Tc, training, saved, keep)
        cin = f
    feat = torch.empty((N, cin), dtype=torch.float32, device=dev)
    ops8.pool_fwd(h, cin, B, Tc * V, M, feat)
    logits = torch.empty((N, eng.num_classes), dtype=torch.float32, device=dev)
    ops.fc_fwd(feat, eng.p["logits.kernel"].view(cin, eng.num_classes), eng.p["logits.bias"], logits)
    saved.update(feat=feat, T_last=Tc, y_last_shape=(cin, B * Tc * V))
    eng._saved = saved if training else None
    if keep is not None:
        keep["feat"] = feat
    return logits


def _block_forward(eng, i, X, cin, f, s, B, T, training, saved, keep):
    V, dev = eng.V, X.device
    pre = "l%d." % i
    kind = eng.kinds[i]
    To, pad, _ = same_pad(T, KT, s)
    n_in, n_out = B * T * V, B * To * V
    epi = L.SAR_EPI_STATS if training else L.SAR_EPI_NONE
    img = eng.packed.image
    g = ops8.empty(f, n_in, dev)
    r1 = ops8.conv_gemm(L.SAR_CONV_GRAPH, X, g, img(pre + "gcn.f"), B=B, V=V, T_src=T, T_out=T, Kc=cin, M=f, taps=KS,
                        bias=eng.p[pre + "gcn.bias"], tables=eng.tab_fwd, epi=epi)
    if training:
        eng._bn_forward_stats(pre + "bn1", r1[0], r1[1], n_in, True, True)
    else:
        eng._bn_eval(pre + "bn1")
    bn1 = eng.bn[pre + "bn1"]
    u = ops8.empty(f, n_out, dev)
    r2 = ops8.conv_gemm(L.SAR_CONV_TEMPORAL, g, u, img(pre + "tcn.f"), B=B, V=V, T_src=T, T_out=To, Kc=f, M=f, taps=KT, stride=s,
                        pad=pad, bias=eng.p[pre + "tcn.bias"], pro=(bn1.scale, bn1.shift), pro_relu=True, epi=epi)
    if training:
        eng._bn_forward_stats(pre + "bn2", r2[0], r2[1], n_out, True, True)
    else:
        eng._bn_eval(pre + "bn2")
    bn2 = eng.bn[pre + "bn2"]
    r = rbn = None
    if kind == "conv":
        r = ops8.empty(f, n_out, dev)
        r3 = ops8.conv_gemm(L.SAR_CONV_TEMPORAL, X, r, img(pre + "res.f"), B=B, V=V, T_src=T, T_out=To, Kc=cin, M=f, taps=1,
                            stride=s, pad=0, bias=eng.p[pre + "res.bias"], epi=epi)
        if training:
            eng._bn_forward_stats(pre + "res_bn", r3[0], r3[1], n_out, True, True)
        else:
            eng._bn_eval(pre + "res_bn")
        rbn = eng.bn[pre + "res_bn"]
    y = ops8.empty(f, n_out, dev)
    res_kind = {"none": 0, "identity": 1, "conv": 2}[kind]
    # training: the tail also writes its ReLU mask, one BYTE per 16-byte unit; the two BatchNorm-backward passes read that
    # instead of y (SAR_CN8_RELU_MASK=0: they read y)
    ymask = ops8.relu_mask(f, n_out, dev) if (training and RELU_MASK) else None
    ops8.bn_add_relu_fwd(u, bn2.scale, bn2.shift, res_kind, X if kind == "identity" else r, rbn.scale if rbn else None,
                         rbn.shift if rbn else None, y, f, mask=ymask)
    if training:
        saved["blocks"].append(dict(X=X, g=g, u=u, r=r, y=y, ymask=ymask, T=T, To=To, pad=pad, cin=cin, f=f, s=s, kind=kind))
    if keep is not None:
        keep[pre + "g"], keep[pre + "u"], keep[pre + "y"] = g, u, y
    return y, To


def backward(eng, dlogits, bucket_cb=None):
    sv = eng._saved
    assert sv is not None, "backward() needs a preceding forward(training=True)"
    dev, V = dlogits.device, eng.V
    N, M = sv["N"], sv["M"]
    B = N * M
    c_last = eng.C_last
    feat = sv["feat"]
    dfeat = torch.empty_like(feat)
    ops.fc_bwd(feat, eng.p["logits.kernel"].view(c_last, eng.num_classes), dlogits.contiguous(),
               eng.g["logits.kernel"].view(c_last, eng.num_classes), eng.g["logits.bias"], dfeat)
    dY = ops8.empty(c_last, sv["y_last_shape"][1], dev)
    ops8.pool_bwd(dfeat, c_last, B, sv["T_last"] * V, M, dY)
    gated = None       # (partials, nparts) of this block's BatchNorm-backward sums when dY arrives gated from the block above
    fuse = FUSE_TAIL and RELU_MASK and bool(eng.tab_bwd.g_flags & L.SAR_GRAPH_FEW_DENSE) and __import__("os").environ.get("SAR_GRAPH_READ_GATHER", "1") != "0"
    for i in reversed(range(len(eng.blocks))):
        below = sv["blocks"][i - 1] if (fuse and i >= 1 and eng.kinds[i - 1] != "conv" and sv["blocks"][i - 1].get("ymask") is not None) else None
        dY, gated = _block_backward(eng, i, sv["blocks"][i], dY, B, gated, below)
        if eng._deferred:
            eng._flush_deferred()
        eng._buckets_after_block(i, bucket_cb)
    # data_bn gamma / beta need the input gradient of block 0 (the input itself needs none)
    x = sv["x"]
    nch = V * eng.C_in
    part = torch.empty((nch, N, 2), dtype=torch.float32, device=dev)
    dbn = eng.bn["data_bn"]
    ops8.data_bn_bwd_reduce(x, eng.bone_parent, dY, dbn.mean, part, eng.motion)
    ops.bn_bwd_finalize(part, N, N * 2, 2, 0, 1, nch, N * M * sv["T"], eng.p["data_bn.gamma"], dbn.mean, dbn.rstd,
                        eng.g["data_bn.gamma"], eng.g["data_bn.beta"])
    eng._finish_backward(bucket_cb)


def _block_backward(eng, i, sb, dY, B, gated=None, below=None):
    """backward of block i.  gated: this block's BatchNorm-backward partial sums when dY already carries the ReLU gate (the
    block above produced both in its graph data gradient); below: the saved tensors of block i - 1 when THIS block's graph data
    gradient is to do the same for it.  Returns (dX, gated-for-the-block-below)."""
    V, dev = eng.V, dY.device
    pre = "l%d." % i
    X, g, u, r, y = sb["X"], sb["g"], sb["u"], sb["r"], sb["y"]
    T, To, pad, cin, f, s, kind = sb["T"], sb["To"], sb["pad"], sb["cin"], sb["f"], sb["s"], sb["kind"]
    n_in, n_out = B * T * V, B * To * V
    bn1, bn2 = eng.bn[pre + "bn1"], eng.bn[pre + "bn2"]
    rbn = eng.bn.get(pre + "res_bn")
    conv = kind == "conv"
    img = eng.packed.image
    # ---- tail: y = relu(bn2(u) + res)
    rk = (rbn.k1, rbn.k2, rbn.k3) if conv else None
    if gated is not None:    # the sums came with dY: (sum dz, sum dz (u - mean)) per channel and partial
        assert not conv
        ops.bn_bwd_finalize(gated[0], gated[1], gated[1] * 2, 2, 0, 1, f, n_out, eng.p[pre + "bn2.gamma"], bn2.mean, bn2.rstd,
                            eng.g[pre + "bn2.gamma"], eng.g[pre + "bn2.beta"], bn2.k1, bn2.k2, bn2.k3)
    elif ops.BN_TAIL:          # the reduce kernel's last workgroup per plane finalises BN2 (and the residual BN): no launch between
        tail = ops.make_bn_tail(dev, n_out, eng.p[pre + "bn2.gamma"], bn2, eng.g[pre + "bn2.gamma"], eng.g[pre + "bn2.beta"],
                                *((eng.p[pre + "res_bn.gamma"], rbn, eng.g[pre + "res_bn.gamma"], eng.g[pre + "res_bn.beta"])
                                  if conv else ()))
        ops8.bn_add_relu_bwd_reduce(dY, y, u, r if conv else None, f, bn2.mean, rbn.mean if conv else None, tail=tail)
    else:
        part, nparts = ops8.bn_add_relu_bwd_reduce(dY, y, u, r if conv else None, f, bn2.mean, rbn.mean if conv else None,
                                                   mask=sb.get("ymask"))
        ops.bn_bwd_finalize(part, nparts, nparts * 4, 4, 0, 1, f, n_out, eng.p[pre + "bn2.gamma"], bn2.mean, bn2.rstd,
                            eng.g[pre + "bn2.gamma"], eng.g[pre + "bn2.beta"], bn2.k1, bn2.k2, bn2.k3)
        if conv:
            ops.bn_bwd_finalize(part, nparts, nparts * 4, 4, 0, 2, f, n_out, eng.p[pre + "res_bn.gamma"], rbn.mean, rbn.rstd,
                                eng.g[pre + "res_bn.gamma"], eng.g[pre + "res_bn.beta"], rbn.k1, rbn.k2, rbn.k3)
    du = ops8.empty(f, n_out, dev)
    dr = ops8.empty(f, n_out, dev) if conv else None
    dz = dY if (kind == "identity" and gated is None) else None      # in place: dY becomes the pre-ReLU gradient for the skip path (already gated: nothing to write)
    ops8.bn_add_relu_bwd_apply(dY, y, u, r if conv else None, (bn2.k1, bn2.k2, bn2.k3), rk, du, dr, dz, f, mask=sb.get("ymask"))
    # ---- temporal conv: weight / bias gradient, then data gradient fused with the ReLU mask and the BN1 reductions
    flat_w = eng.grad[eng.offsets[pre + "tcn.kernel"]:eng.offsets[pre + "tcn.bias"] + f]
    eng._off_critical_path(lambda: ops8.conv_wgrad(
        L.SAR_CONV_TEMPORAL, g, du, flat_w, B=B, V=V, T_src=T, T_out=To, Kc=f, M=f, taps=KT, stride=s, pad=pad,
        pro=(bn1.scale, bn1.shift), pro_relu=True, w_stride_tap=f * f, w_stride_c=f, wsize=KT * f * f, bsize=f, slabs=eng._slabs), g, du)
    dz1 = ops8.empty(f, n_in, dev)
    pm = ops8.conv_gemm(L.SAR_CONV_TEMPORAL, du, dz1, img(pre + "tcn.b"), B=B, V=V, T_src=To, T_out=T, Kc=f, M=f, taps=KT,
                        stride=s, pad=pad, transposed=True, epi=L.SAR_EPI_MASK, aux=g, aux_affine=(bn1.scale, bn1.shift),
                        aux_mean=bn1.mean)
    ops.bn_bwd_finalize(pm[0], pm[1], pm[1] * 2, 2, 0, 1, f, n_in, eng.p[pre + "bn1.gamma"], bn1.mean, bn1.rstd,
                        eng.g[pre + "bn1.gamma"], eng.g[pre + "bn1.beta"], bn1.k1, bn1.k2, bn1.k3)
    dg = dz1
    ops8.affine2(dz1, g, (bn1.k1, bn1.k2, bn1.k3), dg, f)             # BN1 backward apply (in place)
    # ---- graph conv: weight / bias gradient
    flat_g = eng.grad[eng.offsets[pre + "gcn.kernel"]:eng.offsets[pre + "gcn.bias"] + KS * f]
    eng._off_critical_path(lambda: ops8.conv_wgrad(
        L.SAR_CONV_GRAPH, X, dg, flat_g, B=B, V=V, T_src=T, T_out=T, Kc=cin, M=f, taps=KS, tables=eng.tab_fwd,
        w_stride_tap=f, w_stride_c=KS * f, wsize=cin * KS * f, bsize=KS * f, slabs=eng._slabs), X, dg)
    # ---- residual conv branch
    dXres = None
    if conv:
        flat_r = eng.grad[eng.offsets[pre + "res.kernel"]:eng.offsets[pre + "res.bias"] + f]
        eng._off_critical_path(lambda: ops8.conv_wgrad(
            L.SAR_CONV_TEMPORAL, X, dr, flat_r, B=B, V=V, T_src=T, T_out=To, Kc=cin, M=f, taps=1, stride=s, pad=0,
            w_stride_tap=0, w_stride_c=f, wsize=cin * f, bsize=f, slabs=eng._slabs), X, dr)
        dXres = ops8.empty(cin, n_in, dev)
        ops8.conv_gemm(L.SAR_CONV_TEMPORAL, dr, dXres, img(pre + "res.b"), B=B, V=V, T_src=To, T_out=T, Kc=f, M=cin, taps=1,
                       stride=s, pad=0, transposed=True)
    # ---- graph conv data gradient (+ skip-path gradient)
    dX = ops8.empty(cin, n_in, dev)
    aux = dY if kind == "identity" else dXres
    if below is not None and aux is not None:
        # dX = gate_{i-1}(W^T dg . A^T + skip gradient) and block i - 1's BatchNorm-backward sums in one epilogue
        bn2b = eng.bn["l%d.bn2" % (i - 1)]
        pm = ops8.conv_gemm(L.SAR_CONV_GRAPH, dg, dX, img(pre + "gcn.b"), B=B, V=V, T_src=T, T_out=T, Kc=f, M=cin, taps=KS,
                            tables=eng.tab_bwd, epi=L.SAR_EPI_ADD_GATE, aux=aux, aux2=below["u"], aux_mask=below["ymask"],
                            aux_mean=bn2b.mean)
        return dX, pm
    ops8.conv_gemm(L.SAR_CONV_GRAPH, dg, dX, img(pre + "gcn.b"), B=B, V=V, T_src=T, T_out=T, Kc=f, M=cin, taps=KS,
                   tables=eng.tab_bwd, epi=L.SAR_EPI_ADD if aux is not None else L.SAR_EPI_NONE, aux=aux)
    return dX, None
