"""bf16 configuration, kernel level: the CN8 kernels (include/sar_hip.h "CN8" section) against
their exact definitions -- inputs ARE bfloat16 (stored activations), weights are rounded to bfloat16 when packed, products
exact, contraction in float64 here / fp32 on the GPU, outputs rounded to bfloat16 once.  Tolerances: BatchNorm partial
sums (taken from the fp32 accumulators) 1e-4; stored outputs within one bfloat16 rounding (2^-8 relative per element plus
1e-5 of the tensor scale for the accumulation order)."""
import numpy as np
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import stgcn as O
from util import to_cn, from_cn, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _bf(t):
    return t.float().bfloat16().double()


def _A():
    from oracle.graph import spatial_adjacency
    return torch.tensor(spatial_adjacency().astype(np.float32))


def _tables(dev, transpose=False):
    from sar_amd import ops
    from oracle.graph import spatial_adjacency
    return ops.GraphTables(spatial_adjacency().astype(np.float32), dev, transpose)


def _pack(dev, W, st, sc, sm, taps, Kc, M):
    from sar_amd import ops
    pk = ops.PackedWeights()
    pk.add("w", 0, st, sc, sm, taps, Kc, M)
    pk.finalize(dev)
    pk.refresh(W.to(dev).contiguous().reshape(-1))
    return pk.image("w")


def _cn8(x, dev):
    """(B,C,T,V) float tensor whose values are bf16-representable -> CN8 on the device"""
    from sar_amd import ops8
    return ops8.from_cn(to_cn(x.float()).to(dev))


def _back(x8, C, B, T):
    from sar_amd import ops8
    return from_cn(ops8.to_cn(x8, C).cpu(), B, T, 25)


def assert_bf16_close(got, ref, what=""):
    got, ref = got.double(), ref.double()
    scale = ref.abs().max().item()
    bad = (got - ref).abs() - (2.0 ** -8) * ref.abs() - 1e-5 * scale
    assert bad.max().item() <= 0, "%s: worst excess %.3e (scale %.3e)" % (what, bad.max().item(), scale)


def test_layout_roundtrip_and_padding(dev):
    from sar_amd import ops8
    g = torch.Generator().manual_seed(0)
    for C in (3, 8, 20, 64):
        x = torch.randn(C, 1000, generator=g).bfloat16().float().to(dev)
        x8 = ops8.from_cn(x)
        assert x8.shape == ((C + 7) // 8, 1000, 8)
        assert torch.equal(ops8.to_cn(x8, C), x)
        # the definition of the layout: unit (g, col)[j] = channel 8g + j, zero beyond C
        ref = torch.zeros(((C + 7) // 8) * 8, 1000, device=dev)
        ref[:C] = x
        assert torch.equal(x8.float(), ref.view(-1, 8, 1000).permute(0, 2, 1))


@pytest.mark.parametrize("B,f,T,s", [(2, 64, 13, 1), (3, 64, 14, 2), (2, 128, 9, 2), (1, 256, 6, 1), (2, 128, 300, 2),
                                     (2, 72, 11, 1), (1, 200, 9, 2), (2, 40, 7, 1), (1, 48, 10, 2), (2, 24, 30, 1)])
def test_temporal_conv_forward(dev, B, f, T, s):
    from sar_amd import ops8, _lib as L
    g = torch.Generator().manual_seed(f + T + s)
    x = torch.randn(B, f, T, 25, generator=g).bfloat16()
    sc = 1 + 0.2 * torch.randn(f, generator=g); sh = 0.3 * torch.randn(f, generator=g)
    kernel = torch.randn(9, 1, f, f, generator=g) * 0.05
    bias = torch.randn(f, generator=g) * 0.1
    h = torch.relu(torch.addcmul(sh.view(1, -1, 1, 1), x.float(), sc.view(1, -1, 1, 1)))      # fp32 fma like the kernel
    ref = O.temporal_conv(_bf(h), _bf(kernel), bias.double(), s)
    To, pad, _ = O.same_pad(T, 9, s)
    out = ops8.empty(f, B * To * 25, dev)
    r = ops8.conv_gemm(L.SAR_CONV_TEMPORAL, _cn8(x, dev), out, _pack(dev, kernel, f * f, f, 1, 9, f, f), B=B, V=25, T_src=T,
                       T_out=To, Kc=f, M=f, taps=9, stride=s, pad=pad, bias=bias.to(dev), pro=(sc.to(dev), sh.to(dev)),
                       pro_relu=True, epi=L.SAR_EPI_STATS)
    torch.cuda.synchronize()
    assert_bf16_close(_back(out, f, B, To), ref, "temporal conv")
    part = r[0].cpu().double().sum(dim=1)
    assert rel_err(part[:, 0], ref.sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(part[:, 1], (ref * ref).sum(dim=(0, 2, 3))) < 1e-4
    if f % 8:
        assert (out.float()[-1, :, f % 8:] == 0).all()       # channels beyond C stay zero


@pytest.mark.parametrize("B,cin,f,T,s", [(2, 64, 128, 13, 2), (2, 128, 256, 10, 2), (1, 64, 64, 7, 1), (2, 24, 40, 9, 2)])
def test_residual_conv_forward_and_data_gradient(dev, B, cin, f, T, s):
    from sar_amd import ops8, _lib as L
    g = torch.Generator().manual_seed(cin + f)
    x = torch.randn(B, cin, T, 25, generator=g).bfloat16()
    kernel = torch.randn(1, 1, cin, f, generator=g) * 0.1
    bias = torch.randn(f, generator=g) * 0.1
    xd = x.double().requires_grad_(True)
    ref = F.conv2d(xd, O.hwio_to_oihw(_bf(kernel)), bias.double(), stride=(s, 1))
    To = ref.shape[2]
    out = ops8.empty(f, B * To * 25, dev)
    r = ops8.conv_gemm(L.SAR_CONV_TEMPORAL, _cn8(x, dev), out, _pack(dev, kernel, 0, f, 1, 1, cin, f), B=B, V=25, T_src=T, T_out=To,
                       Kc=cin, M=f, taps=1, stride=s, pad=0, bias=bias.to(dev), epi=L.SAR_EPI_STATS)
    torch.cuda.synchronize()
    assert_bf16_close(_back(out, f, B, To), ref.detach(), "residual conv")
    assert rel_err(r[0].cpu().double().sum(dim=1)[:, 1], (ref.detach() ** 2).sum(dim=(0, 2, 3))) < 1e-4
    dr = torch.randn(B, f, To, 25, generator=g).bfloat16()
    gx, = torch.autograd.grad(ref, xd, dr.double())
    dx = ops8.empty(cin, B * T * 25, dev)
    ops8.conv_gemm(L.SAR_CONV_TEMPORAL, _cn8(dr, dev), dx, _pack(dev, kernel, 0, 1, f, 1, f, cin), B=B, V=25, T_src=To, T_out=T,
                   Kc=f, M=cin, taps=1, stride=s, pad=0, transposed=True)
    torch.cuda.synchronize()
    assert_bf16_close(_back(dx, cin, B, T), gx, "residual conv data gradient")


@pytest.mark.parametrize("B,f,T,s", [(2, 64, 13, 1), (2, 64, 14, 2), (2, 128, 9, 2), (1, 256, 7, 1), (2, 64, 11, 2), (4, 256, 75, 1),
                                     (3, 128, 150, 2), (2, 72, 11, 1), (1, 200, 9, 2), (2, 40, 12, 2), (2, 64, 12, 3)])
def test_temporal_conv_data_gradient(dev, B, f, T, s):
    """transposed conv of the stored (bf16) output gradient with the bf16 weights, fused ReLU mask (decided on the STORED
    pre-BatchNorm activation through the folded affine) and the centred BatchNorm-backward reductions."""
    from sar_amd import ops8, _lib as L
    g = torch.Generator().manual_seed(11 * f + T + s)
    gx = torch.randn(B, f, T, 25, generator=g).bfloat16()
    sc = 1 + 0.2 * torch.randn(f, generator=g); sh = 0.3 * torch.randn(f, generator=g); mean = 0.1 * torch.randn(f, generator=g)
    kernel = torch.randn(9, 1, f, f, generator=g) * 0.05
    To, pad, _ = O.same_pad(T, 9, s)
    du = torch.randn(B, f, To, 25, generator=g).bfloat16()
    h = torch.zeros(B, f, T, 25, dtype=torch.float64, requires_grad=True)
    dh, = torch.autograd.grad(O.temporal_conv(h, _bf(kernel), None, s), h, du.double())
    pre = torch.addcmul(sh.view(1, -1, 1, 1), gx.float(), sc.view(1, -1, 1, 1))
    g_pre = dh * (pre > 0)
    dz1 = ops8.empty(f, B * T * 25, dev)
    pm = ops8.conv_gemm(L.SAR_CONV_TEMPORAL, _cn8(du, dev), dz1, _pack(dev, kernel, f * f, 1, f, 9, f, f), B=B, V=25, T_src=To,
                        T_out=T, Kc=f, M=f, taps=9, stride=s, pad=pad, transposed=True, epi=L.SAR_EPI_MASK, aux=_cn8(gx, dev),
                        aux_affine=(sc.to(dev), sh.to(dev)), aux_mean=mean.to(dev))
    torch.cuda.synchronize()
    assert_bf16_close(_back(dz1, f, B, T), g_pre, "temporal data gradient")
    part = pm[0].cpu().double().sum(dim=1)
    assert rel_err(part[:, 0], g_pre.sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(part[:, 1], (g_pre * (gx.double() - mean.double().view(1, -1, 1, 1))).sum(dim=(0, 2, 3))) < 1e-4


def _graph_ref(x, kernel, bias, A, dev_tables):
    """exact definition: z_k = bf16(fp32 gather of the bf16 src in table order), W rounded to bf16, float64 contraction"""
    idx, wt = dev_tables.idx.cpu(), dev_tables.wt.cpu()                   # [3][V][4]
    Bq, cin, T, V = x.shape
    f = kernel.shape[3] // 3
    xs = x.float()
    out = torch.zeros(Bq, f, T, V, dtype=torch.float64)
    Wk = _bf(kernel)[0, 0]                                                 # (cin, 3f)
    for k in range(3):
        z = torch.zeros(Bq, cin, T, V)
        for w in range(V):
            acc = None
            for j in range(dev_tables.nz[k]):
                term = wt[k, w, j] * xs[:, :, :, idx[k, w, j]]
                acc = term if acc is None else torch.addcmul(acc, xs[:, :, :, idx[k, w, j]], wt[k, w, j])   # fp32 fma chain
            z[:, :, :, w] = acc
        zb = _bf(z)
        out += torch.einsum("bctv,cm->bmtv", zb, Wk[:, k * f:(k + 1) * f])
        if bias is not None:
            out += bias.double()[k * f:(k + 1) * f].view(1, -1, 1, 1) * A[k].double().sum(dim=0).view(1, 1, 1, -1)
    return out


@pytest.mark.parametrize("B,cin,f,T", [(3, 3, 64, 13), (2, 64, 64, 10), (2, 64, 128, 7), (1, 128, 256, 5), (4, 256, 256, 3),
                                       (2, 40, 72, 9), (1, 12, 200, 5), (2, 64, 44, 6), (2, 64, 64, 300)])
def test_graph_conv_forward(dev, B, cin, f, T):
    from sar_amd import ops8, _lib as L
    g = torch.Generator().manual_seed(B * 1000 + cin)
    x = torch.randn(B, cin, T, 25, generator=g).bfloat16()
    kernel = torch.randn(1, 1, cin, 3 * f, generator=g) * 0.1
    bias = torch.randn(3 * f, generator=g) * 0.1
    tab = _tables(dev)
    ref = _graph_ref(x, kernel, bias, _A(), tab)
    out = ops8.empty(f, B * T * 25, dev)
    r = ops8.conv_gemm(L.SAR_CONV_GRAPH, _cn8(x, dev), out, _pack(dev, kernel, f, 3 * f, 1, 3, cin, f), B=B, V=25, T_src=T, T_out=T,
                       Kc=cin, M=f, taps=3, bias=bias.to(dev), tables=tab, epi=L.SAR_EPI_STATS)
    torch.cuda.synchronize()
    assert_bf16_close(_back(out, f, B, T), ref, "graph conv")
    part = r[0].cpu().double().sum(dim=1)
    assert rel_err(part[:, 1], (ref * ref).sum(dim=(0, 2, 3))) < 1e-4
    # and it is the reference operator up to bf16 rounding of the operands
    full = O.graph_conv_td(x.double(), kernel.double(), bias.double(), _A().double())
    assert rel_err(_back(out, f, B, T), full) < 2e-2


@pytest.mark.parametrize("B,cin,f,T", [(2, 64, 64, 11), (2, 64, 128, 6), (1, 128, 256, 5), (3, 3, 64, 9), (4, 256, 256, 75),
                                       (2, 40, 72, 9)])
def test_graph_conv_data_gradient(dev, B, cin, f, T):
    """A^T gather lists + the (k, c', m') view of the kernel; SAR_EPI_ADD accumulates the skip-path gradient (CN8 aux)"""
    from sar_amd import ops8, _lib as L
    g = torch.Generator().manual_seed(7 * cin + f)
    kernel = torch.randn(1, 1, cin, 3 * f, generator=g) * 0.1
    dout = torch.randn(B, f, T, 25, generator=g).bfloat16()
    add = torch.randn(B, cin, T, 25, generator=g).bfloat16()
    tabT = _tables(dev, True)
    # data gradient = graph conv of dout with A^T tables and kernel^T: reuse the forward definition on the transposed problem
    kT = kernel[0, 0].view(cin, 3, f).permute(2, 1, 0).reshape(1, 1, f, 3 * cin)      # [m][k*cin + c]
    ref = _graph_ref(dout, kT, None, _A().transpose(1, 2), tabT) + add.double()
    dx = ops8.empty(cin, B * T * 25, dev)
    ops8.conv_gemm(L.SAR_CONV_GRAPH, _cn8(dout, dev), dx, _pack(dev, kernel, f, 1, 3 * f, 3, f, cin), B=B, V=25, T_src=T, T_out=T,
                   Kc=f, M=cin, taps=3, tables=tabT, epi=L.SAR_EPI_ADD, aux=_cn8(add, dev))
    torch.cuda.synchronize()
    assert_bf16_close(_back(dx, cin, B, T), ref, "graph data gradient")
    x = torch.zeros(B, cin, T, 25, dtype=torch.float64, requires_grad=True)
    gx, = torch.autograd.grad(O.graph_conv_td(x, kernel.double(), None, _A().double()), x, dout.double())
    assert rel_err(_back(dx, cin, B, T) - add.double(), gx) < 2e-2


@pytest.mark.parametrize("C,n,kind", [(64, 5000, 1), (128, 3001, 2), (20, 777, 0), (256, 9375, 2)])
def test_block_tail_relu_mask_is_bit_identical(dev, C, n, kind):
    """sar_bn_add_relu_fwd_mask_cn8 / _bwd_reduce_mask_cn8 / _bwd_apply_mask_cn8: the forward tail's one-byte-per-unit ReLU mask
    (bit j = stored channel 8 g + j > 0) replaces the reads of y in both backward passes -- every output bit for bit equal to
    the y-reading kernels, and the mask equal to its definition."""
    from sar_amd import ops8
    g = torch.Generator().manual_seed(3 * C + n)
    rnd = lambda *s: torch.randn(*s, generator=g)
    u, r, dy = (ops8.from_cn(rnd(C, n).bfloat16().float().to(dev)) for _ in range(3))
    sc, sh, rsc, rsh = (t.to(dev) for t in (1 + 0.2 * rnd(C), 0.3 * rnd(C), 1 + 0.2 * rnd(C), 0.3 * rnd(C)))
    mu, mr = (0.1 * rnd(C)).to(dev), (0.1 * rnd(C)).to(dev)
    k = [rnd(C).to(dev) for _ in range(6)]
    y0, y1 = ops8.empty(C, n, dev), ops8.empty(C, n, dev)
    mask = ops8.relu_mask(C, n, dev)
    res = r if kind else None
    ops8.bn_add_relu_fwd(u, sc, sh, kind, res, rsc if kind == 2 else None, rsh if kind == 2 else None, y0, C)
    ops8.bn_add_relu_fwd(u, sc, sh, kind, res, rsc if kind == 2 else None, rsh if kind == 2 else None, y1, C, mask=mask)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    bits = (y0.float() > 0).to(torch.int32)                                   # (G, n, 8)
    want = (bits * (2 ** torch.arange(8, device=dev, dtype=torch.int32))).sum(dim=2).to(torch.uint8)
    assert torch.equal(mask, want)
    rr = r if kind == 2 else None
    p0, np0 = ops8.bn_add_relu_bwd_reduce(dy, y0, u, rr, C, mu, mr if rr is not None else None)
    p1, np1 = ops8.bn_add_relu_bwd_reduce(dy, None, u, rr, C, mu, mr if rr is not None else None, mask=mask)
    torch.cuda.synchronize()
    assert np0 == np1 and torch.equal(p0[:, :, :3], p1[:, :, :3])
    outs = []
    for m in (None, mask):
        du, dr, dz = ops8.empty(C, n, dev), (ops8.empty(C, n, dev) if rr is not None else None), ops8.empty(C, n, dev)
        ops8.bn_add_relu_bwd_apply(dy, y0 if m is None else None, u, rr, k[:3], k[3:] if rr is not None else None, du, dr, dz, C, mask=m)
        outs.append((du, dr, dz))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert (a is None and b is None) or torch.equal(a, b)


@pytest.mark.parametrize("C,n,kind", [(64, 5000, 1), (128, 3001, 2), (20, 777, 0), (256, 9375, 2)])
def test_block_tail_forward_backward(dev, C, n, kind):
    """y = relu(bn2(u) + res) and its backward passes (sar_bn_add_relu_*_cn8, sar_affine2_cn8) against float64 arithmetic on
    the same bf16 inputs."""
    from sar_amd import ops8
    g = torch.Generator().manual_seed(C + n)
    rnd = lambda *s: torch.randn(*s, generator=g)
    u, r, dy = rnd(C, n).bfloat16(), rnd(C, n).bfloat16(), rnd(C, n).bfloat16()
    sc, sh, rsc, rsh = 1 + 0.2 * rnd(C), 0.3 * rnd(C), 1 + 0.2 * rnd(C), 0.3 * rnd(C)
    k = [0.5 * rnd(C) for _ in range(3)]
    rk = [0.5 * rnd(C) for _ in range(3)]
    mu, mr = 0.1 * rnd(C), 0.1 * rnd(C)
    D = lambda t: t.double()
    col = lambda t: D(t).view(-1, 1)
    z = D(u) * col(sc) + col(sh)
    if kind == 1:
        z = z + D(r)
    elif kind == 2:
        z = z + D(r) * col(rsc) + col(rsh)
    yref = torch.relu(z)
    u8, r8, dy8 = ops8.from_cn(u.float().to(dev)), ops8.from_cn(r.float().to(dev)), ops8.from_cn(dy.float().to(dev))
    y8 = ops8.empty(C, n, dev)
    ops8.bn_add_relu_fwd(u8, sc.to(dev), sh.to(dev), kind, r8 if kind else None, rsc.to(dev) if kind == 2 else None,
                         rsh.to(dev) if kind == 2 else None, y8, C)
    torch.cuda.synchronize()
    assert_bf16_close(ops8.to_cn(y8, C).cpu(), yref, "block tail forward")
    ystored = ops8.to_cn(y8, C).cpu().double()
    dz = D(dy) * (ystored > 0)
    conv = kind == 2
    part, nparts = ops8.bn_add_relu_bwd_reduce(dy8, y8, u8, r8 if conv else None, C, mu.to(dev), mr.to(dev) if conv else None)
    du8, dr8, dz8 = ops8.empty(C, n, dev), (ops8.empty(C, n, dev) if conv else None), ops8.empty(C, n, dev)
    ops8.bn_add_relu_bwd_apply(dy8, y8, u8, r8 if conv else None, [t.to(dev) for t in k], [t.to(dev) for t in rk] if conv else None,
                               du8, dr8, dz8, C)
    a8 = ops8.empty(C, n, dev)
    ops8.affine2(dy8, u8, [t.to(dev) for t in k], a8, C)
    torch.cuda.synchronize()
    p = part.cpu().double().sum(dim=1)
    assert rel_err(p[:, 0], dz.sum(dim=1)) < 1e-4
    assert rel_err(p[:, 1], (dz * (D(u) - col(mu))).sum(dim=1)) < 1e-4
    if conv:
        assert rel_err(p[:, 2], (dz * (D(r) - col(mr))).sum(dim=1)) < 1e-4
        assert_bf16_close(ops8.to_cn(dr8, C).cpu(), col(rk[0]) * dz + col(rk[1]) * D(r) + col(rk[2]), "dr")
    assert_bf16_close(ops8.to_cn(du8, C).cpu(), col(k[0]) * dz + col(k[1]) * D(u) + col(k[2]), "du")
    assert torch.equal(ops8.to_cn(dz8, C).cpu().double(), dz)
    assert_bf16_close(ops8.to_cn(a8, C).cpu(), col(k[0]) * D(dy) + col(k[1]) * D(u) + col(k[2]), "affine2")


def test_pooling_and_data_bn(dev):
    from sar_amd import ops, ops8
    g = torch.Generator().manual_seed(3)
    C, B, TV, Mp = 40, 6, 75, 2
    y = torch.randn(C, B * TV, generator=g).bfloat16()
    y8 = ops8.from_cn(y.float().to(dev))
    feat = torch.empty((B // Mp, C), device=dev)
    ops8.pool_fwd(y8, C, B, TV, Mp, feat)
    ref = y.double().view(C, B // Mp, Mp * TV).mean(dim=2).t()
    dfeat = torch.randn(B // Mp, C, generator=g).to(dev)
    dy8 = ops8.empty(C, B * TV, dev)
    ops8.pool_bwd(dfeat, C, B, TV, Mp, dy8)
    torch.cuda.synchronize()
    assert rel_err(feat.cpu(), ref) < 1e-5
    dref = (dfeat.cpu().double().t() / (Mp * TV)).view(C, B // Mp, 1).expand(C, B // Mp, Mp * TV).reshape(C, -1)
    assert_bf16_close(ops8.to_cn(dy8, C).cpu(), dref, "pool backward")
    # data_bn apply: identical values (rounded) to the fp32 kernel, incl. the fused bone transform; backward reduce too
    from sar_amd.bone import NTU_BONE_PAIRS
    N, T, M = 3, 9, 2
    x = (0.3 * torch.randn(N, 3, T, 25, M, generator=g)).to(dev)
    bp = np.full(25, -1, dtype=np.int32)
    for v1, v2 in NTU_BONE_PAIRS:
        bp[v1 - 1] = v2 - 1
    bone = torch.from_numpy(bp).to(dev)
    scale, shift = (1 + 0.1 * torch.randn(75, generator=g)).to(dev), (0.1 * torch.randn(75, generator=g)).to(dev)
    n = N * M * T * 25
    for parent in (None, bone):
        h32 = torch.empty((3, n), device=dev)
        ops.data_bn_apply(x, parent, scale, shift, h32)
        h8 = ops8.empty(3, n, dev)
        ops8.data_bn_apply(x, parent, scale, shift, h8)
        torch.cuda.synchronize()
        assert torch.equal(ops8.to_cn(h8, 3), h32.bfloat16().float())
        assert (h8.float()[0, :, 3:] == 0).all()
        dy = torch.randn(3, n, generator=g).bfloat16().float().to(dev)
        mean = (0.05 * torch.randn(75, generator=g)).to(dev)
        p32, p8 = torch.empty((75, N, 2), device=dev), torch.empty((75, N, 2), device=dev)
        ops.data_bn_bwd_reduce(x, parent, dy, mean, p32)
        ops8.data_bn_bwd_reduce(x, parent, ops8.from_cn(dy), mean, p8)
        torch.cuda.synchronize()
        assert torch.equal(p32, p8)


@pytest.mark.parametrize("B,f,T,s", [(2, 64, 13, 1), (2, 64, 14, 2), (2, 128, 9, 2), (1, 256, 7, 1), (2, 64, 22, 2), (4, 256, 75, 1),
                                     (3, 128, 150, 2), (2, 72, 11, 1), (1, 200, 10, 2), (2, 40, 12, 2), (2, 64, 300, 1)])
def test_temporal_conv_weight_gradient(dev, B, f, T, s):
    """dW / dbias of BN -> ReLU -> Conv2D[9,1] (stride 1, and stride 2 with the even-T SAME padding 3): the src operand is
    bf16(relu(fma(g, scale, shift))) of the STORED g, dout the stored du; float64 contraction of those bf16 values."""
    from sar_amd import ops8, _lib as L
    g = torch.Generator().manual_seed(13 * f + T + s)
    x = torch.randn(B, f, T, 25, generator=g).bfloat16()
    sc = 1 + 0.2 * torch.randn(f, generator=g); sh = 0.3 * torch.randn(f, generator=g)
    To, pad, _ = O.same_pad(T, 9, s)
    du = torch.randn(B, f, To, 25, generator=g).bfloat16()
    h = _bf(torch.relu(torch.addcmul(sh.view(1, -1, 1, 1), x.float(), sc.view(1, -1, 1, 1))))
    kernel = torch.zeros(9, 1, f, f, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(f, dtype=torch.float64, requires_grad=True)
    gk, gb = torch.autograd.grad(O.temporal_conv(h, kernel, bias, s), (kernel, bias), du.double())
    flat = torch.zeros(9 * f * f + f, device=dev)
    ops8.conv_wgrad(L.SAR_CONV_TEMPORAL, _cn8(x, dev), _cn8(du, dev), flat, B=B, V=25, T_src=T, T_out=To, Kc=f, M=f, taps=9,
                    stride=s, pad=pad, pro=(sc.to(dev), sh.to(dev)), pro_relu=True, w_stride_tap=f * f, w_stride_c=f,
                    wsize=9 * f * f, bsize=f)
    torch.cuda.synchronize()
    assert rel_err(flat[:9 * f * f].cpu().view(9, 1, f, f), gk) < 1e-5
    assert rel_err(flat[9 * f * f:].cpu(), gb) < 1e-5


@pytest.mark.parametrize("B,cin,f,T,s", [(2, 64, 128, 13, 2), (2, 128, 256, 10, 2), (1, 64, 64, 7, 1), (2, 24, 40, 9, 2), (2, 64, 128, 300, 2)])
def test_residual_conv_weight_gradient(dev, B, cin, f, T, s):
    from sar_amd import ops8, _lib as L
    g = torch.Generator().manual_seed(5 * cin + f)
    x = torch.randn(B, cin, T, 25, generator=g).bfloat16()
    kernel = torch.zeros(1, 1, cin, f, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(f, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double(), O.hwio_to_oihw(kernel), bias, stride=(s, 1))
    To = y.shape[2]
    dr = torch.randn(B, f, To, 25, generator=g).bfloat16()
    gk, gb = torch.autograd.grad(y, (kernel, bias), dr.double())
    flat = torch.zeros(cin * f + f, device=dev)
    ops8.conv_wgrad(L.SAR_CONV_TEMPORAL, _cn8(x, dev), _cn8(dr, dev), flat, B=B, V=25, T_src=T, T_out=To, Kc=cin, M=f, taps=1,
                    stride=s, pad=0, w_stride_tap=0, w_stride_c=f, wsize=cin * f, bsize=f)
    torch.cuda.synchronize()
    assert rel_err(flat[:cin * f].cpu().view(1, 1, cin, f), gk) < 1e-5
    assert rel_err(flat[cin * f:].cpu(), gb) < 1e-5


@pytest.mark.parametrize("B,cin,f,T", [(2, 64, 64, 11), (2, 64, 128, 6), (1, 128, 256, 5), (3, 3, 64, 9), (4, 256, 256, 75),
                                       (2, 40, 72, 9), (2, 64, 64, 300)])
def test_graph_conv_weight_gradient(dev, B, cin, f, T):
    """dW[c][k F + m] = sum z_k[c, n] dg[m, n] with z_k = bf16(fp32 gather of the stored x), dbias[k][m] = sum dg colsum(A_k)"""
    from sar_amd import ops8, _lib as L
    g = torch.Generator().manual_seed(17 * cin + f)
    x = torch.randn(B, cin, T, 25, generator=g).bfloat16()
    dg = torch.randn(B, f, T, 25, generator=g).bfloat16()
    tab = _tables(dev)
    assert tab.slice0_identity
    idx, wt = tab.idx.cpu(), tab.wt.cpu()
    xs = x.float()
    gk = torch.zeros(cin, 3 * f, dtype=torch.float64)
    gb = torch.zeros(3 * f, dtype=torch.float64)
    A = _A()
    for k in range(3):
        z = torch.zeros(B, cin, T, 25)
        for w in range(25):
            acc = wt[k, w, 0] * xs[:, :, :, idx[k, w, 0]]
            for j in range(1, tab.nz[k]):
                acc = torch.addcmul(acc, xs[:, :, :, idx[k, w, j]], wt[k, w, j])
            z[:, :, :, w] = acc
        gk[:, k * f:(k + 1) * f] = torch.einsum("bctv,bmtv->cm", _bf(z), dg.double())
        gb[k * f:(k + 1) * f] = torch.einsum("bmtv,v->m", dg.double(), A[k].double().sum(dim=0))
    flat = torch.zeros(cin * 3 * f + 3 * f, device=dev)
    ops8.conv_wgrad(L.SAR_CONV_GRAPH, _cn8(x, dev), _cn8(dg, dev), flat, B=B, V=25, T_src=T, T_out=T, Kc=cin, M=f, taps=3,
                    tables=tab, w_stride_tap=f, w_stride_c=3 * f, wsize=cin * 3 * f, bsize=3 * f)
    torch.cuda.synchronize()
    assert rel_err(flat[:cin * 3 * f].cpu().view(cin, 3 * f), gk) < 1e-5
    assert rel_err(flat[cin * 3 * f:].cpu(), gb) < 1e-5
    # the slice-0 shortcut (identity read from the raw tile) equals the general path bit for bit
    tab2 = _tables(dev)
    tab2.slice0_identity = False
    flat2 = torch.zeros_like(flat)
    ops8.conv_wgrad(L.SAR_CONV_GRAPH, _cn8(x, dev), _cn8(dg, dev), flat2, B=B, V=25, T_src=T, T_out=T, Kc=cin, M=f, taps=3,
                    tables=tab2, w_stride_tap=f, w_stride_c=3 * f, wsize=cin * 3 * f, bsize=3 * f)
    torch.cuda.synchronize()
    assert torch.equal(flat, flat2)


@pytest.mark.parametrize("cin,f,s,T", [(64, 64, 1, 300), (64, 128, 2, 300), (128, 128, 1, 150), (256, 256, 1, 75)])
def test_cn8_kernels_repeat_bit_for_bit(dev, cin, f, s, T):
    """Race / hazard detector at kernel level: every CN8 conv kernel of a block, launched 8 times on the same inputs with an
    unrelated kernel in between, must reproduce its outputs AND BatchNorm partial sums bit for bit.  (This is the test that
    exposed the packed-fp32 hazard of conv_gemm_cn8's MASK epilogue: csrc/Makefile.)"""
    from sar_amd import ops8, _lib as L
    B = 12
    To, pad, _ = O.same_pad(T, 9, s)
    n_in, n_out = B * T * 25, B * To * 25
    gen = torch.Generator(device=dev).manual_seed(f + T)
    rnd = lambda C, n: ops8.from_cn(torch.randn((C, n), generator=gen, device=dev))
    X, G, dU, dG = rnd(cin, n_in), rnd(f, n_in), rnd(f, n_out), rnd(f, n_in)
    Wt = torch.randn((9, 1, f, f), generator=gen, device=dev) * 0.05
    Wg = torch.randn((1, 1, cin, 3 * f), generator=gen, device=dev) * 0.1
    sc, sh, mean = (1 + 0.2 * torch.randn(f, generator=gen, device=dev), 0.3 * torch.randn(f, generator=gen, device=dev),
                    0.1 * torch.randn(f, generator=gen, device=dev))
    tab, tabT = _tables(dev), _tables(dev, True)
    pw_tf, pw_tb = _pack(dev, Wt, f * f, f, 1, 9, f, f), _pack(dev, Wt, f * f, 1, f, 9, f, f)
    pw_gf, pw_gb = _pack(dev, Wg, f, 3 * f, 1, 3, cin, f), _pack(dev, Wg, f, 1, 3 * f, 3, f, cin)
    junk = torch.empty(8 << 20, dtype=torch.uint8, device=dev)

    def t_fwd():
        out = ops8.empty(f, n_out, dev)
        r = ops8.conv_gemm(L.SAR_CONV_TEMPORAL, G, out, pw_tf, B=B, V=25, T_src=T, T_out=To, Kc=f, M=f, taps=9, stride=s, pad=pad,
                           pro=(sc, sh), pro_relu=True, epi=L.SAR_EPI_STATS)
        return out, r[0]

    def t_dgrad():
        out = ops8.empty(f, n_in, dev)
        r = ops8.conv_gemm(L.SAR_CONV_TEMPORAL, dU, out, pw_tb, B=B, V=25, T_src=To, T_out=T, Kc=f, M=f, taps=9, stride=s, pad=pad,
                           transposed=True, epi=L.SAR_EPI_MASK, aux=G, aux_affine=(sc, sh), aux_mean=mean)
        return out, r[0]

    def g_fwd():
        out = ops8.empty(f, n_in, dev)
        r = ops8.conv_gemm(L.SAR_CONV_GRAPH, X, out, pw_gf, B=B, V=25, T_src=T, T_out=T, Kc=cin, M=f, taps=3, tables=tab,
                           epi=L.SAR_EPI_STATS)
        return out, r[0]

    def g_dgrad():
        out = ops8.empty(cin, n_in, dev)
        ops8.conv_gemm(L.SAR_CONV_GRAPH, dG, out, pw_gb, B=B, V=25, T_src=T, T_out=T, Kc=f, M=cin, taps=3, tables=tabT,
                       epi=L.SAR_EPI_ADD, aux=X)
        return (out,)

    def t_wgrad():
        flat = torch.zeros(9 * f * f + f, device=dev)
        ops8.conv_wgrad(L.SAR_CONV_TEMPORAL, G, dU, flat, B=B, V=25, T_src=T, T_out=To, Kc=f, M=f, taps=9, stride=s, pad=pad,
                        pro=(sc, sh), pro_relu=True, w_stride_tap=f * f, w_stride_c=f, wsize=9 * f * f, bsize=f)
        return (flat,)

    def g_wgrad():
        flat = torch.zeros(cin * 3 * f + 3 * f, device=dev)
        ops8.conv_wgrad(L.SAR_CONV_GRAPH, X, dG, flat, B=B, V=25, T_src=T, T_out=T, Kc=cin, M=f, taps=3, tables=tab,
                        w_stride_tap=f, w_stride_c=3 * f, wsize=cin * 3 * f, bsize=3 * f)
        return (flat,)

    for name, fn in (("temporal fwd", t_fwd), ("temporal dgrad + mask", t_dgrad), ("graph fwd", g_fwd), ("graph dgrad + add", g_dgrad),
                     ("temporal wgrad", t_wgrad), ("graph wgrad", g_wgrad)):
        ref = None
        for rep in range(8):
            if rep % 2:
                junk.random_(0, 255)
            res = [t.clone() for t in fn()]
            torch.cuda.synchronize()
            if ref is None:
                ref = res
            else:
                assert all(torch.equal(a, b) for a, b in zip(ref, res)), "%s is not repeatable (rep %d)" % (name, rep)


def _run_in_env(code, env_extra):
    """a fresh interpreter with `env_extra` (the switch is read once per process); returns what the snippet saved"""
    import pickle
    import subprocess
    import sys
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.pkl")
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env_extra), capture_output=True, text=True,
                           timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
        return pickle.load(open(out, "rb"))


_GRAPH_AB = r'''
import sys, pickle, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "skeleton-action-recognition_amd")
import test_gpu_cn8 as T
from sar_amd import ops8, _lib as L
dev = torch.device("cuda", 0)
res = {}
for (B, cin, f, Tt) in [(3, 3, 64, 13), (2, 64, 64, 23), (2, 64, 128, 7), (1, 128, 256, 5), (4, 256, 256, 75), (2, 40, 72, 9)]:
    g = torch.Generator().manual_seed(B * 1000 + cin)
    x = torch.randn(B, cin, Tt, 25, generator=g).bfloat16()
    kernel = torch.randn(1, 1, cin, 3 * f, generator=g) * 0.1
    bias = torch.randn(3 * f, generator=g) * 0.1
    dout = torch.randn(B, f, Tt, 25, generator=g).bfloat16()
    add = torch.randn(B, cin, Tt, 25, generator=g).bfloat16()
    tab, tabT = T._tables(dev), T._tables(dev, True)
    out = ops8.empty(f, B * Tt * 25, dev)
    r = ops8.conv_gemm(L.SAR_CONV_GRAPH, T._cn8(x, dev), out, T._pack(dev, kernel, f, 3 * f, 1, 3, cin, f), B=B, V=25, T_src=Tt, T_out=Tt,
                       Kc=cin, M=f, taps=3, bias=bias.to(dev), tables=tab, epi=L.SAR_EPI_STATS)
    dx = ops8.empty(cin, B * Tt * 25, dev)
    ops8.conv_gemm(L.SAR_CONV_GRAPH, T._cn8(dout, dev), dx, T._pack(dev, kernel, f, 1, 3 * f, 3, f, cin), B=B, V=25, T_src=Tt, T_out=Tt,
                   Kc=f, M=cin, taps=3, tables=tabT, epi=L.SAR_EPI_ADD, aux=T._cn8(add, dev))
    torch.cuda.synchronize()
    res[(B, cin, f, Tt)] = (out.cpu(), r[0].cpu(), dx.cpu(), tab.g_flags, tabT.g_flags)
pickle.dump(res, open(sys.argv[1], "wb"))
'''


def test_graph_conv_read_gather_equals_the_unit_builder_bit_for_bit(dev):
    """csrc/conv_graph_cn8.hip (operands of trivial gather lists read straight from the raw tile, the few others built as
    virtual joints) against conv_graph_cn8_kernel (every gathered tile built): same tiles, same fp32 chains, same MFMA order --
    outputs, BatchNorm partial sums and data gradients must agree bit for bit.  NTU: 2 dense lists forward, 8 transposed."""
    from sar_amd import _lib as L
    new = _run_in_env(_GRAPH_AB, {"SAR_GRAPH_READ_GATHER": "1"})
    old = _run_in_env(_GRAPH_AB, {"SAR_GRAPH_READ_GATHER": "0"})
    for key in new:
        o1, p1, d1, fl, flT = new[key]
        o0, p0, d0, _, _ = old[key]
        assert fl & L.SAR_GRAPH_FEW_DENSE and (fl >> L.SAR_GRAPH_FEW_DENSE_SHIFT) & 0xff == 2
        assert flT & L.SAR_GRAPH_FEW_DENSE and (flT >> L.SAR_GRAPH_FEW_DENSE_SHIFT) & 0xff == 8
        assert torch.equal(o1, o0), key
        assert torch.equal(p1, p0), key
        assert torch.equal(d1, d0), key


_GRAPH_NONFINITE = r'''
import sys, pickle, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "skeleton-action-recognition_amd")
import test_gpu_cn8 as T
from sar_amd import ops8, _lib as L
dev = torch.device("cuda", 0)
res = {}
for (B, cin, f, Tt, bad) in [(2, 64, 64, 11, 4), (1, 128, 256, 5, 1), (2, 40, 72, 9, 8)]:
    g = torch.Generator().manual_seed(B * 1000 + cin)
    x = torch.randn(B, cin, Tt, 25, generator=g).bfloat16()
    kernel = torch.randn(1, 1, cin, 3 * f, generator=g) * 0.1
    tab = T._tables(dev)
    outs = []
    for poison in (False, True):
        xx = x.clone()
        xx[0, :, bad] = float("inf") if poison else 0.0
        out = ops8.empty(f, B * Tt * 25, dev)
        ops8.conv_gemm(L.SAR_CONV_GRAPH, T._cn8(xx, dev), out, T._pack(dev, kernel, f, 3 * f, 1, 3, cin, f), B=B, V=25, T_src=Tt,
                       T_out=Tt, Kc=cin, M=f, taps=3, tables=tab)
        torch.cuda.synchronize()
        outs.append(T._back(out, f, B, Tt).cpu())
    res[(B, cin, f, Tt, bad)] = (outs[0], outs[1], tab.g_flags)
pickle.dump(res, open(sys.argv[1], "wb"))
'''


@pytest.mark.parametrize("read_gather", ["0", "1"])
def test_graph_conv_keeps_non_finite_values_inside_their_frame(dev, read_gather):
    """The matrix-core adjacency gather (conv_graph_cn8_kernel, SAR_GRAPH_READ_GATHER=0) reads 32 joints of a 25-joint frame:
    the 7 extra columns belong to the NEXT frame and meet zero rows of A_k -- but 0 x Inf = NaN.  The operand fragment is
    masked to joints < V, so a frame of Inf changes that frame's outputs only, as in the vector gather and in the
    reference operator (models/stgcn.py:26-31: the contraction runs over the joints of ONE frame)."""
    from sar_amd import _lib as L
    got = _run_in_env(_GRAPH_NONFINITE, {"SAR_GRAPH_READ_GATHER": read_gather})
    for (B, cin, f, Tt, bad), (clean, poisoned, fl) in got.items():
        assert fl & L.SAR_GRAPH_WT_BF16_EXACT
        keep = [t for t in range(Tt) if t != bad]
        assert torch.isfinite(poisoned[0][:, keep]).all(), (B, cin, f, Tt)
        assert torch.equal(poisoned[0][:, keep], clean[0][:, keep])
        assert torch.equal(poisoned[1:], clean[1:])
        assert not torch.isfinite(poisoned[0][:, bad]).any()


@pytest.mark.parametrize("B,cin,f,T", [(2, 64, 64, 11), (2, 64, 128, 6), (1, 128, 256, 5), (4, 256, 256, 75), (2, 40, 72, 9)])
def test_graph_data_gradient_gated_epilogue(dev, B, cin, f, T):
    """SAR_EPI_ADD_GATE (round 4): out = gate(W^T dg . A^T + aux) with the gate bytes a block tail wrote (bit j of byte (plane, n) =
    channel 8 plane + j), and the BatchNorm-backward sums of that tail, (sum out, sum out (u - mean)), reduced from the same
    accumulators.  Against the plain SAR_EPI_ADD launch: the gated output must equal the gated ADD output BIT FOR BIT, the sums the
    float64 sums of it within 1e-4."""
    from sar_amd import ops8, _lib as L
    g = torch.Generator().manual_seed(11 * cin + f)
    kernel = torch.randn(1, 1, cin, 3 * f, generator=g) * 0.1
    dout = torch.randn(B, f, T, 25, generator=g).bfloat16()
    add = torch.randn(B, cin, T, 25, generator=g).bfloat16()
    u = torch.randn(B, cin, T, 25, generator=g).bfloat16()
    mean = torch.randn(cin, generator=g) * 0.2
    keep = torch.rand(B, cin, T, 25, generator=g) > 0.4                      # the ReLU mask of the tail below
    n = B * T * 25
    planes = (cin + 7) // 8
    kb = torch.zeros(planes * 8, n, dtype=torch.int32)
    kb[:cin] = keep.permute(1, 0, 2, 3).reshape(cin, n).int()
    mask = (kb.view(planes, 8, n) << torch.arange(8, dtype=torch.int32).view(1, 8, 1)).sum(dim=1).to(torch.uint8).contiguous().to(dev)
    tabT = _tables(dev, True)
    pw = _pack(dev, kernel, f, 1, 3 * f, 3, f, cin)
    args = dict(B=B, V=25, T_src=T, T_out=T, Kc=f, M=cin, taps=3, tables=tabT)
    plain, gated = ops8.empty(cin, n, dev), ops8.empty(cin, n, dev)
    ops8.conv_gemm(L.SAR_CONV_GRAPH, _cn8(dout, dev), plain, pw, epi=L.SAR_EPI_ADD, aux=_cn8(add, dev), **args)
    pm = ops8.conv_gemm(L.SAR_CONV_GRAPH, _cn8(dout, dev), gated, pw, epi=L.SAR_EPI_ADD_GATE, aux=_cn8(add, dev), aux2=_cn8(u, dev),
                        aux_mask=mask, aux_mean=mean.to(dev), **args)
    torch.cuda.synchronize()
    want = torch.where(keep, _back(plain, cin, B, T), torch.zeros(()).double())
    got = _back(gated, cin, B, T)
    assert torch.equal(got, want)
    part = pm[0].cpu().double().sum(dim=1)                                   # (cin, 2)
    # the sums run over the STORED (bfloat16) values: float64 sums of the output, up to fp32 accumulation
    s1 = want.sum(dim=(0, 2, 3))
    s2 = (want * (u.double() - mean.double().view(1, -1, 1, 1))).sum(dim=(0, 2, 3))
    scale1 = want.abs().sum(dim=(0, 2, 3)).max()
    assert (part[:, 0] - s1).abs().max() <= 1e-5 * scale1 and (part[:, 1] - s2).abs().max() <= 2e-5 * scale1


def test_fused_tail_reduction_matches_the_separate_passes(dev):
    """SAR_CN8_FUSE_TAIL (default on): the graph data gradient of block i gates the output gradient of block i - 1 and reduces its
    BatchNorm-backward sums; the unfused schedule (SAR_CN8_FUSE_TAIL=0: bn_add_relu_bwd_reduce pass + masked-gradient write) sums
    the same stored values in another order.  A 1e-6 difference of a BatchNorm-backward constant moves a few of the next
    bfloat16-STORED gradients by one ulp (2^-9), and ten blocks of heavily cancelling sums amplify that ~3x per block (measured:
    7e-7 at l8.bn1 ... 8e-3 at l0): the top blocks must agree tightly, the bottom ones within what bf16 storage makes of any
    re-ordering (two runs of ONE schedule agree bit for bit)."""
    code = r'''
import sys, pickle, torch
sys.path.insert(0, "."); sys.path.insert(0, "skeleton-action-recognition_amd")
from sar_amd.stgcn import STGCN
from sar_amd.train import synthetic_clips
dev = torch.device("cuda", 0)
eng = STGCN(num_classes=60, device=dev, seed=0, mfma="bf16")
x, y = synthetic_clips(4, dev, seed=3, T=60)
logits, loss = eng.loss_and_grad(x, y)
torch.cuda.synchronize()
pickle.dump(dict(g={k: v.cpu().clone() for k, v in eng.g.items()}, logits=logits.cpu(), loss=loss.cpu()), open(sys.argv[1], "wb"))
'''
    on = _run_in_env(code, {"SAR_CN8_FUSE_TAIL": "1"})
    on2 = _run_in_env(code, {"SAR_CN8_FUSE_TAIL": "1"})
    off = _run_in_env(code, {"SAR_CN8_FUSE_TAIL": "0"})
    assert torch.equal(on["logits"], off["logits"]) and torch.equal(on["loss"], off["loss"])
    worst = {}
    for k, v in on["g"].items():
        assert torch.equal(v, on2["g"][k]), k                       # the fused schedule repeats bit for bit
        if k.endswith(".bias") and ("tcn" in k or "res." in k or "gcn" in k):
            continue                                                # a bias in front of a BatchNorm: its gradient is rounding noise around 0
        rel = (v - off["g"][k]).abs().max().item() / max(off["g"][k].abs().max().item(), 1e-30)
        blk = k.split(".")[0]
        worst[blk] = max(worst.get(blk, 0.0), rel)
    print("fused vs separate tail reduction, worst relative gradient difference per block:", {k: "%.1e" % v for k, v in worst.items()})
    assert worst["l9"] == 0.0 and worst["logits"] == 0.0           # nothing above the first fused epilogue changes
    assert worst["l8"] < 1e-3 and worst["l7"] < 2e-3
    assert max(worst.values()) < 5e-2
