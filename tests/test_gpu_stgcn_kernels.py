"""GPU parity: every ST-GCN HIP kernel (through the C ABI) against the CPU oracle on seeded inputs.

Tolerance: north_star asks for fp32 logits/grads within 1e-4 relative of the reference's CPU
forward/backward; single ops are held to 2e-5 (norm-wise: max|a-b| / max|b|) so the stack has room.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import stgcn as O
from util import to_cn, from_cn, rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from sar_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _tables(dev, transpose=False):
    from sar_amd import ops
    from oracle.graph import spatial_adjacency
    return ops.GraphTables(spatial_adjacency().astype(np.float32), dev, transpose)


def _A():
    from oracle.graph import spatial_adjacency
    return torch.tensor(spatial_adjacency().astype(np.float32))


@pytest.mark.parametrize("B,cin,f,T", [(3, 3, 64, 13), (2, 64, 64, 10), (2, 64, 128, 7), (1, 128, 256, 5), (4, 256, 256, 3),
                                       (2, 40, 72, 9), (1, 12, 200, 5), (2, 64, 44, 6), (1, 30, 50, 4)])   # ragged row / channel blocks
def test_graph_conv_forward_and_stats(dev, B, cin, f, T):
    from sar_amd import ops, _lib as L
    g = torch.Generator().manual_seed(B * 1000 + cin)
    x = torch.randn(B, cin, T, 25, generator=g)
    kernel = torch.randn(1, 1, cin, 3 * f, generator=g) * 0.1
    bias = torch.randn(3 * f, generator=g) * 0.1
    ref = O.graph_conv_td(x.double(), kernel.double(), bias.double(), _A().double())
    xc = to_cn(x).to(dev)
    out = torch.empty((f, B * T * 25), device=dev)
    r = ops.conv_gemm(L.SAR_CONV_GRAPH, xc, out, kernel.to(dev), f, 3 * f, B=B, V=25, T_src=T, T_out=T, Kc=cin, M=f, taps=3,
                      bias=bias.to(dev), tables=_tables(dev), epi=L.SAR_EPI_STATS)
    torch.cuda.synchronize()
    assert rel_err(from_cn(out.cpu(), B, T, 25), ref) < TOL
    # BN statistics from the epilogue partials
    mean = torch.empty(f, device=dev); rstd = torch.empty(f, device=dev)
    scale = torch.empty(f, device=dev); shift = torch.empty(f, device=dev)
    gamma = (1 + 0.1 * torch.randn(f, generator=g)).to(dev); beta = (0.1 * torch.randn(f, generator=g)).to(dev)
    rm = torch.zeros(f, device=dev); rv = torch.ones(f, device=dev)
    n = B * T * 25
    ops.bn_finalize(r[0], r[1], f, n, 1e-3, 0.99, True, gamma, beta, rm, rv, mean, rstd, scale, shift)
    torch.cuda.synchronize()
    m_ref = ref.mean(dim=(0, 2, 3)); v_ref = ref.var(dim=(0, 2, 3), unbiased=False)
    assert rel_err(mean.cpu(), m_ref) < TOL
    assert rel_err(rstd.cpu(), torch.rsqrt(v_ref + 1e-3)) < TOL
    assert rel_err(scale.cpu(), gamma.cpu().double() * torch.rsqrt(v_ref + 1e-3)) < TOL
    assert rel_err(rm.cpu(), 0.01 * m_ref) < TOL
    assert rel_err(rv.cpu(), 0.99 + 0.01 * v_ref * n / (n - 1)) < TOL


@pytest.mark.parametrize("B,f,T,s", [(2, 64, 13, 1), (3, 64, 14, 2), (2, 128, 9, 2), (1, 256, 6, 1), (2, 128, 300, 2),
                                     (2, 72, 11, 1), (1, 200, 9, 2), (2, 44, 7, 1), (1, 50, 10, 2)])
def test_temporal_conv_forward_fused_bn_relu(dev, B, f, T, s):
    from sar_amd import ops, _lib as L
    g = torch.Generator().manual_seed(f + T + s)
    x = torch.randn(B, f, T, 25, generator=g)
    sc = 1 + 0.2 * torch.randn(f, generator=g); sh = 0.3 * torch.randn(f, generator=g)
    kernel = torch.randn(9, 1, f, f, generator=g) * 0.05
    bias = torch.randn(f, generator=g) * 0.1
    h = torch.relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    ref = O.temporal_conv(h, kernel.double(), bias.double(), s)
    To, pad, _ = O.same_pad(T, 9, s)
    out = torch.empty((f, B * To * 25), device=dev)
    r = ops.conv_gemm(L.SAR_CONV_TEMPORAL, to_cn(x).to(dev), out, kernel.to(dev), f * f, f, B=B, V=25, T_src=T, T_out=To,
                      Kc=f, M=f, taps=9, stride=s, pad=pad, bias=bias.to(dev), pro=(sc.to(dev), sh.to(dev)),
                      pro_relu=True, epi=L.SAR_EPI_STATS)
    torch.cuda.synchronize()
    assert rel_err(from_cn(out.cpu(), B, To, 25), ref) < TOL
    part = r[0].cpu().double().sum(dim=1)
    assert rel_err(part[:, 0], ref.sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(part[:, 1], (ref * ref).sum(dim=(0, 2, 3))) < TOL


@pytest.mark.parametrize("B,cin,f,T,s", [(2, 64, 128, 13, 2), (2, 128, 256, 10, 2), (1, 3, 64, 7, 1)])
def test_residual_conv_forward(dev, B, cin, f, T, s):
    from sar_amd import ops, _lib as L
    g = torch.Generator().manual_seed(cin + f)
    x = torch.randn(B, cin, T, 25, generator=g)
    kernel = torch.randn(1, 1, cin, f, generator=g) * 0.1
    bias = torch.randn(f, generator=g) * 0.1
    ref = F.conv2d(x.double(), O.hwio_to_oihw(kernel.double()), bias.double(), stride=(s, 1))
    To = ref.shape[2]
    out = torch.empty((f, B * To * 25), device=dev)
    ops.conv_gemm(L.SAR_CONV_TEMPORAL, to_cn(x).to(dev), out, kernel.to(dev), 0, f, B=B, V=25, T_src=T, T_out=To, Kc=cin, M=f,
                  taps=1, stride=s, pad=0, bias=bias.to(dev))
    torch.cuda.synchronize()
    assert rel_err(from_cn(out.cpu(), B, To, 25), ref) < TOL


@pytest.mark.parametrize("B,cin,f,T", [(2, 64, 64, 11), (2, 64, 128, 6), (1, 128, 256, 5), (3, 3, 64, 9), (4, 256, 256, 75),
                                       (2, 3, 40, 5), (1, 2, 72, 4), (5, 3, 64, 300), (2, 4, 100, 7),
                                       (2, 40, 72, 9), (1, 72, 200, 5), (2, 44, 40, 6)])
def test_graph_conv_gradients(dev, B, cin, f, T):
    """data gradient (A^T gather lists + W^T) and weight/bias gradients of GraphConvTD."""
    from sar_amd import ops, _lib as L
    g = torch.Generator().manual_seed(7 * cin + f)
    x = torch.randn(B, cin, T, 25, generator=g).double().requires_grad_(True)
    kernel = (torch.randn(1, 1, cin, 3 * f, generator=g) * 0.1).double().requires_grad_(True)
    bias = (torch.randn(3 * f, generator=g) * 0.1).double().requires_grad_(True)
    dout = torch.randn(B, f, T, 25, generator=g)
    y = O.graph_conv_td(x, kernel, bias, _A().double())
    gx, gk, gb = torch.autograd.grad(y, (x, kernel, bias), dout.double())
    n = B * T * 25
    # data gradient
    gT = torch.empty((3 * f, cin), device=dev)
    ops.transpose(kernel.detach().float().to(dev).contiguous(), gT, 1, cin, 3 * f)
    dx = torch.empty((cin, n), device=dev)
    add = torch.randn(cin, n, generator=g)
    ops.conv_gemm(L.SAR_CONV_GRAPH, to_cn(dout).to(dev), dx, gT, f * cin, cin, B=B, V=25, T_src=T, T_out=T, Kc=f, M=cin,
                  taps=3, tables=_tables(dev, True), epi=L.SAR_EPI_ADD, aux=add.to(dev))
    torch.cuda.synchronize()
    assert rel_err(from_cn((dx.cpu() - add), B, T, 25), gx) < TOL
    # weight + bias gradient
    flat = torch.zeros(cin * 3 * f + 3 * f, device=dev)
    ops.conv_wgrad(L.SAR_CONV_GRAPH, to_cn(x.detach().float()).to(dev), to_cn(dout).to(dev), flat, B=B, V=25, T_src=T, T_out=T,
                   Kc=cin, M=f, taps=3, tables=_tables(dev), w_stride_tap=f, w_stride_c=3 * f, wsize=cin * 3 * f,
                   bsize=3 * f)
    torch.cuda.synchronize()
    assert rel_err(flat[:cin * 3 * f].cpu().view(1, 1, cin, 3 * f), gk) < TOL
    assert rel_err(flat[cin * 3 * f:].cpu(), gb) < TOL


@pytest.mark.parametrize("B,cin,f,T", [(2, 64, 64, 12), (2, 64, 128, 8), (1, 128, 256, 4), (4, 256, 256, 76), (2, 40, 72, 8)])
def test_graph_data_gradient_gated_epilogue_f32(dev, B, cin, f, T):
    """SAR_EPI_ADD_GATE of sar_conv_gemm_f32 (include/sar_hip.h): out = gate(W^T dg . A^T + aux) with the ReLU-mask bytes of the
    block below (one bit per column, sar_bn_add_relu_fwd_mask_f32's layout), partials = (sum out, sum out (aux2 - mean)): the
    stored tensor equals the SAR_EPI_ADD result gated afterwards bit for bit, the sums equal fp64 sums of it to 1e-6."""
    from sar_amd import ops, _lib as L
    g = torch.Generator().manual_seed(11 * cin + f)
    n = B * T * 25
    assert n % 4 == 0
    kernel = torch.randn(1, 1, cin, 3 * f, generator=g) * 0.1
    dout = torch.randn(f, n, generator=g).to(dev)
    add = torch.randn(cin, n, generator=g).to(dev)
    u = torch.randn(cin, n, generator=g).to(dev)
    mean = (0.1 * torch.randn(cin, generator=g)).to(dev)
    keep = (torch.rand(cin, n, generator=g) > 0.4).to(dev)
    mask = (keep.view(cin, n // 4, 4).to(torch.int32) * torch.tensor([1, 2, 4, 8], device=dev, dtype=torch.int32)).sum(dim=2).to(torch.uint8).contiguous()
    gT = torch.empty((3 * f, cin), device=dev)
    ops.transpose(kernel.to(dev).contiguous(), gT, 1, cin, 3 * f)
    args = dict(B=B, V=25, T_src=T, T_out=T, Kc=f, M=cin, taps=3, tables=_tables(dev, True))
    plain, gated = torch.empty((cin, n), device=dev), torch.empty((cin, n), device=dev)
    ops.conv_gemm(L.SAR_CONV_GRAPH, dout, plain, gT, f * cin, cin, epi=L.SAR_EPI_ADD, aux=add, **args)
    pm = ops.conv_gemm(L.SAR_CONV_GRAPH, dout, gated, gT, f * cin, cin, epi=L.SAR_EPI_ADD_GATE, aux=add, aux2=u, aux_mask=mask,
                       aux_mean=mean, **args)
    torch.cuda.synchronize()
    want = torch.where(keep, plain, torch.zeros_like(plain))
    assert torch.equal(gated, want)
    part = pm[0].double().sum(dim=1).cpu()
    s1 = want.double().sum(dim=1).cpu()
    s2 = (want.double() * (u.double() - mean.double().view(-1, 1))).sum(dim=1).cpu()
    assert rel_err(part[:, 0], s1) < 1e-6 and rel_err(part[:, 1], s2) < 1e-6


@pytest.mark.parametrize("B,cin,f,T", [(2, 64, 128, 12), (3, 128, 256, 9), (2, 64, 128, 300), (1, 3, 64, 7), (5, 64, 128, 33)])
def test_graph_data_gradient_adds_an_even_frame_skip_gradient(dev, B, cin, f, T):
    """SAR_GRAPH_AUX_EVEN_FRAMES (include/sar_hip.h): `aux` holds the EVEN output frames only -- the skip gradient through a stride-2
    1x1 residual convolution (models/stgcn.py:47-56), which the engines compute as a dense product over the To frames.  The result
    must equal, bit for bit, the plain SAR_EPI_ADD / SAR_EPI_ADD_GATE launch fed with the zero-interleaved tensor (odd T, ragged
    tiles, the 3-channel generic epilogue, both arithmetics through tests/conftest.py)."""
    from sar_amd import ops, _lib as L
    g = torch.Generator().manual_seed(5 * cin + f + T)
    n, Ta = B * T * 25, (T + 1) // 2
    kernel = torch.randn(1, 1, cin, 3 * f, generator=g) * 0.1
    dout = torch.randn(f, n, generator=g).to(dev)
    compact = torch.randn(cin, B * Ta * 25, generator=g)
    full = torch.zeros(cin, B, T, 25)
    full[:, :, 0::2] = compact.view(cin, B, Ta, 25)
    full = full.reshape(cin, n).to(dev)
    compact = compact.to(dev)
    gT = torch.empty((3 * f, cin), device=dev)
    ops.transpose(kernel.to(dev).contiguous(), gT, 1, cin, 3 * f)
    args = dict(B=B, V=25, T_src=T, T_out=T, Kc=f, M=cin, taps=3, tables=_tables(dev, True))
    want, got = torch.empty((cin, n), device=dev), torch.full((cin, n), float("nan"), device=dev)
    ops.conv_gemm(L.SAR_CONV_GRAPH, dout, want, gT, f * cin, cin, epi=L.SAR_EPI_ADD, aux=full, **args)
    ops.conv_gemm(L.SAR_CONV_GRAPH, dout, got, gT, f * cin, cin, epi=L.SAR_EPI_ADD, aux=compact, aux_even_frames=True, **args)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    if cin % 8 == 0 and n % 4 == 0:
        u = torch.randn(cin, n, generator=g).to(dev)
        mean = (0.1 * torch.randn(cin, generator=g)).to(dev)
        keep = (torch.rand(cin, n, generator=g) > 0.4).to(dev)
        mask = (keep.view(cin, n // 4, 4).to(torch.int32) * torch.tensor([1, 2, 4, 8], device=dev, dtype=torch.int32)).sum(dim=2).to(torch.uint8).contiguous()
        want2, got2 = torch.empty((cin, n), device=dev), torch.full((cin, n), float("nan"), device=dev)
        p1 = ops.conv_gemm(L.SAR_CONV_GRAPH, dout, want2, gT, f * cin, cin, epi=L.SAR_EPI_ADD_GATE, aux=full, aux2=u, aux_mask=mask,
                           aux_mean=mean, **args)
        p2 = ops.conv_gemm(L.SAR_CONV_GRAPH, dout, got2, gT, f * cin, cin, epi=L.SAR_EPI_ADD_GATE, aux=compact, aux2=u, aux_mask=mask,
                           aux_mean=mean, aux_even_frames=True, **args)
        torch.cuda.synchronize()
        assert torch.equal(got2, want2) and torch.equal(p1[0], p2[0])


@pytest.mark.parametrize("B,f,T,s", [(2, 64, 13, 1), (2, 64, 14, 2), (2, 128, 9, 2), (1, 256, 7, 1), (2, 64, 11, 2), (4, 256, 75, 1), (3, 128, 150, 2),
                                     (2, 72, 11, 1), (1, 200, 9, 2), (2, 44, 12, 2)])
def test_temporal_conv_gradients(dev, B, f, T, s):
    """weight/bias gradient (with the folded BN+ReLU operand) and the transposed-conv data gradient with the
    fused ReLU mask + BN-backward reductions."""
    from sar_amd import ops, _lib as L
    g = torch.Generator().manual_seed(11 * f + T + s)
    gx = torch.randn(B, f, T, 25, generator=g).double()
    sc = (1 + 0.2 * torch.randn(f, generator=g)).double(); sh = (0.3 * torch.randn(f, generator=g)).double()
    kernel = (torch.randn(9, 1, f, f, generator=g) * 0.05).double().requires_grad_(True)
    bias = (torch.randn(f, generator=g) * 0.1).double().requires_grad_(True)
    pre = (gx * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    h = torch.relu(pre)
    y = O.temporal_conv(h, kernel, bias, s)
    To, pad, _ = O.same_pad(T, 9, s)
    du = torch.randn(B, f, To, 25, generator=g)
    g_pre, g_k, g_b = torch.autograd.grad(y, (pre, kernel, bias), du.double())   # g_pre = dh * relu mask
    n_in = B * T * 25
    scd, shd = sc.float().to(dev), sh.float().to(dev)
    flat = torch.zeros(9 * f * f + f, device=dev)
    ops.conv_wgrad(L.SAR_CONV_TEMPORAL, to_cn(gx.float()).to(dev), to_cn(du).to(dev), flat, B=B, V=25, T_src=T, T_out=To, Kc=f,
                   M=f, taps=9, stride=s, pad=pad, pro=(scd, shd), pro_relu=True, w_stride_tap=f * f, w_stride_c=f,
                   wsize=9 * f * f, bsize=f)
    torch.cuda.synchronize()
    assert rel_err(flat[:9 * f * f].cpu().view(9, 1, f, f), g_k) < TOL
    assert rel_err(flat[9 * f * f:].cpu(), g_b) < TOL
    wT = torch.empty((9, f, f), device=dev)
    ops.transpose(kernel.detach().float().to(dev).contiguous(), wT, 9, f, f)
    dz1 = torch.empty((f, n_in), device=dev)
    gcn_d = to_cn(gx.float()).to(dev)
    pm = ops.conv_gemm(L.SAR_CONV_TEMPORAL, to_cn(du).to(dev), dz1, wT, f * f, f, B=B, V=25, T_src=To, T_out=T, Kc=f, M=f,
                       taps=9, stride=s, pad=pad, transposed=True, epi=L.SAR_EPI_MASK, aux=gcn_d, aux_affine=(scd, shd))
    torch.cuda.synchronize()
    assert rel_err(from_cn(dz1.cpu(), B, T, 25), g_pre) < TOL
    part = pm[0].cpu().double().sum(dim=1)
    assert rel_err(part[:, 0], g_pre.sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(part[:, 1], (g_pre * gx).sum(dim=(0, 2, 3))) < 1e-4   # aux_mean=None -> raw second moment


@pytest.mark.parametrize("B,cin,f,T,s", [(2, 64, 128, 13, 2), (2, 128, 256, 10, 2)])
def test_residual_conv_gradients(dev, B, cin, f, T, s):
    from sar_amd import ops, _lib as L
    g = torch.Generator().manual_seed(cin * 3 + f)
    x = torch.randn(B, cin, T, 25, generator=g).double().requires_grad_(True)
    kernel = (torch.randn(1, 1, cin, f, generator=g) * 0.1).double().requires_grad_(True)
    bias = (torch.randn(f, generator=g) * 0.1).double().requires_grad_(True)
    y = F.conv2d(x, O.hwio_to_oihw(kernel), bias, stride=(s, 1))
    To = y.shape[2]
    dr = torch.randn(B, f, To, 25, generator=g)
    gx, gk, gb = torch.autograd.grad(y, (x, kernel, bias), dr.double())
    flat = torch.zeros(cin * f + f, device=dev)
    ops.conv_wgrad(L.SAR_CONV_TEMPORAL, to_cn(x.detach().float()).to(dev), to_cn(dr).to(dev), flat, B=B, V=25, T_src=T,
                   T_out=To, Kc=cin, M=f, taps=1, stride=s, pad=0, w_stride_tap=0, w_stride_c=f, wsize=cin * f, bsize=f)
    rT = torch.empty((f, cin), device=dev)
    ops.transpose(kernel.detach().float().to(dev).contiguous(), rT, 1, cin, f)
    dx = torch.empty((cin, B * T * 25), device=dev)
    ops.conv_gemm(L.SAR_CONV_TEMPORAL, to_cn(dr).to(dev), dx, rT, 0, cin, B=B, V=25, T_src=To, T_out=T, Kc=f, M=cin, taps=1,
                  stride=s, pad=0, transposed=True)
    torch.cuda.synchronize()
    assert rel_err(flat[:cin * f].cpu().view(1, 1, cin, f), gk) < TOL
    assert rel_err(flat[cin * f:].cpu(), gb) < TOL
    assert rel_err(from_cn(dx.cpu(), B, T, 25), gx) < TOL


@pytest.mark.parametrize("C,n,res", [(64, 2 * 300 * 25, True), (128, 40000, False), (20, 777, True), (256, 9375 * 4, True)])
def test_folded_bn_backward_finalisation(dev, C, n, res):
    """sar_bn_add_relu_bwd_reduce_tail_f32 / _cn8 (the reduce kernel's last workgroup per channel finalises: dgamma, dbeta,
    k1..k3 of the block's BatchNorm and of the residual branch's) against the separate sar_bn_bwd_finalize_f32 launches on the
    same partials -- the same fp64 sums in another order: equal to 1e-6 -- run several times (the ticket array must come back
    to zero; a stale partial would show as a wrong coefficient)."""
    from sar_amd import ops, ops8
    from sar_amd.stgcn import _BN
    g = torch.Generator().manual_seed(C + n)
    rnd = lambda *s: torch.randn(*s, generator=g)
    for cn8 in (False, True):
        cast = (lambda t: t.bfloat16().float()) if cn8 else (lambda t: t)
        u, r, dy, y = cast(rnd(C, n)).to(dev), cast(rnd(C, n)).to(dev), cast(rnd(C, n)).to(dev), cast(rnd(C, n)).to(dev)
        mu, mr = (0.1 * rnd(C)).to(dev), (0.1 * rnd(C)).to(dev)
        gam, rgam = (1 + 0.2 * rnd(C)).to(dev), (1 + 0.2 * rnd(C)).to(dev)
        bn, rbn, bn0, rbn0 = _BN(C, dev), _BN(C, dev), _BN(C, dev), _BN(C, dev)
        for b in (bn, rbn, bn0, rbn0):
            b.rstd.copy_(0.5 + torch.rand(C, generator=g).to(dev))
        rbn0.rstd.copy_(rbn.rstd), bn0.rstd.copy_(bn.rstd)
        z = lambda: torch.zeros(C, device=dev)
        if cn8:
            a8 = [ops8.from_cn(t) for t in (dy, y, u, r)]
            red = lambda tail=None: ops8.bn_add_relu_bwd_reduce(a8[0], a8[1], a8[2], a8[3] if res else None, C, mu, mr if res else None,
                                                                tail=tail)
        else:
            red = lambda tail=None: ops.bn_add_relu_bwd_reduce(dy, y, u, r if res else None, mu, mr if res else None, tail=tail)
        part, nparts = red()
        dg0, db0, rdg0, rdb0 = z(), z(), z(), z()
        ops.bn_bwd_finalize(part, nparts, nparts * 4, 4, 0, 1, C, n, gam, mu, bn0.rstd, dg0, db0, bn0.k1, bn0.k2, bn0.k3)
        if res:
            ops.bn_bwd_finalize(part, nparts, nparts * 4, 4, 0, 2, C, n, rgam, mr, rbn0.rstd, rdg0, rdb0, rbn0.k1, rbn0.k2, rbn0.k3)
        for rep in range(3):
            dg, db, rdg, rdb = z(), z(), z(), z()
            for b in (bn, rbn):
                b.k1.zero_(), b.k2.zero_(), b.k3.zero_()
            tail = ops.make_bn_tail(dev, n, gam, bn, dg, db, *((rgam, rbn, rdg, rdb) if res else ()))
            red(tail)
            torch.cuda.synchronize()
            assert int(ops.bn_tail_tickets(dev).abs().sum().item()) == 0
            pairs = [(dg, dg0), (db, db0), (bn.k1, bn0.k1), (bn.k2, bn0.k2), (bn.k3, bn0.k3)]
            if res:
                pairs += [(rdg, rdg0), (rdb, rdb0), (rbn.k1, rbn0.k1), (rbn.k2, rbn0.k2), (rbn.k3, rbn0.k3)]
            for got, ref in pairs:
                assert rel_err(got.cpu(), ref.cpu().double()) < 1e-6, (cn8, rep)


@pytest.mark.parametrize("C,n,kind", [(64, 5000, 1), (128, 3000, 2), (20, 776, 0), (256, 9376, 2)])
def test_block_tail_relu_mask_is_bit_identical_f32(dev, C, n, kind):
    """sar_bn_add_relu_{fwd,bwd_reduce,bwd_apply}_mask_f32: one byte per float4 of y written by the forward tail, read by both
    backward passes instead of y -- every output bit for bit equal to the y-reading kernels, the mask equal to its definition."""
    from sar_amd import ops
    g = torch.Generator().manual_seed(3 * C + n)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    u, r, dy = rnd(C, n), rnd(C, n), rnd(C, n)
    sc, sh, rsc, rsh, mu, mr = 1 + 0.2 * rnd(C), 0.3 * rnd(C), 1 + 0.2 * rnd(C), 0.3 * rnd(C), 0.1 * rnd(C), 0.1 * rnd(C)
    k = [rnd(C) for _ in range(6)]
    y0, y1 = torch.empty_like(u), torch.empty_like(u)
    mask = ops.relu_mask(y1)
    assert mask is not None
    res = r if kind else None
    ops.bn_add_relu_fwd(u, sc, sh, kind, res, rsc if kind == 2 else None, rsh if kind == 2 else None, y0)
    ops.bn_add_relu_fwd(u, sc, sh, kind, res, rsc if kind == 2 else None, rsh if kind == 2 else None, y1, mask=mask)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    want = ((y0 > 0).view(C, n // 4, 4).to(torch.int32) * torch.tensor([1, 2, 4, 8], device=dev, dtype=torch.int32)).sum(dim=2)
    assert torch.equal(mask, want.to(torch.uint8))
    rr = r if kind == 2 else None
    p0, n0 = ops.bn_add_relu_bwd_reduce(dy, y0, u, rr, mu, mr if rr is not None else None)
    p1, n1 = ops.bn_add_relu_bwd_reduce(dy, None, u, rr, mu, mr if rr is not None else None, mask=mask)
    torch.cuda.synchronize()
    assert n0 == n1 and torch.equal(p0[:, :, :3], p1[:, :, :3])
    outs = []
    for m in (None, mask):
        du, dr, dz = torch.empty_like(u), (torch.empty_like(u) if rr is not None else None), torch.empty_like(u)
        ops.bn_add_relu_bwd_apply(dy, y0 if m is None else None, u, rr, k[:3], k[3:] if rr is not None else None, du, dr, dz, mask=m)
        outs.append((du, dr, dz))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert (a is None and b is None) or torch.equal(a, b)


def test_block_tail_forward_backward(dev):
    """y = relu(bn2(u) + bn_r(r)) and its backward (reductions, coefficients, apply)."""
    from sar_amd import ops
    g = torch.Generator().manual_seed(5)
    C, n = 64, 2 * 7 * 25
    u = torch.randn(C, n, generator=g).double().requires_grad_(True)
    r = torch.randn(C, n, generator=g).double().requires_grad_(True)
    g2 = (1 + 0.2 * torch.randn(C, generator=g)).double().requires_grad_(True)
    b2 = (0.2 * torch.randn(C, generator=g)).double().requires_grad_(True)
    gr = (1 + 0.2 * torch.randn(C, generator=g)).double().requires_grad_(True)
    br = (0.2 * torch.randn(C, generator=g)).double().requires_grad_(True)

    def bn(x, ga, be):
        m = x.mean(1, keepdim=True); v = x.var(1, unbiased=False, keepdim=True)
        return (x - m) * torch.rsqrt(v + 1e-3) * ga.view(-1, 1) + be.view(-1, 1), m.squeeze(1), torch.rsqrt(v + 1e-3).squeeze(1)

    zu, mu, ru = bn(u, g2, b2)
    zr, mr, rr = bn(r, gr, br)
    y = torch.relu(zu + zr)
    dy = torch.randn(C, n, generator=g)
    gu, grr, gg2, gb2, ggr, gbr = torch.autograd.grad(y, (u, r, g2, b2, gr, br), dy.double())
    d = lambda t: t.detach().float().to(dev).contiguous()
    sc2, sh2 = d(g2 * ru), d(b2 - mu * g2 * ru)
    scr, shr = d(gr * rr), d(br - mr * gr * rr)
    yd = torch.empty((C, n), device=dev)
    ops.bn_add_relu_fwd(d(u), sc2, sh2, 2, d(r), scr, shr, yd)
    torch.cuda.synchronize()
    assert rel_err(yd.cpu(), y) < TOL
    part, nparts = ops.bn_add_relu_bwd_reduce(d(dy), yd, d(u), d(r), d(mu), d(mr))
    z = lambda: torch.empty(C, device=dev)
    dg2, db2, k1, k2, k3 = z(), z(), z(), z(), z()
    dgr, dbr, q1, q2, q3 = z(), z(), z(), z(), z()
    ops.bn_bwd_finalize(part, nparts, nparts * 4, 4, 0, 1, C, n, d(g2), d(mu), d(ru), dg2, db2, k1, k2, k3)
    ops.bn_bwd_finalize(part, nparts, nparts * 4, 4, 0, 2, C, n, d(gr), d(mr), d(rr), dgr, dbr, q1, q2, q3)
    du, drr, dz = torch.empty((C, n), device=dev), torch.empty((C, n), device=dev), torch.empty((C, n), device=dev)
    ops.bn_add_relu_bwd_apply(d(dy), yd, d(u), d(r), (k1, k2, k3), (q1, q2, q3), du, drr, dz)
    torch.cuda.synchronize()
    assert rel_err(du.cpu(), gu) < TOL and rel_err(drr.cpu(), grr) < TOL
    assert rel_err(dg2.cpu(), gg2) < TOL and rel_err(db2.cpu(), gb2) < TOL
    assert rel_err(dgr.cpu(), ggr) < TOL and rel_err(dbr.cpu(), gbr) < TOL
    assert rel_err(dz.cpu(), dy.double() * (y > 0)) < 1e-7
    # identity residual, odd row length (scalar path)
    n2 = 3 * 25
    u2, x2 = torch.randn(C, n2, generator=g), torch.randn(C, n2, generator=g)
    y2 = torch.empty((C, n2), device=dev)
    ops.bn_add_relu_fwd(u2.to(dev), sc2, sh2, 1, x2.to(dev), None, None, y2)
    torch.cuda.synchronize()
    ref2 = torch.relu(u2.double() * sc2.cpu().double().view(-1, 1) + sh2.cpu().double().view(-1, 1) + x2.double())
    assert rel_err(y2.cpu(), ref2) < TOL


def test_data_bn_forward_backward(dev):
    from sar_amd import ops
    p = O.randomize_affine(O.init_params(5, blocks=[(64, 1, False)]))
    x, _ = O.synthetic_batch(3, seed=4, T=17, num_classes=5)
    new = {}
    xd = x.double().requires_grad_(False)
    pd = {k: v.double() for k, v in p.items()}
    gam = pd["data_bn.gamma"].clone().requires_grad_(True); bet = pd["data_bn.beta"].clone().requires_grad_(True)
    pd["data_bn.gamma"], pd["data_bn.beta"] = gam, bet
    ref = O.data_bn(xd, pd, True, new)                  # (N*M, C, T, V)
    N, C, T, V, M = x.shape
    part = torch.empty((V * C, N, 2), device=dev)
    xg = x.to(dev)
    ops.data_bn_stats(xg, None, part)
    z = lambda: torch.empty(V * C, device=dev)
    mean, rstd, scale, shift = z(), z(), z(), z()
    rm, rv = p["data_bn.moving_mean"].to(dev), p["data_bn.moving_var"].to(dev)
    ops.bn_finalize(part, N, V * C, N * M * T, 1e-3, 0.99, False, p["data_bn.gamma"].to(dev), p["data_bn.beta"].to(dev), rm, rv,
                    mean, rstd, scale, shift)
    out = torch.empty((C, N * M * T * V), device=dev)
    ops.data_bn_apply(xg, None, scale, shift, out)
    torch.cuda.synchronize()
    assert rel_err(from_cn(out.cpu(), N * M, T, V), ref) < TOL
    assert rel_err(rm.cpu(), new["data_bn.moving_mean"]) < TOL and rel_err(rv.cpu(), new["data_bn.moving_var"]) < TOL
    dy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1))
    gg, gb = torch.autograd.grad(ref, (gam, bet), dy.double())
    ops.data_bn_bwd_reduce(xg, None, to_cn(dy).to(dev), mean, part)
    dgam, dbet = z(), z()
    ops.bn_bwd_finalize(part, N, N * 2, 2, 0, 1, V * C, N * M * T, p["data_bn.gamma"].to(dev), mean, rstd, dgam, dbet)
    torch.cuda.synchronize()
    assert rel_err(dgam.cpu(), gg) < TOL and rel_err(dbet.cpu(), gb) < TOL


def test_bone_transform_is_bit_exact(dev, golden_dir):
    """data_gen/gen_bone_data.py:36-41 fused into the data_bn prologue: pure subtraction -> bit-exact."""
    import os
    from sar_amd import ops
    from sar_amd.bone import NTU_BONE_PAIRS, bone_parent_array
    x = torch.from_numpy(np.load(os.path.join(golden_dir, "ntu_clips_0_2.npy")))[:, :, :40].contiguous()
    N, C, T, V, M = x.shape
    bone = x.clone()
    for v1, v2 in NTU_BONE_PAIRS:
        bone[:, :, :, v1 - 1, :] = x[:, :, :, v1 - 1, :] - x[:, :, :, v2 - 1, :]
    one = torch.ones(V * C, device=dev); zero = torch.zeros(V * C, device=dev)
    out = torch.empty((C, N * M * T * V), device=dev)
    ops.data_bn_apply(x.to(dev), torch.from_numpy(bone_parent_array(V)).to(dev), one, zero, out)
    torch.cuda.synchronize()
    got = out.cpu().view(C, N, M, T, V).permute(1, 0, 3, 4, 2)
    assert torch.equal(got, bone)


def test_motion_stream_is_bit_exact(dev, golden_dir):
    """data_gen/gen_motion_data.py:24-27 (frame t+1 minus frame t, last frame 0) of the joint and of the bone data, fused
    into the data_bn prologue: float32 subtractions in the offline passes' order -> bit-exact; and a motion-stream train
    step of the engine on joints equals the oracle fed the motion tensor."""
    import os
    from oracle import stgcn as O
    from sar_amd import ops
    from sar_amd.bone import NTU_BONE_PAIRS, bone_parent_array
    from sar_amd.stgcn import STGCN
    x = torch.from_numpy(np.load(os.path.join(golden_dir, "ntu_clips_0_2.npy")))[:, :, :40].contiguous()
    N, C, T, V, M = x.shape
    bone = x.clone()
    for v1, v2 in NTU_BONE_PAIRS:
        bone[:, :, :, v1 - 1, :] = x[:, :, :, v1 - 1, :] - x[:, :, :, v2 - 1, :]
    one = torch.ones(V * C, device=dev); zero = torch.zeros(V * C, device=dev)
    for src, parent in ((x, None), (bone, torch.from_numpy(bone_parent_array(V)).to(dev))):
        motion = torch.zeros_like(src)
        motion[:, :, :T - 1] = src[:, :, 1:] - src[:, :, :-1]
        out = torch.empty((C, N * M * T * V), device=dev)
        ops.data_bn_apply(x.to(dev), parent, one, zero, out, motion=True)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu().view(C, N, M, T, V).permute(1, 0, 3, 4, 2), motion)
    blocks = [(64, 1, False), (64, 1, True)]
    p = O.randomize_affine(O.init_params(10, seed=3, dtype=torch.float64, blocks=blocks), seed=4)
    xs, ys = O.synthetic_batch(2, seed=3, T=16, num_classes=10)
    mo = torch.zeros_like(xs)
    mo[:, :, :15] = xs[:, :, 1:] - xs[:, :, :-1]
    eng = STGCN(num_classes=10, device=dev, blocks=blocks, motion=True)
    eng.load_params(p)
    logits, loss = eng.loss_and_grad(xs.to(dev), ys.to(dev))
    lref, loss_ref, gref, _, _ = O.loss_and_grads(p, mo.double(), ys, blocks=blocks)
    assert rel_err(logits.cpu(), lref) < 1e-4
    for k in ("data_bn.gamma", "data_bn.beta"):
        assert rel_err(eng.g[k].cpu(), gref[k]) < 1e-3, k


def test_head_loss_and_sgd(dev):
    from sar_amd import ops
    g = torch.Generator().manual_seed(9)
    N, Mp, C, TV, K = 5, 2, 256, 3 * 25, 60
    y = torch.randn(C, N * Mp * TV, generator=g).double().requires_grad_(True)
    W = (torch.randn(C, K, generator=g) * 0.1).double().requires_grad_(True)
    b = (torch.randn(K, generator=g) * 0.1).double().requires_grad_(True)
    labels = torch.randint(0, K, (N,), generator=g)
    pooled = y.view(C, N * Mp, TV).mean(2).t()
    feat = pooled.reshape(N, Mp, C).mean(1)
    logits = feat @ W + b
    loss = O.loss_fn(logits, labels, 8)
    gy, gW, gb = torch.autograd.grad(loss, (y, W, b))
    d = lambda t: t.detach().float().to(dev).contiguous()
    featd = torch.empty((N, C), device=dev); lg = torch.empty((N, K), device=dev)
    ops.pool_fwd(d(y), N * Mp, TV, Mp, featd)
    ops.fc_fwd(featd, d(W), d(b), lg)
    lossd = torch.empty(1, device=dev); dl = torch.empty((N, K), device=dev); pr = torch.empty((N, K), device=dev)
    ops.softmax_ce(lg, labels.to(dev), 1.0 / 8, lossd, dl, pr)
    dW, db, dfeat = torch.empty((C, K), device=dev), torch.empty(K, device=dev), torch.empty((N, C), device=dev)
    ops.fc_bwd(featd, d(W), dl, dW, db, dfeat)
    dy = torch.empty((C, N * Mp * TV), device=dev)
    ops.pool_bwd(dfeat, N * Mp, TV, Mp, dy)
    torch.cuda.synchronize()
    assert rel_err(lg.cpu(), logits) < TOL and rel_err(lossd.cpu(), loss.reshape(1)) < TOL
    assert rel_err(pr.cpu(), torch.softmax(logits, 1)) < TOL
    assert rel_err(dW.cpu(), gW) < TOL and rel_err(db.cpu(), gb) < TOL and rel_err(dy.cpu(), gy) < TOL
    # Nesterov SGD, two steps
    n = 1000
    w0, g0, g1 = torch.randn(n, generator=g), torch.randn(n, generator=g), torch.randn(n, generator=g)
    p = {"w": w0.clone().double()}; vel = {}
    wd, vd = w0.to(dev), torch.zeros(n, device=dev)
    lr = torch.tensor([0.1], device=dev)
    for gi in (g0, g1):
        O.sgd_nesterov_step(p, {"w": gi.double()}, vel, 0.1)
        ops.sgd_nesterov(wd, vd, gi.to(dev), lr, 0.9)
    torch.cuda.synchronize()
    assert rel_err(wd.cpu(), p["w"]) < 1e-6 and rel_err(vd.cpu(), vel["w"]) < 1e-6


def test_argument_errors_are_reported(dev):
    from sar_amd import ops, _lib as L
    x = torch.zeros((4, 50), device=dev)
    with pytest.raises(L.SarError):
        ops.conv_gemm(L.SAR_CONV_TEMPORAL, x, x.clone(), x, 0, 4, B=1, V=25, T_src=2, T_out=2, Kc=4, M=4, taps=5)  # unbuilt taps
    with pytest.raises(L.SarError):
        ops.bn_add_relu_fwd(x, x[0], x[0], 2, None, None, None, x.clone())   # missing residual operand
