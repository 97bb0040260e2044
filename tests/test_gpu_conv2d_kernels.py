"""GPU parity of the 2-D convolution kernels (forward, data gradient, weight gradient) at every layer shape of the
reference's ResNet-18 (models/resnet18.py) against torch CPU float64 convolutions; max-pool stem tail; Adam."""
import pytest
import torch
import torch.nn.functional as F

from util import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5

# (cin, cout, k, stride, H)   -- square images; all conv shapes of resnet18(num_filters=64) on 256x256 inputs
SHAPES = [(1, 64, 7, 2, 256), (64, 64, 3, 1, 64), (64, 128, 3, 2, 64), (64, 128, 1, 2, 64), (128, 128, 3, 1, 32),
          (128, 256, 3, 2, 32), (128, 256, 1, 2, 32), (256, 256, 3, 1, 16), (256, 512, 3, 2, 16), (256, 512, 1, 2, 16),
          (512, 512, 3, 1, 8), (16, 32, 3, 2, 64), (8, 8, 3, 1, 16), (24, 40, 3, 1, 20)]


def cn(x):     # (B,C,H,W) -> [C][B*H*W]
    B, C, H, W = x.shape
    return x.permute(1, 0, 2, 3).reshape(C, B * H * W).contiguous()


def uncn(y, B, H, W):
    return y.reshape(y.shape[0], B, H, W).permute(1, 0, 2, 3)


@pytest.mark.parametrize("cin,cout,k,s,H", SHAPES)
def test_conv2d_forward_dgrad_wgrad(cin, cout, k, s, H):
    from sar_amd import ops, _lib as L
    dev = torch.device("cuda:0")
    B, pad, taps = 2, k // 2, k * k
    g = torch.Generator().manual_seed(cin * 7 + cout + k + s)
    x = torch.randn(B, cin, H, H, generator=g).double().requires_grad_(True)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * taps) ** 0.5).double().requires_grad_(True)
    sc = (1 + 0.2 * torch.randn(cin, generator=g)).double()
    sh = (0.3 * torch.randn(cin, generator=g)).double()
    use_pro = cin > 1
    pre = (x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if use_pro else x
    hin = torch.relu(pre) if use_pro else x
    y = F.conv2d(hin, w, None, stride=s, padding=pad)
    Ho = y.shape[2]
    dy = torch.randn(y.shape, generator=g)
    gh, gw = torch.autograd.grad(y, (hin if use_pro else x, w), dy.double())
    geo = dict(B=B, Kc=cin, M=cout, H_src=H, W_src=H, H_out=Ho, W_out=Ho, KH=k, KW=k, stride=s, pad=pad)
    wd = w.detach().float().to(dev).contiguous()
    wf = torch.empty(taps * cin * cout, device=dev)
    ops.permute3(wd, wf, taps, cin, cout, 1, taps, cin * taps)
    xd, dyd = cn(x.detach().float()).to(dev), cn(dy).to(dev)
    pro = (sc.float().to(dev), sh.float().to(dev)) if use_pro else None
    # forward (+ BN statistics)
    out = torch.empty((cout, B * Ho * Ho), device=dev)
    r = ops.conv2d_gemm(xd, out, wf, cin * cout, cout, epi=L.SAR_EPI_STATS, pro=pro, pro_relu=use_pro, **geo)
    torch.cuda.synchronize()
    assert rel_err(uncn(out.cpu(), B, Ho, Ho), y) < TOL
    part = r[0].cpu().double().sum(1)
    assert rel_err(part[:, 1], (y * y).sum(dim=(0, 2, 3))) < TOL
    # weight gradient
    tmp = torch.empty(taps * cin * cout, device=dev)
    ops.conv2d_wgrad(xd, dyd, tmp, pro=pro, pro_relu=use_pro, **geo)
    gwd = torch.empty((cout, cin, k, k), device=dev)
    ops.permute3(tmp, gwd, cout, cin, taps, 1, cout, cin * cout)
    torch.cuda.synchronize()
    assert rel_err(gwd.cpu(), gw) < TOL
    # data gradient (not needed for the 1-channel stem)
    if cin > 1:
        wb = torch.empty(taps * cin * cout, device=dev)
        ops.permute3(wd, wb, taps, cout, cin, 1, cin * taps, taps)
        dx = torch.empty((cin, B * H * H), device=dev)
        add = torch.randn(cin, B * H * H, generator=g)
        ops.conv2d_gemm(dyd, dx, wb, cout * cin, cin, epi=L.SAR_EPI_ADD, aux=add.to(dev), B=B, Kc=cout, M=cin, H_src=Ho,
                        W_src=Ho, H_out=H, W_out=H, KH=k, KW=k, stride=s, pad=pad, transposed=True)
        torch.cuda.synchronize()
        assert rel_err(uncn(dx.cpu() - add, B, H, H), gh) < TOL


@pytest.mark.parametrize("cin,cout,H", [(64, 128, 64), (256, 512, 16), (16, 32, 12), (8, 16, 15)])
def test_stride2_data_gradient_parity_classes(cin, cout, H):
    """3x3 / stride 2 / pad 1 data gradient: the parity-class launches (H even; H = 15 takes the masked kernel) with (a) the
    ReLU-mask epilogue + BatchNorm partial sums spread over the four launches, (b) SAR_C2D_AUX_EVEN_PIXELS: the compact data
    gradient of the parallel 1x1 / stride 2 convolution (models/resnet18.py:92-100) added at the even pixels."""
    from sar_amd import ops, _lib as L
    dev = torch.device("cuda:0")
    B, k, s, pad = 3, 3, 2, 1
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(B, cin, H, H, generator=g).double().requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).double()
    y = F.conv2d(x, w, None, stride=2, padding=1)
    Ho = y.shape[2]
    dy = torch.randn(y.shape, generator=g)
    gx, = torch.autograd.grad(y, x, dy.double())
    wb = torch.empty(9 * cin * cout, device=dev)
    ops.permute3(w.float().to(dev).contiguous(), wb, 9, cout, cin, 1, cin * 9, 9)
    dyd = cn(dy).to(dev)
    geo = dict(B=B, Kc=cout, M=cin, H_src=Ho, W_src=Ho, H_out=H, W_out=H, KH=3, KW=3, stride=2, pad=1, transposed=True)
    # (a) MASK: dz = dx where relu'(aux * sc + sh); partials = (sum dz, sum dz (aux - mean))
    aux = torch.randn(cin, B * H * H, generator=g)
    sc, sh, mu = 1 + 0.2 * torch.randn(cin, generator=g), 0.3 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g)
    dx = torch.empty((cin, B * H * H), device=dev)
    r = ops.conv2d_gemm(dyd, dx, wb, cout * cin, cin, epi=L.SAR_EPI_MASK, aux=aux.to(dev), aux_affine=(sc.to(dev), sh.to(dev)),
                        aux_mean=mu.to(dev), **geo)
    torch.cuda.synchronize()
    mask = (aux.double() * sc.double()[:, None] + sh.double()[:, None]) > 0
    ref = cn(gx) * mask
    assert rel_err(dx.cpu(), ref) < TOL
    part = r[0].cpu().double().sum(1)
    assert rel_err(part[:, 0], ref.sum(1)) < 1e-4 and rel_err(part[:, 1], (ref * (aux.double() - mu.double()[:, None])).sum(1)) < 1e-4
    # (b) compact aux at the even pixels
    if H % 2 == 0:
        small = torch.randn(cin, B * Ho * Ho, generator=g)
        dx2 = torch.empty((cin, B * H * H), device=dev)
        ops.conv2d_gemm(dyd, dx2, wb, cout * cin, cin, epi=L.SAR_EPI_ADD, aux=small.to(dev), aux_even_pixels=True, **geo)
        torch.cuda.synchronize()
        full = torch.zeros(B, cin, H, H, dtype=torch.float64)
        full[:, :, ::2, ::2] = uncn(small.double(), B, Ho, Ho)
        assert rel_err(uncn(dx2.cpu(), B, H, H), gx + full) < TOL
    else:
        with pytest.raises(L.SarError):
            ops.conv2d_gemm(dyd, dx, wb, cout * cin, cin, epi=L.SAR_EPI_ADD, aux=aux.to(dev), aux_even_pixels=True, **geo)


def test_stem_tail_maxpool_forward_backward():
    from sar_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    B, C, H = 2, 8, 30
    x = torch.randn(B, C, H, H, generator=g).double().requires_grad_(True)
    sc = (1 + 0.2 * torch.randn(C, generator=g)).double()
    sh = (0.3 * torch.randn(C, generator=g)).double()
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    y = F.max_pool2d(a, 3, 2, 1)
    Ho = y.shape[2]
    dy = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad(y, x, dy.double())
    gz = gx / sc.view(1, -1, 1, 1)                       # gradient w.r.t. the BN output
    xd = cn(x.detach().float()).to(dev)
    yd = torch.empty((C, B * Ho * Ho), device=dev)
    scd, shd = sc.float().to(dev), sh.float().to(dev)
    ops.bn_relu_maxpool_fwd(xd, scd, shd, yd, B, H, H)
    dz = torch.empty_like(xd)
    mean = x.detach().mean(dim=(0, 2, 3)).float().to(dev)
    part, nparts = ops.bn_relu_maxpool_bwd(xd, scd, shd, mean, cn(dy).to(dev), dz, B, H, H)
    torch.cuda.synchronize()
    assert rel_err(uncn(yd.cpu(), B, Ho, Ho), y) < TOL
    assert rel_err(uncn(dz.cpu(), B, H, H), gz) < TOL
    ps = part.cpu().double().sum(1)
    assert rel_err(ps[:, 0], gz.sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(ps[:, 1], (gz * (x.detach() - x.detach().mean(dim=(0, 2, 3), keepdim=True))).sum(dim=(0, 2, 3))) < 1e-4
