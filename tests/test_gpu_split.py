"""GPU parity of the split arithmetics (csrc/conv_gemm_split.hip, conv_wgrad_split.hip) beyond the fp32 parity suites that run them
through tests/conftest.py's `arith_mode`: the RANGE behaviour of the fp16 arithmetic (f16x3a) -- operand bounds, wide dynamic
range, sums that exceed the source bound, scale invariance, saturation instead of infinities -- and the bound kernels themselves.
Reference: float64 torch on the CPU (oracle primitives), tolerance = the fp32 kernels' 2e-5 (norm-wise)."""
import numpy as np
import pytest
import torch

from oracle import stgcn as O
from util import to_cn, from_cn, rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5
ARITHS = ["f16x3a", "bf16x6"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from sar_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _bits(x):
    return np.float32(x).view(np.uint32).item()


def _cell(dev, value):
    return torch.tensor([_bits(value)], dtype=torch.int64).to(torch.int32).to(dev) if _bits(value) < 2 ** 31 else None


def test_amax_and_bound_kernels(dev):
    from sar_amd import ops
    g = torch.Generator().manual_seed(0)
    for shape, ld in (((7, 1001), 1001), ((64, 4000), 4096), ((3, 16), 16)):
        x = torch.randn(shape[0], ld, generator=g)[:, :shape[1]]
        x[shape[0] // 2, shape[1] // 3] = -37.5
        xd = x.to(dev)
        if ld != shape[1]:
            buf = torch.zeros(shape[0], ld, device=dev)
            buf[:, :shape[1]] = xd
            buf[:, shape[1]:] = 1e9            # beyond n: must not be read
            xd = buf[:, :shape[1]]
        cell = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.amax(xd, cell)
        assert cell.item() == _bits(37.5)
        ops.amax(xd * 0.5, cell)               # a cell is only ever RAISED
        assert cell.item() == _bits(37.5)
    gamma, beta = torch.randn(40, generator=g), torch.randn(40, generator=g)
    cell = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.bn_bound(gamma.to(dev), beta.to(dev), 960000, cell)
    want = (gamma.abs().double() * np.sqrt(960000 - 1) + beta.abs().double()).max().item()
    got = np.uint32(cell.item()).view(np.float32).item()
    assert want <= got <= want * 1.002
    src = torch.tensor([_bits(3.25)], dtype=torch.int32, device=dev)
    cell.zero_()
    ops.affine_bound(gamma.to(dev), beta.to(dev), src, cell)
    want = gamma.abs().max().item() * 3.25 + beta.abs().max().item()
    got = np.uint32(cell.item()).view(np.float32).item()
    assert want <= got <= want * 1.002


def test_samuelson_bound_holds_for_heavy_tails(dev):
    """|gamma (x - mean) rstd + beta| <= |gamma| sqrt(n - 1) + |beta| for ANY data normalised by its own statistics -- a single
    outlier in otherwise constant data reaches sqrt(n - 1) exactly"""
    n = 4096
    x = torch.zeros(n, dtype=torch.float64)
    x[17] = 1e6
    z = (x - x.mean()) / x.var(unbiased=False).sqrt()
    assert abs(z.abs().max().item() - np.sqrt(n - 1)) < 1e-6


def _temporal_case(dev, f, T, B, s, seed, src_scale=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, f, T, 25, generator=g)
    if src_scale is not None:
        x = x * src_scale
    kernel = torch.randn(9, 1, f, f, generator=g) * 0.05
    bias = torch.randn(f, generator=g) * 0.1
    return x, kernel, bias


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("f,T,s", [(64, 23, 1), (128, 20, 2)])
def test_temporal_forward_and_gradients_with_a_gradient_like_source(dev, arith, f, T, s):
    """a source with 13 binades between its channels and an outlier 2^11 above the largest of them (what du / dg look like; 25
    binades in all, inside the 2^29 window below the bound in which f16x3a keeps 22 bits per element -- DESIGN 3.10b states the
    limit): the data gradient (src = the wide operand) and the weight gradient (dout = the wide operand) against float64"""
    from sar_amd import ops, _lib as L
    B = 2
    g = torch.Generator().manual_seed(f + T)
    To, pad, _ = O.same_pad(T, 9, s)
    du = torch.randn(B, f, To, 25, generator=g) * torch.logspace(-4, 0, f).view(1, f, 1, 1) * 1e-4
    du[0, 3, 1, 7] = 0.3          # an outlier 1e3 .. 1e7 above everything else
    gx = torch.randn(B, f, T, 25, generator=g).double()
    sc = (1 + 0.2 * torch.randn(f, generator=g)).double(); sh = (0.3 * torch.randn(f, generator=g)).double()
    kernel = (torch.randn(9, 1, f, f, generator=g) * 0.05).double().requires_grad_(True)
    bias = torch.zeros(f, dtype=torch.float64, requires_grad=True)
    pre = (gx * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    y = O.temporal_conv(torch.relu(pre), kernel, bias, s)
    g_pre, g_k, g_b = torch.autograd.grad(y, (pre, kernel, bias), du.double())
    scd, shd = sc.float().to(dev), sh.float().to(dev)
    # data gradient, mask open everywhere (the ReLU mask is the fp32 suites' business): compare with the unmasked float64 gradient
    ones = torch.ones_like(pre)
    y2 = O.temporal_conv(pre * ones, kernel, bias, s)
    g_lin, = torch.autograd.grad(y2, pre, du.double())
    wT = torch.empty((9, f, f), device=dev)
    ops.transpose(kernel.detach().float().to(dev).contiguous(), wT, 9, f, f)
    dz = torch.empty((f, B * T * 25), device=dev)
    ops.conv_gemm(L.SAR_CONV_TEMPORAL, to_cn(du).to(dev), dz, wT, f * f, f, B=B, V=25, T_src=To, T_out=T, Kc=f, M=f, taps=9, stride=s,
                  pad=pad, transposed=True, split=arith)
    torch.cuda.synchronize()
    got = from_cn(dz.cpu(), B, T, 25)
    assert rel_err(got, g_lin) < TOL
    # per output CHANNEL too: W mixes the source channels, so every output row sees the large ones; but the rows of the outlier's
    # receptive field must not swamp the others
    far = torch.ones(T, dtype=torch.bool)
    far[max(0, 1 * s - 8):1 * s + 9] = False
    assert rel_err(got[0][:, far], g_lin[0][:, far]) < TOL
    if s == 1:
        flat = torch.zeros(9 * f * f + f, device=dev)
        ops.conv_wgrad(L.SAR_CONV_TEMPORAL, to_cn(gx.float()).to(dev), to_cn(du).to(dev), flat, B=B, V=25, T_src=T, T_out=To, Kc=f, M=f,
                       taps=9, stride=s, pad=pad, pro=(scd, shd), pro_relu=True, w_stride_tap=f * f, w_stride_c=f, wsize=9 * f * f,
                       bsize=f, split=arith)
        torch.cuda.synchronize()
        gk = flat[:9 * f * f].cpu().view(9, 1, f, f)
        assert rel_err(gk, g_k) < TOL
        # column m of dW sums dout row m only: the 1e-4-scaled rows must be as accurate (relative to themselves) as the large ones
        for m in (0, f // 2, f - 1):
            assert rel_err(gk[..., m], g_k[..., m]) < 5 * TOL, m
        assert rel_err(flat[9 * f * f:].cpu(), g_b) < TOL


def test_f16_results_scale_exactly_with_a_power_of_two(dev):
    """the operand scale is a power of two taken from the bound: src * 2^k gives the same term images, hence out * 2^k BIT FOR BIT"""
    from sar_amd import ops, _lib as L
    f, T, B = 64, 12, 2
    x, kernel, _ = _temporal_case(dev, f, T, B, 1, 5)
    outs = []
    for k in (0, -60, 50):
        out = torch.empty((f, B * T * 25), device=dev)
        ops.conv_gemm(L.SAR_CONV_TEMPORAL, to_cn(x * 2.0 ** k).to(dev), out, kernel.to(dev), f * f, f, B=B, V=25, T_src=T, T_out=T,
                      Kc=f, M=f, taps=9, stride=1, pad=4, split="f16x3a")
        torch.cuda.synchronize()
        outs.append(out.cpu().double() * 2.0 ** -k)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_a_stale_bound_saturates_instead_of_overflowing(dev):
    """a bound 2^20 too small (a caller's bug): values clamp at the fp16 maximum -- a wrong, FINITE result, never inf / nan"""
    from sar_amd import ops, _lib as L
    f, T, B = 64, 10, 1
    x, kernel, _ = _temporal_case(dev, f, T, B, 1, 6)
    img, wb = ops._pack_split_single(kernel.to(dev), f * f, f, 9, f, f, "f16x3a")
    small = torch.tensor([_bits(x.abs().max().item() * 2.0 ** -20)], dtype=torch.int32, device=dev)
    out = torch.empty((f, B * T * 25), device=dev)
    ops.conv_gemm(L.SAR_CONV_TEMPORAL, to_cn(x).to(dev), out, kernel.to(dev), f * f, f, B=B, V=25, T_src=T, T_out=T, Kc=f, M=f,
                  taps=9, stride=1, pad=4, split="f16x3a", packed=img, bounds=(small, wb))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), -float("inf")])
def test_a_non_finite_operand_reaches_the_outputs(dev, bad):
    """one NaN / Inf element in the source (forward, data gradient) or in dout (weight gradient) must come out of the
    fp16 split kernels as non-finite results, as it does out of the fp32 kernels and the reference's framework ops -- not as the
    finite values the +-65504 operand clamp and a NaN-dropping amax would launder it into.  The bound cells carry the BITS of the
    largest |value| under unsigned order (NaN > Inf > finite); a non-finite cell makes every output of the launch NaN."""
    from sar_amd import ops, _lib as L
    f, T, B = 64, 12, 2
    x, kernel, _ = _temporal_case(dev, f, T, B, 1, 7)
    x[1, 5, 3, 11] = bad
    for split in (None, "f16x3a"):
        out = torch.empty((f, B * T * 25), device=dev)
        ops.conv_gemm(L.SAR_CONV_TEMPORAL, to_cn(x).to(dev), out, kernel.to(dev), f * f, f, B=B, V=25, T_src=T, T_out=T, Kc=f, M=f,
                      taps=9, stride=1, pad=4, split=split)
        torch.cuda.synchronize()
        got = from_cn(out.cpu(), B, T, 25)
        # the fp32 kernel poisons the element's receptive field; the split kernel the whole launch: both are loud
        assert not torch.isfinite(got[1, :, 0:8, 11]).any(), split
    # weight gradient: a non-finite dout element
    dout, gx = torch.randn(B, f, T, 25), torch.randn(B, f, T, 25)
    dout[0, 2, 4, 4] = bad
    for split in (None, "f16x3a"):
        flat = torch.zeros(9 * f * f + f, device=dev)
        ops.conv_wgrad(L.SAR_CONV_TEMPORAL, to_cn(gx).to(dev), to_cn(dout).to(dev), flat, B=B, V=25, T_src=T, T_out=T, Kc=f, M=f,
                       taps=9, stride=1, pad=4, w_stride_tap=f * f, w_stride_c=f, wsize=9 * f * f, bsize=f, split=split)
        torch.cuda.synchronize()
        gk = flat[:9 * f * f].cpu().view(9, f, f)
        assert not torch.isfinite(gk[:, :, 2]).any(), split          # output channel 2 of every tap and source channel
        assert not torch.isfinite(flat[9 * f * f + 2].cpu()), split   # and its bias gradient
    # the by-product bounds of the element-wise passes keep it too (fmaxf would have dropped a NaN)
    u = torch.randn(8, 1000)
    u[3, 77] = bad
    cell = torch.zeros(1, dtype=torch.int32, device=dev)
    y = torch.empty(8, 1000, device=dev)
    mask = torch.empty(8, 250, dtype=torch.uint8, device=dev)
    ops.bn_add_relu_fwd(u.to(dev), torch.ones(8, device=dev), torch.zeros(8, device=dev), 0, None, None, None, y, mask=mask, amax_cell=cell)
    torch.cuda.synchronize()
    if bad != -float("inf"):          # relu(-inf) = 0 is a finite, correct value
        assert (cell.item() & 0xffffffff) >= 0x7f800000 and not torch.isfinite(y[3, 77].cpu())
    # a BatchNorm whose affine parameters diverged: the Samuelson bound is non-finite as well
    gamma = torch.ones(16)
    gamma[5] = bad
    cell.zero_()
    ops.bn_bound(gamma.to(dev), torch.zeros(16, device=dev), 1000, cell)
    assert (cell.item() & 0xffffffff) >= 0x7f800000


def test_a_diverged_step_is_visible_in_the_split_engine(dev):
    """engine level: one NaN weight (what a diverged optimizer step leaves behind) gives a NaN loss and NaN gradients in f32_split as
    in fp32 -- bench.py and the training loops assert on exactly that.  (A NaN in the INPUT is swallowed by the first block's
    folded BatchNorm + ReLU prologue, max(NaN, 0) = 0, in both arithmetics alike: not what this test is about.)"""
    from sar_amd.stgcn import STGCN
    blocks = [(64, 1, False), (64, 1, True), (128, 2, True)]
    p = O.init_params(10, seed=4, dtype=torch.float64, blocks=blocks)
    p["l1.tcn.kernel"][4, 0, 7, 9] = float("nan")
    x, y = O.synthetic_batch(2, seed=3, T=20, num_classes=10)
    for mode in ("fp32", "f32_split"):
        eng = STGCN(num_classes=10, device=dev, blocks=blocks, mfma=mode)
        eng.load_params(p)
        logits, loss = eng.loss_and_grad(x.to(dev), y.to(dev))
        torch.cuda.synchronize()
        assert not torch.isfinite(loss).item(), mode
        assert not torch.isfinite(eng.g["l0.gcn.kernel"]).all().item(), mode


def test_block_tail_backward_raises_the_bounds_of_du_and_dr(dev):
    """sar_bn_add_relu_bwd_apply_mask_amax_f32 raises amax_du AND (round 6) amax_dr to the largest magnitude it wrote -- what a
    separate sar_amax_f32 pass over each tensor gives -- and the outputs do not depend on the cells being asked for"""
    from sar_amd import ops
    g = torch.Generator().manual_seed(5)
    C, n = 24, 4000
    dy, u, r = (torch.randn(C, n, generator=g).to(dev) for _ in range(3))
    y = torch.randn(C, n, generator=g).to(dev)
    mask = ops.relu_mask(y)
    ops.bn_add_relu_fwd(u, torch.ones(C, device=dev), torch.zeros(C, device=dev), 0, None, None, None, y, mask=mask)
    k = [torch.randn(C, generator=g).to(dev) for _ in range(3)]
    rk = [torch.randn(C, generator=g).to(dev) * 3 for _ in range(3)]
    du, dr, dz = (torch.empty(C, n, device=dev) for _ in range(3))
    cells = torch.zeros(2, dtype=torch.int32, device=dev)
    ops.bn_add_relu_bwd_apply(dy, y, u, r, k, rk, du, dr, dz, mask=mask, amax_cell=cells[0:1], amax_dr_cell=cells[1:2])
    du2, dr2, dz2 = (torch.empty(C, n, device=dev) for _ in range(3))
    ops.bn_add_relu_bwd_apply(dy, y, u, r, k, rk, du2, dr2, dz2, mask=mask)
    torch.cuda.synchronize()
    assert torch.equal(du, du2) and torch.equal(dr, dr2) and torch.equal(dz, dz2)
    assert cells[0].item() == _bits(du.abs().max().item()) and cells[1].item() == _bits(dr.abs().max().item())
    assert cells[0].item() != cells[1].item()
    only_du = torch.zeros(1, dtype=torch.int32, device=dev)       # no residual branch: the second cell is not touched
    ops.bn_add_relu_bwd_apply(dy, y, u, None, k, None, du2, None, None, mask=mask, amax_cell=only_du, amax_dr_cell=cells[1:2])
    torch.cuda.synchronize()
    assert only_du.item() == _bits(du2.abs().max().item()) and cells[1].item() == _bits(dr.abs().max().item())


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("B,cin,f,T,s", [(2, 64, 128, 13, 2), (2, 128, 256, 10, 2), (3, 64, 128, 300, 2), (2, 256, 128, 75, 1),
                                         (1, 48, 72, 23, 2), (2, 16, 200, 31, 1), (1, 72, 40, 10, 2), (4, 128, 64, 150, 1)])
def test_one_tap_temporal_operator_on_the_split_kernel(dev, arith, B, cin, f, T, s):
    """conv_tap1_split_kernel (round 6): the strided 1x1 residual convolution forward with bias and BatchNorm partial sums, and the
    dense 1x1 product of its data gradient with the ADD / MASK epilogues -- ragged channel counts (not multiples of 32 / 64), tiles
    that end inside a sequence, T > one tile, gradient-like magnitudes; the launch must have taken the split kernel"""
    import torch.nn.functional as F
    from sar_amd import ops, _lib as L
    assert ops.split_applicable(L.SAR_CONV_TEMPORAL, 25, cin, f, 1, s, None, None, False, 0)
    g = torch.Generator().manual_seed(cin * 7 + f + T)
    x = torch.randn(B, cin, T, 25, generator=g)
    kernel = torch.randn(1, 1, cin, f, generator=g) * 0.1
    bias = torch.randn(f, generator=g) * 0.1
    ref = F.conv2d(x.double(), O.hwio_to_oihw(kernel.double()), bias.double(), stride=(s, 1))
    To = ref.shape[2]
    out = torch.empty((f, B * To * 25), device=dev)
    r = ops.conv_gemm(L.SAR_CONV_TEMPORAL, to_cn(x).to(dev), out, kernel.to(dev), 0, f, B=B, V=25, T_src=T, T_out=To, Kc=cin, M=f,
                      taps=1, stride=s, pad=0, bias=bias.to(dev), epi=L.SAR_EPI_STATS, split=arith)
    torch.cuda.synchronize()
    assert rel_err(from_cn(out.cpu(), B, To, 25), ref) < TOL
    part = r[0].cpu().double().sum(dim=1)
    assert rel_err(part[:, 0], ref.sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(part[:, 1], (ref * ref).sum(dim=(0, 2, 3))) < TOL
    # the dense data gradient (forward form with W^T) of a gradient-like dout, added to a skip gradient / masked by a source
    dr = torch.randn(B, f, To, 25, generator=g) * 3e-6
    dr[0, f // 2, To // 2, 7] = 1e-2                       # an outlier 3 000 x the typical magnitude
    # the weight / bias gradient (wgrad_tap1_split_kernel where T_src == stride T_out, else the fp32 kernel): dW[c][m] = sum x dr
    xs = x[:, :, ::s][:, :, :To].double()
    gk = torch.einsum("bcty,bfty->cf", xs, dr.double())
    gb = dr.double().sum(dim=(0, 2, 3))
    flat = torch.zeros(cin * f + f, device=dev)
    for nsplit in (None, 3):
        ops.conv_wgrad(L.SAR_CONV_TEMPORAL, to_cn(x).to(dev), to_cn(dr).to(dev), flat, B=B, V=25, T_src=T, T_out=To, Kc=cin, M=f, taps=1,
                       stride=s, pad=0, w_stride_tap=0, w_stride_c=f, wsize=cin * f, bsize=f, split=arith, nsplit=nsplit)
        torch.cuda.synchronize()
        assert rel_err(flat[:cin * f].cpu().view(cin, f), gk) < TOL, nsplit
        assert rel_err(flat[cin * f:].cpu(), gb) < TOL, nsplit
    skip = torch.randn(B, cin, To, 25, generator=g) * 1e-5
    want = torch.einsum("bfty,cf->bcty", dr.double(), kernel[0, 0].double())
    rT = kernel[0, 0].t().contiguous().to(dev)           # [f][cin]
    dx = torch.empty((cin, B * To * 25), device=dev)
    if cin % 8 == 0:
        ops.conv_gemm(L.SAR_CONV_TEMPORAL, to_cn(dr).to(dev), dx, rT, 0, cin, B=B, V=25, T_src=To, T_out=To, Kc=f, M=cin, taps=1,
                      stride=1, pad=0, epi=L.SAR_EPI_ADD, aux=to_cn(skip).to(dev), split=arith)
        torch.cuda.synchronize()
        assert rel_err(from_cn(dx.cpu(), B, To, 25), want + skip.double()) < TOL
        src = torch.randn(B, cin, To, 25, generator=g)
        sc, sh = 1 + 0.1 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g)
        pm = ops.conv_gemm(L.SAR_CONV_TEMPORAL, to_cn(dr).to(dev), dx, rT, 0, cin, B=B, V=25, T_src=To, T_out=To, Kc=f, M=cin, taps=1,
                           stride=1, pad=0, epi=L.SAR_EPI_MASK, aux=to_cn(src).to(dev), aux_affine=(sc.to(dev), sh.to(dev)), split=arith)
        torch.cuda.synchronize()
        keep = (src.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)) > 0
        assert rel_err(from_cn(dx.cpu(), B, To, 25), want * keep) < TOL
        assert rel_err(pm[0].cpu().double().sum(dim=1)[:, 0], (want * keep).sum(dim=(0, 2, 3))) < 1e-4


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("transpose", [False, True])
def test_graph_conv_with_sums_beyond_the_source_bound(dev, arith, transpose):
    """the transposed NTU lists hold sums of four joints with weight 1: the gathered value reaches 4x the source's amax (the first
    version of the kernel scaled from amax alone and saturated exactly there)"""
    from sar_amd import ops, _lib as L
    from oracle.graph import spatial_adjacency
    A = spatial_adjacency().astype(np.float32)
    tab = ops.GraphTables(A, dev, transpose)
    B, cin, f, T = 2, 64, 64, 7
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, cin, T, 25, generator=g) * 0.1
    x[:, :, :, [1, 2, 4, 8, 12, 16]] = 1.0            # every entry of the dense lists at the amax, same sign
    kernel = torch.randn(1, 1, cin, 3 * f, generator=g) * 0.1
    Aeff = torch.tensor(A).double()
    if transpose:
        Aeff = Aeff.transpose(1, 2)
    ref = O.graph_conv_td(x.double(), kernel.double(), torch.zeros(3 * f, dtype=torch.float64), Aeff)
    out = torch.empty((f, B * T * 25), device=dev)
    ops.conv_gemm(L.SAR_CONV_GRAPH, to_cn(x).to(dev), out, kernel.to(dev), f, 3 * f, B=B, V=25, T_src=T, T_out=T, Kc=cin, M=f, taps=3,
                  tables=tab, split=arith)
    torch.cuda.synchronize()
    assert rel_err(from_cn(out.cpu(), B, T, 25), ref) < TOL


@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("B,cin,f,T", [(7, 64, 64, 300), (3, 64, 128, 150), (5, 128, 128, 75), (2, 256, 256, 75), (1, 16, 64, 9), (130, 64, 64, 33)])
def test_persistent_graph_kernel_is_bit_identical_with_the_one_tile_kernel(dev, transpose, B, cin, f, T):
    """round 6: conv_graph_split2_kernel (persistent workgroups, raw source by LDS-DMA two stages ahead, virtual joints from the raw
    LDS tile) against conv_graph_split_kernel (one tile per workgroup, register staging) -- the same term images, products and
    epilogue, so EQUAL bits: outputs and partial sums, forward and transposed tables, the MASK epilogue here (the model suites run the
    gated one, and with SAR_GRAPH_SPLIT2=1 every launch), shapes with several tiles per workgroup (130 x 64 x 33 frames: 520 tiles on 512 workgroups),
    ragged last tiles (T = 75, 33, 9), two / four row blocks, row starts that are only 4-byte aligned (odd B x T = 75), Kc = 16."""
    from sar_amd import ops, _lib as L
    from oracle.graph import spatial_adjacency
    A = spatial_adjacency().astype(np.float32)
    tab = ops.GraphTables(A, dev, transpose)
    g = torch.Generator().manual_seed(B + cin + f + T)
    x = (torch.randn(B, cin, T, 25, generator=g) * torch.logspace(-2, 1, cin).view(1, cin, 1, 1)).to(dev)
    kernel = (torch.randn(1, 1, cin, 3 * f, generator=g) * 0.1).to(dev)
    bias = torch.randn(3 * f, generator=g).to(dev)
    # the persistent kernel takes the launches whose epilogue is not STATS (the data gradients of the engine: csrc/conv_gemm_split.hip
    # graph_split_v2; SAR_GRAPH_SPLIT2=1 in the environment of the process: every launch): the MASK epilogue carries both an output
    # tensor and partial sums, so it is the one compared here
    aux = torch.randn(f, B * T * 25, generator=g).to(dev)
    asc, ash, amu = (1 + 0.1 * torch.randn(f, generator=g)).to(dev), (0.2 * torch.randn(f, generator=g)).to(dev), (0.1 * torch.randn(f, generator=g)).to(dev)
    res = []
    for one_tile in (True, False):
        ops.GRAPH_ONE_TILE_WG = one_tile
        try:
            out = torch.full((f, B * T * 25), float("nan"), device=dev)
            r = ops.conv_gemm(L.SAR_CONV_GRAPH, to_cn(x.cpu()).to(dev), out, kernel, f, 3 * f, bias=bias, B=B, V=25, T_src=T, T_out=T, Kc=cin,
                              M=f, taps=3, tables=tab, epi=L.SAR_EPI_MASK, aux=aux, aux_affine=(asc, ash), aux_mean=amu, split="f16x3a")
            torch.cuda.synchronize()
        finally:
            ops.GRAPH_ONE_TILE_WG = False
        res.append((out.cpu(), r[0].cpu()))
    assert torch.isfinite(res[1][0]).all()
    assert torch.equal(res[0][0], res[1][0]), "outputs differ"
    assert torch.equal(res[0][1], res[1][1]), "BatchNorm-backward partial sums differ"
    keep = (aux.cpu().double() * asc.cpu().double().view(-1, 1) + ash.cpu().double().view(-1, 1)) > 0
    Aeff = torch.tensor(A).double().transpose(1, 2) if transpose else torch.tensor(A).double()
    ref = O.graph_conv_td(x.cpu().double(), kernel.cpu().double(), bias.cpu().double(), Aeff)
    assert rel_err(from_cn(res[1][0], B, T, 25), ref * from_cn(keep, B, T, 25)) < TOL


@pytest.mark.parametrize("arith", ARITHS)
def test_graph_weight_gradient_with_a_wide_range_dout(dev, arith):
    from sar_amd import ops, _lib as L
    from oracle.graph import spatial_adjacency
    A = spatial_adjacency().astype(np.float32)
    tab = ops.GraphTables(A, dev, False)
    B, cin, f, T = 2, 64, 128, 9
    g = torch.Generator().manual_seed(11)
    x = torch.relu(torch.randn(B, cin, T, 25, generator=g)).double().requires_grad_(True)
    kernel = (torch.randn(1, 1, cin, 3 * f, generator=g) * 0.1).double().requires_grad_(True)
    bias = torch.zeros(3 * f, dtype=torch.float64, requires_grad=True)
    dout = torch.randn(B, f, T, 25, generator=g) * torch.logspace(-6, 0, f).view(1, f, 1, 1) * 1e-5
    y = O.graph_conv_td(x, kernel, bias, torch.tensor(A).double())
    gk, gb = torch.autograd.grad(y, (kernel, bias), dout.double())
    flat = torch.zeros(cin * 3 * f + 3 * f, device=dev)
    ops.conv_wgrad(L.SAR_CONV_GRAPH, to_cn(x.detach().float()).to(dev), to_cn(dout).to(dev), flat, B=B, V=25, T_src=T, T_out=T, Kc=cin,
                   M=f, taps=3, tables=tab, w_stride_tap=f, w_stride_c=3 * f, wsize=cin * 3 * f, bsize=3 * f, split=arith)
    torch.cuda.synchronize()
    got = flat[:cin * 3 * f].cpu().view(1, 1, cin, 3 * f)
    assert rel_err(got, gk) < TOL
    for m in (0, f // 2, f - 1):          # output channel m of every slice: its own scale
        cols = [m, f + m, 2 * f + m]
        assert rel_err(got[..., cols], gk[..., cols]) < 5 * TOL, m
    assert rel_err(flat[cin * 3 * f:].cpu(), gb) < TOL


def test_pack_reports_each_items_amax(dev):
    from sar_amd import ops
    g = torch.Generator().manual_seed(3)
    flat = torch.randn(9 * 64 * 64 + 3 * 64 * 128, generator=g)
    flat[100] = 5.5
    flat[9 * 64 * 64 + 77] = -0.75 * 16
    pk = ops.PackedSplitWeights("f16x3a")
    pk.add("t", 0, 64 * 64, 64, 1, 9, 64, 64)
    pk.add("g", 9 * 64 * 64, 128, 3 * 128, 1, 3, 64, 128)
    pk.finalize(dev)
    pk.refresh(flat.to(dev))
    torch.cuda.synchronize()
    assert pk.bound("t").item() == _bits(flat[:9 * 64 * 64].abs().max().item())
    assert pk.bound("g").item() == _bits(flat[9 * 64 * 64:].abs().max().item())


@pytest.mark.parametrize("mode", ["f32_split", "f32_split_bf16x6"])
def test_split_engines_match_the_float64_oracle(dev, mode):
    """both split arithmetics as ENGINE modes (the fp32 parity suites run "f32_split" through tests/conftest.py; this is the x6 form's
    whole-step check beside it): logits, loss and every mask-conditioned gradient within the fp32 tolerance 1e-4"""
    from sar_amd.stgcn import STGCN
    from sar_amd import ops
    blocks = [(64, 1, False), (64, 1, True), (128, 2, True), (128, 1, True)]
    p = O.randomize_affine(O.init_params(11, seed=31, dtype=torch.float64, blocks=blocks), seed=32)
    x, y = O.synthetic_batch(2, seed=31, T=24, num_classes=11)
    eng = STGCN(num_classes=11, device=dev, blocks=blocks, mfma=mode)
    assert eng.split == {"f32_split": "f16x3a", "f32_split_bf16x6": "bf16x6"}[mode] and eng.spacked is not None
    eng.load_params(p)
    keep = {}
    eng.forward(x.to(dev), training=True, keep=keep)
    B, T = x.shape[0] * x.shape[4], x.shape[2]
    masks = {}
    for i, (f, s, _) in enumerate(blocks):
        To = -(-T // s)
        bn1 = eng.bn["l%d.bn1" % i]
        h = torch.empty_like(keep["l%d.g" % i])
        ops.bn_add_relu_fwd(keep["l%d.g" % i], bn1.scale, bn1.shift, 0, None, None, None, h)
        masks["l%d.h" % i] = from_cn((h > 0).cpu(), B, T, 25)
        masks["l%d.y" % i] = from_cn((keep["l%d.y" % i] > 0).cpu(), B, To, 25)
        T = To
    logits_ref, loss_ref, grads_ref, _, _ = O.loss_and_grads(p, x.double(), y, blocks=blocks, masks=masks)
    eng.load_params(p)
    logits, loss = eng.loss_and_grad(x.to(dev), y.to(dev))
    torch.cuda.synchronize()
    assert rel_err(logits.cpu(), logits_ref) < 1e-4 and rel_err(loss.cpu(), loss_ref.reshape(1)) < 1e-4
    for k, g in grads_ref.items():
        if g.abs().max().item() > 1e-9:
            assert rel_err(eng.g[k].cpu(), g) < 1e-4, k


def test_f32_split_trains_a_learnable_task_like_fp32(dev):
    """long-horizon behaviour (VERDICT r05 next #1d; the 1 200-step record is profiles/r06_f32split_training_curve.txt, written by
    tools/split_curve.py): 400 Nesterov-SGD steps of the full 10-block model on the learnable task of tests/test_gpu_bf16_training.py,
    fp32 and f32_split engines from the same weights on the same data stream -- and, as the YARDSTICK, the fp32 engine once more from
    weights ONE ulp away.  SGD at this rate separates ANY two runs after a few steps: at 400 steps the two fp32 runs end 0.02-0.07 /
    0.00-0.04 apart (loss / top-1 over the last 50 steps; 0.003 / 0.003 at 1 200 steps), differently on every change of a summation
    order anywhere in the step (a fixed 5 % band around ONE fp32 run failed on exactly that in round 6).  So the split engine must
    end inside the band the two fp32 runs span, widened by 1.5 x their distance and a small absolute margin -- far inside the bf16
    engine's 7.5 % / 0.03 when the fp32 runs agree, and no tighter than fp32 is with itself when they do not."""
    import test_gpu_bf16_training as TT
    from sar_amd.stgcn import STGCN
    classes, steps, bs = 10, 400, 32
    batch = TT._task(dev, classes)
    p = O.init_params(classes, seed=7, dtype=torch.float64)
    p_ulp = {k: (torch.nextafter(v.float(), torch.full_like(v.float(), float("inf"))).double() if v.is_floating_point() else v)
             for k, v in p.items()}
    res = {}
    for name, mode, params in (("fp32", "fp32", p), ("f32_split", "f32_split", p), ("fp32+1ulp", "fp32", p_ulp)):
        eng = STGCN(num_classes=classes, device=dev, mfma=mode)
        eng.load_params(params)
        losses, correct = [], []
        for s in range(steps):
            x, y = batch(bs, s)
            logits, loss = eng.loss_and_grad(x, y)
            eng.sgd_step(0.02 if s < 300 else 0.002)
            losses.append(loss.reshape(()))
            correct.append((logits.argmax(1) == y).float().mean())
        losses, correct = torch.stack(losses).cpu(), torch.stack(correct).cpu()
        assert torch.isfinite(losses).all(), name
        res[name] = (losses[0].item(), losses[-50:].mean().item(), correct[-50:].mean().item())
        del eng
    (f0, fl, fa), (s0, sl, sa), (c0, cl, ca) = res["fp32"], res["f32_split"], res["fp32+1ulp"]
    print("learnable task, %d steps: fp32 loss %.4f -> %.4f top-1 %.3f | f32_split %.4f -> %.4f top-1 %.3f | fp32 + 1 ulp %.4f -> %.4f top-1 %.3f"
          % (steps, f0, fl, fa, s0, sl, sa, c0, cl, ca))
    assert abs(s0 - f0) <= 1e-4 * abs(f0), "the first step's loss (same weights, same batch) is the parity tolerance's business"
    for l_, a_ in ((fl, fa), (cl, ca), (sl, sa)):
        assert l_ < 0.5 * f0 and a_ > 0.6 and l_ > 0.5, res            # every run learns the task down to its irreducible error
    assert abs(sl - 0.5 * (fl + cl)) <= 1.5 * abs(fl - cl) + 0.03 * fl, res
    assert abs(sa - 0.5 * (fa + ca)) <= 1.5 * abs(fa - ca) + 0.02, res


# ---------------------------------------------------------------------------------------------- Path B: csrc/conv2d_split.hip
@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 64, 64, 64, 2), (512, 512, 8, 8, 7), (128, 128, 32, 32, 1), (256, 256, 16, 16, 3),
                                            (24, 40, 20, 20, 2), (16, 8, 10, 12, 5), (8, 72, 7, 9, 9)])
def test_conv2d_3x3_forward_and_masked_data_gradient(dev, arith, cin, cout, H, W, B):
    """3x3 / stride 1 / pad 1 (models/resnet18.py:5-14) on the split kernels: forward behind a folded BatchNorm + ReLU with the
    BatchNorm sums, and the data gradient with the ReLU-mask epilogue and its centred sums, at the resnet's four resolutions
    (several images per tile, batches that do not fill the last tile, rectangular images) against float64 torch."""
    import torch.nn.functional as F
    from sar_amd import ops, _lib as L
    g = torch.Generator().manual_seed(cin + 3 * cout + H + B)
    cn = lambda t: t.permute(1, 0, 2, 3).reshape(t.shape[1], -1).contiguous()
    uncn = lambda y, b, h, w: y.reshape(y.shape[0], b, h, w).permute(1, 0, 2, 3)
    x = torch.randn(B, cin, H, W, generator=g).double()
    x[0, 0, 0, 0] = 300.0                                    # a source with outliers: the bound sits far above the bulk
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).double().requires_grad_(True)
    sc, sh = (1 + 0.2 * torch.randn(cin, generator=g)).double(), (0.3 * torch.randn(cin, generator=g)).double()
    hin = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    y = F.conv2d(hin, w, None, stride=1, padding=1)
    dy = torch.randn(y.shape, generator=g).double() * torch.exp(3 * torch.randn(y.shape, generator=g)).double()   # gradient-like
    gh, = torch.autograd.grad(y, hin, dy)
    geo = dict(B=B, Kc=cin, M=cout, H_src=H, W_src=W, H_out=H, W_out=W, KH=3, KW=3, stride=1, pad=1)
    assert ops.conv2d_split_applicable(**geo)
    wf = w.detach().float().permute(2, 3, 1, 0).reshape(-1).contiguous().to(dev)          # (tap, c, m)
    wb = w.detach().float().permute(2, 3, 0, 1).reshape(-1).contiguous().to(dev)          # (tap, m, c): the fp32 data gradient's operand
    xd = cn(x.float()).to(dev)
    out = torch.empty((cout, B * H * W), device=dev)
    r = ops.conv2d_gemm(xd, out, wf, cin * cout, cout, epi=L.SAR_EPI_STATS, pro=(sc.float().to(dev), sh.float().to(dev)),
                        pro_relu=True, split=arith, **geo)
    torch.cuda.synchronize()
    assert rel_err(uncn(out.cpu(), B, H, W), y.detach()) < TOL
    part = r[0].cpu().double().sum(1)
    assert (part[:, 0] - y.detach().sum(dim=(0, 2, 3))).abs().max() < TOL * y.detach().abs().sum(dim=(0, 2, 3)).max()
    assert rel_err(part[:, 1], (y.detach() ** 2).sum(dim=(0, 2, 3))) < TOL
    # data gradient with the ReLU mask of an upstream BatchNorm (aux = its pre-activation) and the centred second sum
    aux = torch.randn(cin, B * H * W, generator=g)
    asc, ash, amu = 1 + 0.1 * torch.randn(cin, generator=g), 0.2 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g)
    dx = torch.empty((cin, B * H * W), device=dev)
    rg = ops.conv2d_gemm(cn(dy.float()).to(dev), dx, wb, cout * cin, cin, epi=L.SAR_EPI_MASK, aux=aux.to(dev),
                         aux_affine=(asc.to(dev), ash.to(dev)), aux_mean=amu.to(dev), split=arith, B=B, Kc=cout, M=cin, H_src=H,
                         W_src=W, H_out=H, W_out=W, KH=3, KW=3, stride=1, pad=1, transposed=True)
    torch.cuda.synchronize()
    keep = (aux.double() * asc.double().view(-1, 1) + ash.double().view(-1, 1)) > 0
    want = cn(gh) * keep
    assert rel_err(dx.cpu(), want) < TOL
    pg = rg[0].cpu().double().sum(1)
    assert rel_err(pg[:, 1], (want * (aux.double() - amu.double().view(-1, 1))).sum(1)) < 5 * TOL


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 64, 64, 64, 2), (512, 512, 8, 8, 7), (128, 128, 32, 32, 1), (256, 256, 16, 16, 3),
                                            (24, 40, 12, 16, 2), (8, 72, 5, 8, 9), (40, 8, 3, 32, 3)])
def test_conv2d_3x3_weight_gradient(dev, arith, cin, cout, H, W, B):
    """weight gradient of the 3x3 / stride 1 / pad 1 convolution on the split kernels (the batch as one flat sequence: image borders by
    masked dout elements / the zero row) behind a folded BatchNorm + ReLU, with a gradient-like dout; against float64 torch."""
    import torch.nn.functional as F
    from sar_amd import ops
    g = torch.Generator().manual_seed(2 * cin + cout + H + W + B)
    cn = lambda t: t.permute(1, 0, 2, 3).reshape(t.shape[1], -1).contiguous()
    x = torch.randn(B, cin, H, W, generator=g).double()
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    sc, sh = (1 + 0.2 * torch.randn(cin, generator=g)).double(), (0.3 * torch.randn(cin, generator=g)).double()
    hin = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    y = F.conv2d(hin, w, None, stride=1, padding=1)
    dy = torch.randn(y.shape, generator=g).double() * torch.exp(2 * torch.randn(y.shape, generator=g)).double()
    gw, = torch.autograd.grad(y, w, dy)
    geo = dict(B=B, Kc=cin, M=cout, H_src=H, W_src=W, H_out=H, W_out=W, KH=3, KW=3, stride=1, pad=1)
    tmp = torch.empty(9 * cin * cout, device=dev)
    ops.conv2d_wgrad(cn(x.float()).to(dev), cn(dy.float()).to(dev), tmp, pro=(sc.float().to(dev), sh.float().to(dev)), pro_relu=True,
                     split=arith, **geo)
    torch.cuda.synchronize()
    got = tmp.cpu().view(3, 3, cin, cout).permute(3, 2, 0, 1)
    assert rel_err(got, gw) < TOL
