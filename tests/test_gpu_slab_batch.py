"""One slab reduction per gradient bucket (sar_slab_reduce_batch_f32, ops.SlabBatch; round 6): the weight-gradient slabs of a
backward pass are summed by ONE launch per flush instead of one launch behind every weight-gradient kernel.  The batch kernel
performs, per output element, the additions of sar_slab_reduce_f32 in the same order, so

  * kernel level: every item of a ragged batch (1 .. 513 slabs, 1 .. 150 000 outputs, strided slabs) equals the single launch BIT
    FOR BIT, outputs beyond an item's n are not touched;
  * engine level: the gradients of a train step with the batch are bit-identical with the per-launch schedule -- ST-GCN in the three
    arithmetics (fp32, f32_split, bf16 storage) and the resnet of Path B (fp32, f32_split), also when the step hands its buckets
    to a data-parallel callback (a flush per bucket), and across repeated steps (the slab buffers and the device table are re-used).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from sar_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def test_batched_slab_reduce_equals_the_single_launches_bit_for_bit(dev):
    from sar_amd import _lib as L
    from sar_amd import ops
    lib = L.load()
    g = torch.Generator(device=dev).manual_seed(7)
    shapes = [(1, 1, 0), (3, 63, 0), (4, 64, 5), (17, 65, 0), (16, 36928, 0), (512, 4160, 0), (513, 1000, 24), (33, 150000, 0)]
    batch = ops.SlabBatch()
    outs, refs = [], []
    for nsplit, n, extra in shapes:
        slab = torch.randn((nsplit, n + extra), generator=g, device=dev) * (10.0 ** (nsplit % 5 - 2))
        out = torch.full((n + 8,), -7.0, device=dev)
        ref = torch.full((n + 8,), -7.0, device=dev)
        ops.check(lib.sar_slab_reduce_f32(ops.ptr(slab), nsplit, slab.stride(0), n, ops.ptr(ref), ops.stream_ptr()), "single")
        batch.add(slab, nsplit, n, out)
        outs.append((out, slab))
        refs.append(ref)
    batch.flush()
    batch.flush()            # nothing pending: no launch
    torch.cuda.synchronize()
    for (out, _), ref, (nsplit, n, _) in zip(outs, refs, shapes):
        assert torch.equal(out, ref), (nsplit, n)
        assert (out[n:] == -7.0).all()
    # against float64 (the sums themselves)
    out, slab = outs[5]
    want = slab[:, :4160].double().sum(0)
    assert (out[:4160].double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()


def _stgcn_grads(dev, mfma, batch, with_buckets, steps=2):
    from sar_amd import ops
    from sar_amd.stgcn import STGCN
    from sar_amd.train import synthetic_clips
    blocks = [(64, 1, False), (64, 1, True), (128, 2, True), (128, 1, True)]
    old = ops.SLAB_BATCH
    ops.SLAB_BATCH = batch
    try:
        eng = STGCN(num_classes=10, device=dev, blocks=blocks, mfma=mfma, seed=3)
    finally:
        ops.SLAB_BATCH = old
    assert (eng._slabs is not None) == batch
    x, y = synthetic_clips(4, dev, seed=11, num_classes=10, T=40)
    seen = []

    def cb(bi, flat, events):
        for ev in events:
            torch.cuda.current_stream().wait_event(ev)
        seen.append((bi, flat.clone()))
    out = []
    for _ in range(steps):
        eng.loss_and_grad(x, y, bucket_cb=cb if with_buckets else None)
        out.append(eng.grad.clone())
        eng.sgd_step(0.05)
    torch.cuda.synchronize()
    return out, seen


@pytest.mark.parametrize("mfma", ["fp32", "f32_split", "bf16"])
@pytest.mark.parametrize("with_buckets", [False, True])
def test_stgcn_gradients_are_bit_identical_with_one_slab_reduction_per_bucket(dev, mfma, with_buckets):
    a, seen_a = _stgcn_grads(dev, mfma, True, with_buckets)
    b, seen_b = _stgcn_grads(dev, mfma, False, with_buckets)
    for ga, gb in zip(a, b):
        assert torch.isfinite(ga).all() and ga.abs().max().item() > 0
        assert torch.equal(ga, gb)
    assert len(seen_a) == len(seen_b)
    for (bi, fa), (bj, fb) in zip(seen_a, seen_b):      # what the all-reduce of a bucket would have read
        assert bi == bj and torch.equal(fa, fb)
    if with_buckets:
        assert len(seen_a) >= 2 * 2


@pytest.mark.parametrize("mfma", ["fp32", "f32_split"])
def test_resnet_gradients_are_bit_identical_with_one_slab_reduction_per_bucket(dev, mfma):
    from sar_amd import ops
    from sar_amd.resnet import ResNet18

    from sar_amd import resnet

    def grads(batch, with_buckets):
        old = (ops.SLAB_BATCH, resnet.SLAB_BATCH_PATHB)       # (the resnet keeps its per-gradient reductions by default: measured)
        ops.SLAB_BATCH = resnet.SLAB_BATCH_PATHB = batch
        try:
            net = ResNet18(num_classes=10, device=dev, seed=5, mfma=mfma)
        finally:
            ops.SLAB_BATCH, resnet.SLAB_BATCH_PATHB = old
        assert (net._slabs is not None) == batch
        g = torch.Generator(device=dev).manual_seed(2)
        x = torch.randn((4, 1, 64, 64), generator=g, device=dev)
        y = torch.tensor([1, 3, 5, 7], device=dev)
        seen = []

        def cb(bi, flat, events):
            for ev in events:
                torch.cuda.current_stream().wait_event(ev)
            seen.append(flat.clone())
        out = []
        for _ in range(2):
            net.loss_and_grad(x, y, bucket_cb=cb if with_buckets else None)
            out.append(net.grad.clone())
        torch.cuda.synchronize()
        return out, seen
    for with_buckets in (False, True):
        a, sa = grads(True, with_buckets)
        b, sb = grads(False, with_buckets)
        for ga, gb in zip(a, b):
            assert torch.isfinite(ga).all() and ga.abs().max().item() > 0
            assert torch.equal(ga, gb)
        assert len(sa) == len(sb) and all(torch.equal(u, v) for u, v in zip(sa, sb))


@pytest.mark.parametrize("mfma", ["fp32", "f32_split"])
def test_resnet_with_the_down_sampling_branch_on_its_own_stream_is_bit_identical(dev, mfma):
    """SAR_PATHB_DS_STREAM (sar_amd/resnet.py: DS_STREAM): the 1x1 down-sampling convolution + its BatchNorm finalisation (forward) and the
    dense 1x1 product of its data gradient (backward) forked onto a third stream -- same launches, same order per tensor: logits,
    loss and every gradient bit for bit, over repeated steps (moving statistics included)"""
    from sar_amd import resnet
    from sar_amd.resnet import ResNet18

    def run(forked):
        old = resnet.DS_STREAM
        resnet.DS_STREAM = forked
        try:
            net = ResNet18(num_classes=10, device=dev, seed=5, mfma=mfma)
        finally:
            resnet.DS_STREAM = old
        assert (net._aux is not None) == forked
        g = torch.Generator(device=dev).manual_seed(2)
        x = torch.randn((4, 1, 64, 64), generator=g, device=dev)
        y = torch.tensor([1, 3, 5, 7], device=dev)
        out = []
        for _ in range(3):
            logits, loss = net.loss_and_grad(x, y)[:2]
            out.append((logits.clone(), loss.clone(), net.grad.clone()))
        torch.cuda.synchronize()
        stats = torch.cat([b.moving_mean for b in net.bn.values()] + [b.moving_var for b in net.bn.values()])
        return out, stats
    a, sa = run(True)
    b, sb = run(False)
    for (la, lsa, ga), (lb, lsb, gb) in zip(a, b):
        assert torch.equal(la, lb) and torch.equal(lsa, lsb) and torch.equal(ga, gb)
        assert torch.isfinite(ga).all() and ga.abs().max().item() > 0
    assert torch.equal(sa, sb)


@pytest.mark.parametrize("mfma", ["f32_split", "f32_split_bf16x6"])
def test_stgcn_with_its_weight_images_on_the_third_stream_is_bit_identical(dev, mfma):
    """sar_amd/stgcn.py: AUX_STREAM -- the split engine's term images, the zeroing of its bound cells and its Samuelson cells issued on a
    third stream beside the data_bn stage and the first layer, joined in front of their first reader: logits, losses and gradients
    bit for bit with the serial schedule over repeated steps.  (The bf16 engine's one pack launch measured SLOWER forked: not forked.)"""
    from sar_amd import stgcn
    from sar_amd.stgcn import STGCN
    from sar_amd.train import synthetic_clips
    blocks = [(64, 1, False), (64, 1, True), (128, 2, True), (128, 1, True)]

    def run(forked):
        old = stgcn.AUX_STREAM
        stgcn.AUX_STREAM = forked
        try:
            eng = STGCN(num_classes=10, device=dev, blocks=blocks, mfma=mfma, seed=3)
        finally:
            stgcn.AUX_STREAM = old
        assert (eng._aux is not None) == forked
        out = []
        for i in range(3):
            x, y = synthetic_clips(4, dev, seed=20 + i, num_classes=10, T=40)
            logits, loss = eng.loss_and_grad(x, y)
            out.append((logits.clone(), loss.clone(), eng.grad.clone()))
            eng.sgd_step(0.05)
        torch.cuda.synchronize()
        return out
    for (la, lsa, ga), (lb, lsb, gb) in zip(run(True), run(False)):
        assert torch.equal(la, lb) and torch.equal(lsa, lsb) and torch.equal(ga, gb)
        assert torch.isfinite(ga).all() and ga.abs().max().item() > 0
