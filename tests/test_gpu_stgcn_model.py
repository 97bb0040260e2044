"""GPU parity: the whole ST-GCN train step (forward, loss, backward, SGD) on the HIP engine against
the CPU oracle.

Tolerance (norm-wise relative, max|a-b|/max|b|, reference value = the float64 oracle):
  * activations, logits, loss, BN moving statistics: 1e-4 (north_star: "fp32 logits/grads within 1e-4 rel");
  * gradients: 1e-4 on every tensor against the float64 oracle CONDITIONED ON THE
    ENGINE'S ACTIVATION PATTERN (every ReLU replaced by multiplication with the mask the HIP path used).  Why: a
    pre-activation that is zero to within rounding lands on either side in any two float32 implementations; one
    such flip moves that channel's heavily-cancelled sum(dz) -- hence that block's bn1.beta / gcn.kernel
    gradients -- by ~1/sqrt(positions) ~ 6e-3 (the float32 CPU oracle itself is up to 3e-3 from the float64
    truth on this stack for the same reason).  With the pattern fixed the comparison is exact again; the number
    of flipped elements is printed (a handful out of ~1e8)."""
import os

import numpy as np
import pytest
import torch

from oracle import stgcn as O
from util import rel_err, rel_err_fro, from_cn

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _engine_masks(eng, keep, blocks, B, T):
    """Activation pattern of the HIP path: block outputs y > 0; the ReLU folded into the temporal conv's operand
    load is re-evaluated with the engine's own arithmetic relu(fma(g, scale, shift))."""
    from sar_amd import ops
    masks = {}
    for i, (f, s, _) in enumerate(blocks):
        To = -(-T // s)
        bn1 = eng.bn["l%d.bn1" % i]
        hbuf = torch.empty_like(keep["l%d.g" % i])
        ops.bn_add_relu_fwd(keep["l%d.g" % i], bn1.scale, bn1.shift, 0, None, None, None, hbuf)
        masks["l%d.h" % i] = from_cn((hbuf > 0).cpu(), B, T, 25)
        masks["l%d.y" % i] = from_cn((keep["l%d.y" % i] > 0).cpu(), B, To, 25)
        T = To
    return masks


def offline_bone(x):
    """data_gen/gen_bone_data.py:36-41: the tensor the reference's offline pass writes for the bone stream."""
    from sar_amd.bone import NTU_BONE_PAIRS
    bone = x.clone()
    for v1, v2 in NTU_BONE_PAIRS:
        bone[:, :, :, v1 - 1, :] = x[:, :, :, v1 - 1, :] - x[:, :, :, v2 - 1, :]
    return bone


def _compare(dev, blocks, N, T, classes, seed, tol=TOL, x=None, y=None, stream="joint", band_check=True):
    """stream='bone': the ENGINE is fed joints and applies the bone transform in its data_bn prologue; the ORACLE is fed
    the offline bone tensor."""
    from sar_amd.stgcn import STGCN
    from sar_amd.bone import NTU_BONE_PAIRS
    p = O.randomize_affine(O.init_params(classes, seed=seed, dtype=torch.float64, blocks=blocks), seed=seed + 1)
    if x is None:
        x, y = O.synthetic_batch(N, seed=seed, T=T, num_classes=classes)
    x_engine = x
    if stream == "bone":
        x = offline_bone(x)
    logits_ref, loss_ref, grads_unc, new_stats, taps = O.loss_and_grads(p, x.double(), y, blocks=blocks)
    _, _, grads32, _, taps32 = O.loss_and_grads({k: v.float() for k, v in p.items()}, x.float(), y, blocks=blocks)
    band = {k: rel_err(grads32[k], g) for k, g in grads_unc.items() if g.abs().max().item() >= 1e-9}
    band_max = max(band.values())
    eng = STGCN(num_classes=classes, device=dev, blocks=blocks, bone_pairs=NTU_BONE_PAIRS if stream == "bone" else None)
    eng.load_params(p)
    keep = {}
    xg, yg = x_engine.to(dev), y.to(dev)
    logits = eng.forward(xg, training=True, keep=keep)
    torch.cuda.synchronize()
    B = x.shape[0] * x.shape[4]
    worst = {}
    Tc = x.shape[2]
    worst["x0"] = rel_err(from_cn(keep["x0"].cpu(), B, Tc, 25), taps["x0"])
    for i, (f, s, _) in enumerate(blocks):
        To = -(-Tc // s)
        worst["l%d.g" % i] = rel_err(from_cn(keep["l%d.g" % i].cpu(), B, Tc, 25), taps["l%d.g" % i])
        worst["l%d.u" % i] = rel_err(from_cn(keep["l%d.u" % i].cpu(), B, To, 25), taps["l%d.u" % i])
        worst["l%d.y" % i] = rel_err(from_cn(keep["l%d.y" % i].cpu(), B, To, 25), taps["l%d.y" % i])
        Tc = To
    worst["logits"] = rel_err(logits.cpu(), logits_ref)
    masks = _engine_masks(eng, keep, blocks, B, x.shape[2])
    flips = sum(int((masks["l%d.y" % i] != (taps["l%d.y" % i] > 0)).sum()) for i in range(len(blocks)))
    print("ReLU-tie flips vs the unconditioned oracle (block outputs): %d" % flips)
    _, _, grads_ref, _, _ = O.loss_and_grads(p, x.double(), y, blocks=blocks, masks=masks)
    logits2, loss = eng.loss_and_grad(xg, yg)
    torch.cuda.synchronize()
    worst["loss"] = rel_err(loss.cpu(), loss_ref.reshape(1))
    for k, gref in grads_ref.items():
        scale = gref.abs().max().item()
        if scale < 1e-9:      # conv biases in front of a BatchNorm: analytically zero gradient
            wk = grads_ref[k.replace(".bias", ".kernel")].abs().max().item()
            worst["grad " + k] = eng.g[k].abs().max().item() / max(wk, 1e-30)
        else:
            worst["grad " + k] = rel_err(eng.g[k].cpu(), gref)
    for k, v in new_stats.items():
        name = k.rsplit(".", 1)[0]
        got = eng.bn[name].moving_mean if k.endswith("moving_mean") else eng.bn[name].moving_var
        # two training forwards ran on the engine -> apply the momentum update twice for the reference
        m = 0.99
        batch = (v - m * p[k]) / (1 - m)
        second = m * v + (1 - m) * batch
        worst["stat " + k] = rel_err(got.cpu(), second)
    bad = {k: (v, tol) for k, v in worst.items() if not (v < tol)}
    # UNCONDITIONED checks (no knowledge of the engine's masks goes into the reference):
    # (1) the activation pattern itself: wherever the engine's ReLU decision differs from the plain float64 oracle's, the
    #     oracle's pre-activation must be a rounding-level tie (|z| <= 1e-4 of that tensor's largest pre-activation; measured
    #     ~1e-7).  A mask convention that was wrong consistently in forward and backward (wrong tensor, wrong affine, >= vs >)
    #     would pass the conditioned gradient test above and fail HERE on ordinary elements.
    # (2) every gradient tensor against the plain float64 oracle.  Two float32 evaluations cannot agree better than the
    #     float32 ORACLE agrees with the float64 one (each tie that lands on the other side moves one channel's cancelled
    #     sums by ~1/sqrt(positions)), so the yardstick is that oracle-vs-oracle band; which ties flip is luck of the
    #     rounding (max over ~80 tensors of a heavy-tailed per-tie perturbation: measured 1.0-3.0x the band on the 10-block
    #     stack, where the engine's folded BN affine flips 4-13 elements of ~1e8 and the float32 oracle 2-5), hence 4x.
    #     With (1) and the conditioned 1e-4 this bound is implied; it is asserted as the end-to-end envelope.
    worst_tie, n_flip = 0.0, 0
    for i in range(len(blocks)):
        for site in ("h", "y"):
            pre64 = taps["l%d.%s_pre" % (i, site)]
            diff = masks["l%d.%s" % (i, site)] != (pre64 > 0)
            if diff.any():
                n_flip += int(diff.sum())
                worst_tie = max(worst_tie, (pre64[diff].abs().max() / pre64.abs().max()).item())
    print("activation-pattern differences vs the float64 oracle: %d elements, largest |pre-activation| among them %.2e of "
          "its tensor's max" % (n_flip, worst_tie))
    assert worst_tie <= 1e-4, "the engine's ReLU pattern differs from the float64 oracle's away from ties (%.3e)" % worst_tie
    live = [k for k, g in grads_unc.items() if g.abs().max().item() >= 1e-9]
    unc = {k: rel_err(eng.g[k].cpu(), grads_unc[k]) for k in live}
    unc_fro = {k: rel_err_fro(eng.g[k].cpu(), grads_unc[k]) for k in live}
    band_fro = max(rel_err_fro(grads32[k], grads_unc[k]) for k in live)
    unc_max, unc_key = max((v, k) for k, v in unc.items())
    fro_max, fro_key = max((v, k) for k, v in unc_fro.items())
    flips32 = sum(int(((taps32["l%d.y" % i] > 0) != (taps["l%d.y" % i] > 0)).sum()) for i in range(len(blocks)))
    print("unconditioned gradient error vs float64 oracle: max-norm %.3e (%s) [float32-oracle band %.3e], Frobenius %.3e (%s) "
          "[band %.3e]; block-output ReLU flips vs float64: engine %d, float32 oracle %d"
          % (unc_max, unc_key, band_max, fro_max, fro_key, band_fro, flips, flips32))
    # (band_check=False: tensors of ~1e4 positions, where ONE rounding-level tie landing on the other side moves a channel's
    # sums by ~1e-2 while the float32 oracle often has no flip at all, i.e. a band of rounding size: the tie check (1) above and
    # the conditioned 1e-4 comparison carry those cases)
    assert not band_check or unc_max <= max(4 * band_max, tol), \
        "unconditioned max-norm error %.3e (%s) > 4x the float32-oracle band %.3e" % (unc_max, unc_key, band_max)
    print("float32-oracle gradient error band (unconditioned): max %.3e" % band_max)
    report = "\n".join("%-28s %.3e" % kv for kv in sorted(worst.items(), key=lambda kv: -kv[1])[:12])
    print(report)
    assert not bad, "parity failures (tol %g):\n%s\nworst:\n%s" % (tol, bad, report)
    return worst, eng, p


def test_two_blocks_small(dev):
    _compare(dev, [(64, 1, False), (64, 1, True)], N=2, T=12, classes=10, seed=0)


def test_stride2_conv_residual_blocks(dev):
    _compare(dev, [(64, 1, False), (128, 2, True), (128, 1, True), (256, 2, True)], N=2, T=22, classes=12, seed=1)


def test_odd_sizes_single_body(dev):
    """T not a multiple of the frame tile, one body (M=1), odd batch."""
    x, y = O.synthetic_batch(3, seed=7, T=17, M=1, num_classes=9)
    _compare(dev, [(64, 1, False), (64, 1, True), (128, 2, True)], N=3, T=17, classes=9, seed=2, x=x, y=y)


@pytest.mark.parametrize("N,T,M", [(1, 4, 1), (1, 3, 2), (5, 33, 3)])
def test_tiny_and_ragged_shapes(dev, N, T, M):
    """clips shorter than the temporal kernel (T = 3, 4 < 9: every tap but the centre ones reads TF-SAME padding), a single
    clip, three bodies, T not a multiple of any tile -- forward, loss and every gradient against the oracle"""
    x, y = O.synthetic_batch(N, seed=20 + T, T=T, M=M, num_classes=7)
    _compare(dev, [(64, 1, False), (64, 1, True), (128, 2, True)], N=N, T=T, classes=7, seed=21, x=x, y=y, band_check=False)


def test_full_model_ntu_shape(dev):
    """All 10 blocks, T=300, V=25, M=2, 60 classes (config 1/2 shape at a batch the oracle handles)."""
    _compare(dev, list(O.BLOCKS), N=2, T=300, classes=60, seed=3)


def test_full_model_on_reference_clips(dev, golden_dir):
    """The reference's bundled NTU clips (data/NTU_preprocessed_skeleton_examples.npy clips 0, 2):
    real data with an all-zero second body and zero-padded tail frames."""
    x = torch.from_numpy(np.load(os.path.join(golden_dir, "ntu_clips_0_2.npy")))
    y = torch.tensor([3, 41])
    _compare(dev, list(O.BLOCKS), N=2, T=300, classes=60, seed=4, x=x, y=y)


@pytest.mark.parametrize("stream", ["joint", "bone"])
def test_config5_ntu120_full_shape(dev, stream):
    """BASELINE.json configs[4] AT SHAPE: NTU-120 head (120 classes), all 10 blocks, T = 300, V = 25, M = 2, joint and bone
    streams (data_gen/gen_bone_data.py:36-41 fused into the data_bn prologue; the oracle gets the offline bone tensor)."""
    worst, eng, _ = _compare(dev, list(O.BLOCKS), N=2, T=300, classes=120, seed=12, stream=stream)
    assert eng.num_classes == 120 and eng.n_params == 3095502


def test_engine_level_outliers_through_all_ten_blocks(dev):
    """VERDICT r05 next #1d -- the RANGE behaviour of the arithmetic at engine level (tests/test_gpu_split.py has the kernel-level
    cases): all ten blocks at T = 300 with
      * one input coordinate at 1e4 x the clip's typical magnitude (after data_bn: one element at ~sqrt(n) sigma in a channel whose
        other elements shrink by the same factor -- BatchNorm caps what an outlier can be downstream, which is why the split
        kernels' Samuelson bound holds for ANY data),
      * one channel of a mid-stack block output at 1e4 x the others (bn2.gamma: the next block's graph convolution contracts a
        source whose bound sits 13 binades above its typical element -- the window f16x3a is built for is 29),
      * one logit gradient at 1e4 x the typical one (a gradient tensor whose second clip is 1e4 x the first through every block).
    Logits and every gradient against the float64 oracle (conditioned on the engine's ReLU pattern) at the fp32 tolerance, in both
    arithmetics (tests/conftest.py: arith_mode)."""
    from sar_amd.stgcn import STGCN
    blocks, N, T, classes = list(O.BLOCKS), 2, 300, 60
    p = O.randomize_affine(O.init_params(classes, seed=40, dtype=torch.float64, blocks=blocks), seed=41)
    p["l4.bn2.gamma"][17] = 1e4 * p["l4.bn2.gamma"].abs().mean()
    x, _ = O.synthetic_batch(N, seed=40, T=T, num_classes=classes)
    x[0, 1, 137, 9, 0] = 1e4 * x.abs().mean()
    g = torch.Generator().manual_seed(42)
    dlogits = torch.randn(N, classes, generator=g) * 1e-3
    dlogits[1, 17] = 1e4 * dlogits.abs().mean()
    eng = STGCN(num_classes=classes, device=dev, blocks=blocks)
    eng.load_params(p)
    keep = {}
    logits = eng.forward(x.to(dev), training=True, keep=keep)
    torch.cuda.synchronize()
    B = N * x.shape[4]
    masks = _engine_masks(eng, keep, blocks, B, T)
    names = O.trainable_names(p)
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    taps = {}
    logits_ref = O.forward(q, x.double(), True, {}, taps, blocks, masks)
    grads_ref = dict(zip(names, torch.autograd.grad(logits_ref, [leaves[k] for k in names], dlogits.double())))
    # the outliers are where they were meant to be
    y4 = taps["l4.y"].detach()
    assert y4[:, 17].abs().mean() > 1e3 * y4[:, [c for c in range(y4.shape[1]) if c != 17]].abs().mean()
    eng.backward(dlogits.to(dev))
    torch.cuda.synchronize()
    worst = {"logits": rel_err(logits.cpu(), logits_ref.detach())}
    for i in range(len(blocks)):
        To = taps["l%d.y" % i].shape[2]
        worst["l%d.y" % i] = rel_err(from_cn(keep["l%d.y" % i].cpu(), B, To, 25), taps["l%d.y" % i].detach())
    for k, gref in grads_ref.items():
        if gref.abs().max().item() >= 1e-9 * max(1.0, dlogits.abs().max().item()):
            worst["grad " + k] = rel_err(eng.g[k].cpu(), gref)
    report = "\n".join("%-28s %.3e" % kv for kv in sorted(worst.items(), key=lambda kv: -kv[1])[:10])
    print(report)
    bad = {k: v for k, v in worst.items() if not (v < TOL)}
    assert not bad, "parity failures with engine-level outliers (tol %g):\n%s" % (TOL, report)


def test_sgd_training_steps_track_the_oracle(dev):
    """Three full train steps (loss -> grads -> Nesterov SGD with the reference LR schedule)."""
    from sar_amd.stgcn import STGCN
    blocks = [(64, 1, False), (64, 1, True), (128, 2, True)]
    p = O.init_params(10, seed=5, dtype=torch.float64, blocks=blocks)
    eng = STGCN(num_classes=10, device=dev, blocks=blocks)
    eng.load_params(p)
    vel = {}
    for step in range(3):
        x, y = O.synthetic_batch(4, seed=10 + step, T=20, num_classes=10)
        # the oracle's gradients are conditioned on the engine's activation pattern (module docstring); the extra
        # forward that reads the pattern must not advance the moving statistics
        stats = {n: (bn.moving_mean.clone(), bn.moving_var.clone()) for n, bn in eng.bn.items()}
        keep = {}
        eng.forward(x.to(dev), training=True, keep=keep)
        masks = _engine_masks(eng, keep, blocks, x.shape[0] * x.shape[4], x.shape[2])
        for n, (mm, mv) in stats.items():
            eng.bn[n].moving_mean.copy_(mm)
            eng.bn[n].moving_var.copy_(mv)
        _, loss_ref, grads, new, _ = O.loss_and_grads(p, x.double(), y, blocks=blocks, masks=masks)
        lr = O.lr_schedule(step)
        O.sgd_nesterov_step(p, grads, vel, lr)
        p.update(new)
        _, loss = eng.loss_and_grad(x.to(dev), y.to(dev))
        eng.sgd_step(lr)
        torch.cuda.synchronize()
        assert rel_err(loss.cpu(), loss_ref.reshape(1)) < 1e-4
    sd = eng.state_dict()
    for k in O.trainable_names(p):
        if k.endswith("tcn.bias") or k.endswith("res.bias"):
            # a bias in front of a train-mode BatchNorm has an analytically zero gradient: only rounding noise moves it
            assert sd[k].abs().max().item() < 1e-6, k
        else:
            assert rel_err(sd[k], p[k]) < 2e-4, k


def test_inference_mode_uses_moving_statistics(dev):
    from sar_amd.stgcn import STGCN
    blocks = [(64, 1, False), (64, 1, True), (128, 2, True)]
    p = O.randomize_affine(O.init_params(10, seed=6, dtype=torch.float64, blocks=blocks))
    x, _ = O.synthetic_batch(3, seed=3, T=16, num_classes=10)
    ref = torch.softmax(O.forward(p, x.double(), False, blocks=blocks), 1)
    eng = STGCN(num_classes=10, device=dev, blocks=blocks)
    eng.load_params(p)
    probs = eng.predict(x.to(dev))
    torch.cuda.synchronize()
    assert rel_err(probs.cpu(), ref) < 1e-4


def test_bone_stream_120_classes(dev):
    """Config 5: the bone stream (data_gen/gen_bone_data.py:36-41 fused into the data_bn prologue) with the NTU-120
    head: a full train step of the engine fed JOINTS with bone_pairs must equal the oracle fed the BONE tensor the
    reference's offline pass would have written."""
    from sar_amd.stgcn import STGCN
    from sar_amd.bone import NTU_BONE_PAIRS
    blocks = [(64, 1, False), (64, 1, True), (128, 2, True), (256, 2, True)]
    p = O.randomize_affine(O.init_params(120, seed=9, dtype=torch.float64, blocks=blocks), seed=10)
    x, y = O.synthetic_batch(3, seed=9, T=20, num_classes=120)
    bone = x.clone()
    for v1, v2 in NTU_BONE_PAIRS:
        bone[:, :, :, v1 - 1, :] = x[:, :, :, v1 - 1, :] - x[:, :, :, v2 - 1, :]
    eng = STGCN(num_classes=120, device=dev, blocks=blocks, bone_pairs=NTU_BONE_PAIRS)
    eng.load_params(p)
    keep = {}
    logits = eng.forward(x.to(dev), training=True, keep=keep)
    masks = _engine_masks(eng, keep, blocks, x.shape[0] * x.shape[4], x.shape[2])
    logits_ref, loss_ref, grads_ref, _, _ = O.loss_and_grads(p, bone.double(), y, blocks=blocks, masks=masks)
    eng.load_params(p)
    logits, loss = eng.loss_and_grad(x.to(dev), y.to(dev))
    torch.cuda.synchronize()
    assert logits.shape == (3, 120)
    assert rel_err(logits.cpu(), logits_ref) < TOL and rel_err(loss.cpu(), loss_ref.reshape(1)) < TOL
    for k, g in grads_ref.items():
        if g.abs().max().item() >= 1e-9:
            assert rel_err(eng.g[k].cpu(), g) < TOL, k


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_train_step_is_bitwise_deterministic(dev, mode):
    """Full NTU shape (T = 300, 10 blocks): no atomics, every reduction in a fixed order => logits, loss and the whole
    gradient buffer repeat bit for bit; a mismatch would mean a race in a kernel (LDS hazard, missing barrier).
    tools/determinism_check.py runs the same check at the bench batch size."""
    from sar_amd.stgcn import STGCN
    from sar_amd.train import synthetic_clips
    x, y = synthetic_clips(6, dev, seed=3, num_classes=60)
    eng = STGCN(num_classes=60, device=dev, seed=0, mfma=mode)
    state = {k: v.clone() for k, v in eng.state_dict().items()}
    ref = None
    for _ in range(3):
        eng.load_params(state)
        logits, loss = eng.loss_and_grad(x, y)
        torch.cuda.synchronize()
        cur = (logits.clone(), loss.clone(), eng.grad.clone())
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(p, q) for p, q in zip(ref, cur))
    assert torch.isfinite(ref[2]).all()
