"""GPU parity of the spectrogram path's classifier: the HIP ResNet-18 engine (through the C ABI) against
 (1) the golden outputs of the REFERENCE's own models/resnet18.py (tests/golden/resnet18_tiny.npz: 8 filters, 64x64),
 (2) the CPU oracle at the reference's real shape (64 filters, 256x256 images),
and the drop-in models.resnet.Model / main_spectrogram.py."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import resnet as RN
from util import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_tiny_resnet_matches_reference_golden(dev, golden_dir):
    from sar_amd.resnet import ResNet18
    gold = np.load(os.path.join(golden_dir, "resnet18_tiny.npz"))
    p = {k[6:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("param:")}
    eng = ResNet18(num_classes=7, num_filters=8, device=dev)
    eng.load_params(p)
    x, y = torch.from_numpy(gold["x"]).to(dev), torch.from_numpy(gold["y"]).to(dev)
    logits, loss = eng.loss_and_grad(x, y)
    torch.cuda.synchronize()
    assert rel_err(logits.cpu(), torch.from_numpy(gold["logits_train"])) < 1e-4
    assert abs(loss.item() - float(gold["loss"])) < 1e-4 * abs(float(gold["loss"]))
    worst = {}
    for k in eng.shapes:
        worst[k] = rel_err(eng.g[k].cpu(), torch.from_numpy(gold["grad:" + k]))
    bad = {k: v for k, v in worst.items() if not v < 1e-3}
    print(sorted(worst.items(), key=lambda kv: -kv[1])[:6])
    assert not bad, bad
    for name, b in eng.bn.items():
        assert rel_err(b.moving_mean.cpu(), torch.from_numpy(gold["after:" + name + ".running_mean"])) < 1e-4
        assert rel_err(b.moving_var.cpu(), torch.from_numpy(gold["after:" + name + ".running_var"])) < 1e-4
    out = eng.forward(x, training=False)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), torch.from_numpy(gold["logits_eval"])) < 1e-4


def _engine_masks(eng, keep, B):
    """The engine's activation pattern at every ReLU site, in the oracle's NCHW layout.  Block outputs: y > 0.  The
    ReLUs folded into the conv operand load (stem bn1, every block's bn1) are re-evaluated with the engine's own
    kernel arithmetic relu(fma(x, scale, shift)) through sar_bn_add_relu_fwd_f32."""
    from sar_amd import ops
    masks = {}
    def nchw(t):
        n = t.shape[1] // B
        h = int(round(n ** 0.5))
        return (t > 0).cpu().view(t.shape[0], B, h, h).permute(1, 0, 2, 3)
    def folded(x, bn):
        out = torch.empty_like(x)
        ops.bn_add_relu_fwd(x, bn.scale, bn.shift, 0, None, None, None, out)
        return nchw(out)
    masks["bn1"] = folded(keep["conv1"], eng.bn["bn1"])
    for pre, _, _, _, _ in eng.blocks:
        masks[pre + "bn1"] = folded(keep[pre + "c1"], eng.bn[pre + "bn1"])
        masks[pre + "out"] = nchw(keep[pre + "out"])
    return masks


def test_full_width_resnet_matches_oracle(dev):
    """64 filters, 256x256 spectrogram images (config 4 shape), batch 2: activations, logits, loss vs the fp64 oracle;
    gradients vs the fp64 oracle CONDITIONED ON THE ENGINE'S ACTIVATION PATTERN (ReLU = multiply by the engine's
    mask): a pre-activation that is zero to within rounding may land on either side in any two float32
    implementations, and one such flip moves a heavily-cancelled BatchNorm-backward sum by ~1e-2; with the pattern
    fixed every gradient must agree to 1e-4."""
    from sar_amd.resnet import ResNet18
    eng = ResNet18(num_classes=60, num_filters=64, device=dev, seed=3)
    p = {k: v.double() for k, v in eng.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 1, 256, 256, generator=g) * 3 - 4
    y = torch.tensor([7, 33])
    lref, loss_ref, _, stats, taps = RN.loss_and_grads(p, x.double(), y)
    keep = {}
    logits = eng.forward(x.to(dev), training=True, keep=keep)
    torch.cuda.synchronize()
    B = 2
    def nchw(t, Cc, H, W):
        return t.cpu().view(Cc, B, H, W).permute(1, 0, 2, 3)
    assert rel_err(nchw(keep["conv1"], 64, 128, 128), taps["conv1"]) < 1e-4
    assert rel_err(nchw(keep["pool"], 64, 64, 64), taps["pool"]) < 1e-4
    assert rel_err(nchw(keep["layer4.1.out"], 512, 8, 8), taps["layer4.1.out"]) < 1e-4
    assert rel_err(logits.cpu(), lref) < 1e-4
    masks = _engine_masks(eng, keep, B)
    flips = sum(int((masks[pre + "out"] != (taps[pre + "out"] > 0)).sum()) for pre, _, _, _, _ in eng.blocks)
    _, _, gref, _, _ = RN.loss_and_grads(p, x.double(), y, masks=masks)
    logits, loss = eng.loss_and_grad(x.to(dev), y.to(dev))
    torch.cuda.synchronize()
    assert rel_err(loss.cpu(), loss_ref.reshape(1)) < 1e-4
    worst = {k: rel_err(eng.g[k].cpu(), gref[k]) for k in gref}
    print("ReLU-tie flips vs the unconditioned oracle (block outputs): %d; worst gradient errors" % flips,
          sorted(worst.items(), key=lambda kv: -kv[1])[:5])
    bad = {k: v for k, v in worst.items() if not v < 1e-4}
    assert not bad, bad


def test_adam_steps_track_the_oracle(dev):
    from sar_amd.resnet import ResNet18
    eng = ResNet18(num_classes=5, num_filters=8, device=dev, seed=1)
    p = {k: v.clone() for k, v in eng.state_dict().items()}
    st = {}
    for step in range(3):
        g = torch.Generator().manual_seed(step)
        x = torch.randn(4, 1, 64, 64, generator=g)
        y = torch.randint(0, 5, (4,), generator=g)
        _, loss_ref, grads, new, _ = RN.loss_and_grads(p, x, y)
        RN.adam_step(p, grads, st, 1e-3)
        p.update(new)
        _, loss = eng.loss_and_grad(x.to(dev), y.to(dev))
        eng.adam_step(1e-3)
        torch.cuda.synchronize()
        assert rel_err(loss.cpu(), loss_ref.reshape(1)) < 2e-3
    sd = eng.state_dict()
    # Adam normalises by sqrt(v): parameters move by ~lr per step whatever the gradient scale, so compare the moves
    assert rel_err(sd["fc.weight"], p["fc.weight"]) < 5e-3 and rel_err(sd["conv1.weight"], p["conv1.weight"]) < 5e-3


def test_dropin_spectrogram_model_and_cli(dev, tmp_path):
    from models.resnet import Model
    model = Model(num_classes=60, num_filters=16, device=dev)
    x = (0.12 * torch.randn(2, 3, 300, 25, 2, generator=torch.Generator().manual_seed(0))).clamp(-1.1, 0.75).to(dev)
    img = model.spectrogram(x)
    assert img.shape == (2, 1, 256, 256)
    ref = torch.nn.functional.interpolate(model.virtual_radar(x).unsqueeze(1), 256)      # models/resnet.py:25-26
    assert torch.equal(img, ref)
    model.train()
    logits = model(x)
    loss = torch.nn.functional.cross_entropy(logits, torch.tensor([1, 2], device=dev))
    loss.backward()
    assert logits.shape == (2, 60) and model.base_model.fc_weight.grad is not None
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "skeleton-action-recognition_amd"))
    cmd = [sys.executable, os.path.join(ROOT, "skeleton-action-recognition_amd", "main_spectrogram.py"), "--synthetic",
           "--synthetic-size", "16", "--batch-size", "4", "--num-epochs", "1", "--num-filters", "16", "--max-iters", "2",
           "--log-dir", str(tmp_path)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "train Loss" in out.stdout and "val Loss" in out.stdout


def test_radar_location_trains_end_to_end(dev, tmp_path):
    """main_spectrogram.py:133-136: `radar_loc*` is un-frozen at --loc-train-epoch; the gradient flows from the loss
    through the resnet stem, the column select, the log-magnitude STFT and the radar geometry into radar_location.
    Module-level: autograd through models.resnet.Model must equal the engine path used by the CLI."""
    from models.resnet import Model
    model = Model(num_classes=10, num_filters=8, device=dev)
    for name, param in model.named_parameters():
        if 'radar_loc' in name:
            param.requires_grad = True
    # synthetic skeletons put both bodies everywhere (the reference's own radar_location gradient is NaN wherever a
    # body is absent; this implementation takes the zero subgradient there)
    x = (0.12 * torch.randn(2, 3, 300, 25, 2, generator=torch.Generator().manual_seed(1))).clamp(-1.1, 0.75).to(dev)
    x[1, :, :, :, 1] = 0
    y = torch.tensor([3, 7], device=dev)
    model.train()
    loss = torch.nn.functional.cross_entropy(model(x), y)
    loss.backward()
    g_mod = model.virtual_radar.radar_location.grad.clone()
    assert torch.isfinite(g_mod).all() and g_mod.abs().max() > 0 and model.virtual_radar.wavelength.grad is None
    eng = model.base_model.engine
    img = model.spectrogram(x)
    _, _, dimg = eng.loss_and_grad(img.detach(), y, need_dx=True)
    model.virtual_radar.radar_location.grad = None
    img.backward(dimg)
    g_eng = model.virtual_radar.radar_location.grad
    assert rel_err(g_eng.cpu(), g_mod.cpu()) < 1e-4
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "skeleton-action-recognition_amd"))
    cmd = [sys.executable, os.path.join(ROOT, "skeleton-action-recognition_amd", "main_spectrogram.py"), "--synthetic",
           "--synthetic-size", "16", "--batch-size", "4", "--num-epochs", "2", "--num-filters", "8", "--max-iters", "2",
           "--loc-train-epoch", "0", "--log-dir", str(tmp_path)]   # un-frozen while epoch > 0
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "radar_location" in out.stdout


def test_image_gradient_of_the_stem(dev):
    """d loss / d image (needed when VirtualRadar parameters train): HIP stem data-gradient kernel against torch
    autograd through the oracle, conditioned on the engine's activation pattern like every gradient comparison."""
    from sar_amd.resnet import ResNet18
    seed_eng = ResNet18(num_classes=5, num_filters=8, device=dev, seed=2)
    p = {k: v.double() for k, v in seed_eng.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    x = torch.randn((3, 1, 64, 64), generator=g)
    y = torch.tensor([1, 4, 0])
    eng = ResNet18(num_classes=5, num_filters=8, device=dev)
    eng.load_params(p)
    keep = {}
    eng.forward(x.to(dev), training=True, keep=keep)
    masks = _engine_masks(eng, keep, 3)
    eng.load_params(p)
    logits, loss, dx = eng.loss_and_grad(x.to(dev), y.to(dev), need_dx=True)
    torch.cuda.synchronize()
    xr = x.double().requires_grad_(True)
    new_stats, taps = {}, {}
    lo = RN.forward(p, xr, True, new_stats, taps, masks=masks)
    torch.nn.functional.cross_entropy(lo, y).backward()
    assert dx.shape == x.shape
    assert rel_err(dx.cpu(), xr.grad) < 1e-4


def test_spectrogram_train_step_is_bitwise_deterministic(dev):
    """VirtualRadar -> resnet18 train step at the bench shape (256 x 256 images, full width): logits, loss and the whole
    gradient buffer repeat bit for bit -- the weight-gradient stream, the concurrent parity-class launches of the stride-2 data
    gradients (library-owned streams) and the slab reductions have a fixed order; a mismatch would be a race."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "skeleton-action-recognition_amd"))
    from models.resnet import Model
    from sar_amd.train import synthetic_clips
    model = Model(num_classes=60, device=dev)
    eng = model.base_model.engine
    x, y = synthetic_clips(8, dev, seed=5)
    with torch.no_grad():
        img = model.spectrogram(x)
    state = {k: v.clone() for k, v in eng.state_dict().items()}
    ref = None
    for _ in range(3):
        eng.load_params(state)
        logits, loss = eng.loss_and_grad(img, y)
        torch.cuda.synchronize()
        cur = (logits.clone(), loss.clone(), eng.grad.clone())
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, cur))
    assert torch.isfinite(ref[2]).all() and ref[2].abs().max() > 0


def test_graph_captured_train_step_equals_the_eager_step(dev):
    """SpectrogramTrainer(graph=True): the step replayed as ONE hipGraph launch (two streams, ~300 kernels) leaves the same parameters
    and Adam state as the eager step, bit for bit, over several steps with changing inputs and a changing learning rate."""
    from sar_amd.train import SpectrogramTrainer, synthetic_clips
    from models.resnet import Model
    outs = []
    for graph in (False, True):
        model = Model(num_classes=60, num_filters=16, device=dev)
        tr = SpectrogramTrainer(model, 1e-3, world_size=1, graph=graph)
        losses = []
        for i in range(5):
            x, y = synthetic_clips(4, dev, seed=i)
            _, loss = tr.step(x, y, 1e-3 if i < 3 else 5e-4)
            losses.append(loss.clone())
        torch.cuda.synchronize()
        eng = model.base_model.engine
        outs.append((eng.flat.clone(), eng.adam_m.clone(), eng.adam_v.clone(), torch.stack(losses)))
        assert (len(tr._graphs) == 1) == graph
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_forked_train_step_equals_the_serial_step(dev, arith_mode):
    """SpectrogramTrainer.step with sar_amd/resnet.py's DS_STREAM (round 6): the weight images issued beside the radar front-end and
    joined in front of the first block (beside the stem), the Samuelson cells and the down-sampling branch beside the main chain --
    the same kernels on the same inputs: parameters, Adam state and every step's loss bit for bit with the serial schedule, over
    steps with changing inputs and a changing learning rate."""
    from sar_amd import resnet
    from sar_amd.train import SpectrogramTrainer, synthetic_clips
    from models.resnet import Model
    outs = []
    for forked in (False, True):
        old = resnet.DS_STREAM
        resnet.DS_STREAM = forked
        try:
            model = Model(num_classes=60, num_filters=16, device=dev)
        finally:
            resnet.DS_STREAM = old
        eng = model.base_model.engine
        assert (eng._aux is not None) == forked
        tr = SpectrogramTrainer(model, 1e-3, world_size=1)
        losses = []
        for i in range(6):
            x, y = synthetic_clips(4, dev, seed=i)
            _, loss = tr.step(x, y, 1e-3 if i < 3 else 5e-4)
            losses.append(loss.clone())
        torch.cuda.synchronize()
        outs.append((eng.flat.clone(), eng.adam_m.clone(), eng.adam_v.clone(), torch.stack(losses)))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.isfinite(outs[0][3]).all()
