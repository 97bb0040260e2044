"""bf16 training QUALITY beyond a gradient cosine (VERDICT r02 weak #1a / next #7).

(1) 400 Nesterov-SGD steps of the full 10-block model on a LEARNABLE synthetic task -- the label is a function of the clip:
    ten smooth prototype motions + noise strong enough to leave an irreducible error --, a fresh seeded batch every step, fp32
    and bf16 engines from the same initial weights on the same data stream: mean loss and top-1 over the last 50 steps must
    agree within 7.5 % / 0.03 (measured 4.9 % / 0.014; at a learning rate where fp32 itself is at the edge of stability the
    trajectories of ANY two arithmetic modes drift apart, tools/bf16_curve.py).
(2) What the 0.908 gradient cosine of the full-shape test (test_gpu_bf16.py) is: the float64 oracle with bfloat16 STORAGE
    emulated at the engine's storage sites (oracle/stgcn.py `quant`, straight-through) reproduces that angle on the CPU
    (tools/bf16_ablation.py: 0.9079 for block 0 with every site rounded; rounding ONLY the weights -- plain mixed precision --
    already gives 0.962, any single activation tensor 0.953-0.977; rounding the gradients adds nothing: 0.9074 with fp32
    gradients).  The angle is a property of this ill-conditioned probe (random weights, noise clips, gradients that are
    sums of ~1e6 cancelling terms), not of one tensor of the engine.  So the engine is compared with the EMULATED network:
    its gradient must be as close to the emulated bf16 gradient as two float64 runs of that network with different rounding
    luck are to each other."""
import pytest
import torch

from oracle import stgcn as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from sar_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _task(dev, classes=10, T=64, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    t = torch.linspace(0, 1, T, device=dev)
    # prototype k: a few low-frequency sinusoids per (coordinate, joint)
    freq = torch.randint(1, 4, (classes, 3, 1, 25), generator=g, device=dev).float()
    phase = 6.28318 * torch.rand((classes, 3, 1, 25), generator=g, device=dev)
    amp = 0.5 + torch.rand((classes, 3, 1, 25), generator=g, device=dev)
    proto = 0.1 * amp * torch.sin(6.28318 * freq * t.view(1, 1, T, 1) + phase)        # (classes, 3, T, 25)

    def batch(n, step):
        gb = torch.Generator(device=dev).manual_seed(1000 + step)
        k = torch.randint(0, classes, (n,), generator=gb, device=dev)                   # the motion shown
        x = proto[k] + 0.25 * torch.randn((n, 3, T, 25), generator=gb, device=dev)      # noise 2.5x the signal amplitude
        x = torch.stack([x, torch.zeros_like(x)], dim=-1)                               # second body absent
        # 25 % of the labels are drawn at random: an irreducible error (best top-1 0.775, cross-entropy ~0.95) remains, so the
        # comparison is made on a plateau both engines have to find, not on a loss that goes to zero
        flip = torch.rand(n, generator=gb, device=dev) < 0.25
        y = torch.where(flip, torch.randint(0, classes, (n,), generator=gb, device=dev), k)
        return x.contiguous(), y
    return batch


def test_bf16_trains_a_learnable_task_like_fp32(dev):
    """400 Nesterov-SGD steps of the full model on the learnable task above: fp32, bf16 storage and -- the YARDSTICK -- the fp32 engine
    once more from weights ONE ulp away.  SGD at this rate separates any two runs after a few steps; at 400 steps two fp32 runs end
    0.02-0.07 / 0.00-0.04 apart (loss / top-1 over the last 50 steps), differently on every change of a summation order anywhere in
    the step (round 6: a fixed band around ONE fp32 run failed on the first layer's new weight-gradient kernel).  The bf16 engine
    must end inside the band the two fp32 runs span, widened by 1.5 x their distance and the margin measured for bf16 storage
    (5 % of the loss, 0.02 top-1: fp32 1.204 / 0.748, bf16 1.263 / 0.734, tools/bf16_curve.py)."""
    from sar_amd.stgcn import STGCN
    classes, steps, bs = 10, 400, 32
    batch = _task(dev, classes)
    p = O.init_params(classes, seed=7, dtype=torch.float64)
    p_ulp = {k: (torch.nextafter(v.float(), torch.full_like(v.float(), float("inf"))).double() if v.is_floating_point() else v)
             for k, v in p.items()}
    res = {}
    for name, mode, params in (("fp32", "fp32", p), ("bf16", "bf16", p), ("fp32+1ulp", "fp32", p_ulp)):
        eng = STGCN(num_classes=classes, device=dev, mfma=mode)
        eng.load_params(params)
        losses, correct = [], []
        for s in range(steps):
            x, y = batch(bs, s)
            logits, loss = eng.loss_and_grad(x, y)
            eng.sgd_step(0.02 if s < 300 else 0.002)    # (at 0.05 the fp32 run itself is at the edge of stability: trajectories of
            #                                              any two arithmetic modes then drift apart -- tools/bf16_curve.py)
            losses.append(loss.reshape(()))
            correct.append((logits.argmax(1) == y).float().mean())
        losses, correct = torch.stack(losses).cpu(), torch.stack(correct).cpu()
        res[name] = (losses[:10].mean().item(), losses[-50:].mean().item(), correct[-50:].mean().item())
        del eng
    (f0, fl, fa), (b0, bl, ba), (c0, cl, ca) = res["fp32"], res["bf16"], res["fp32+1ulp"]
    print("learnable task, %d steps: fp32 loss %.4f -> %.4f top-1 %.3f | bf16 loss %.4f -> %.4f top-1 %.3f | fp32 + 1 ulp %.4f -> %.4f top-1 %.3f"
          % (steps, f0, fl, fa, b0, bl, ba, c0, cl, ca))
    for l_, a_ in ((fl, fa), (cl, ca), (bl, ba)):
        assert l_ < 0.5 * f0 and a_ > 0.6, "the task must be learnable: %s" % (res,)
        assert 0.5 < l_, "an irreducible error must remain, or the comparison says nothing"
    assert abs(bl - 0.5 * (fl + cl)) <= 1.5 * abs(fl - cl) + 0.05 * fl, res
    assert abs(ba - 0.5 * (fa + ca)) <= 1.5 * abs(fa - ca) + 0.02, res


def test_engine_gradient_is_the_gradient_of_the_bf16_storage_network(dev):
    """Full NTU shape (10 blocks, T = 300).  Reference A: float64 oracle with bf16 storage emulated at every site.
    Reference B: the same emulation in float32 arithmetic (a second draw of 'rounding luck': other values land on the
    other side of a bfloat16 rounding boundary).  The
    engine's gradient must be closer to A than the plain float64 gradient is (i.e. it IS the gradient of the rounded
    network, not a noisy float64 gradient), and about as close to A as B is."""
    from sar_amd.stgcn import STGCN
    blocks = list(O.BLOCKS)
    p = O.randomize_affine(O.init_params(60, seed=3, dtype=torch.float64, blocks=blocks), seed=4)
    x, y = O.synthetic_batch(2, seed=3, T=300, num_classes=60)
    ALL = {"x0", "g", "h", "u", "r", "y", "w"}
    _, _, g64, _, _ = O.loss_and_grads(p, x.double(), y, blocks=blocks)
    _, _, gA, _, _ = O.loss_and_grads(p, x.double(), y, blocks=blocks, quant=ALL)
    p32 = {k: (v.float() if v.is_floating_point() else v) for k, v in p.items()}
    _, _, gB, _, _ = O.loss_and_grads(p32, x.float(), y, blocks=blocks, quant=ALL)       # float32 arithmetic: other rounding ties
    eng = STGCN(num_classes=60, device=dev, blocks=blocks, mfma="bf16")
    eng.load_params(p)
    eng.loss_and_grad(x.to(dev), y.to(dev))
    torch.cuda.synchronize()

    def cos(a, b):
        return ((a * b).sum() / (a.norm() * b.norm())).item()
    worst = {"engine~A": 1.0, "f64~A": 1.0, "B~A": 1.0}
    for k, g in gA.items():
        if k.endswith("kernel") and g.numel() >= 64:
            ge = eng.g[k].cpu().double()
            worst["engine~A"] = min(worst["engine~A"], cos(ge, g))
            worst["f64~A"] = min(worst["f64~A"], cos(g64[k], g))
            worst["B~A"] = min(worst["B~A"], cos(gB[k].double(), g))
    print("worst kernel-gradient cosines at the NTU shape:", {k: round(v, 4) for k, v in worst.items()})
    # measured: engine~A 0.934, B~A 0.953 (two EMULATIONS that differ only in float32 / float64 arithmetic!), f64~A 0.892
    assert worst["engine~A"] > worst["f64~A"] + 0.02, worst
    assert worst["engine~A"] > worst["B~A"] - 0.035 and worst["engine~A"] > 0.91, worst
