"""Oracle: the reference's 1-channel ResNet-18 (models/resnet18.py:131-254, BasicBlock :26-72) and the
VirtualRadar->resnet wrapper (models/resnet.py:23-28) restated with torch CPU functional ops on a flat
parameter dict (fp32 or fp64), plus the train step of main_spectrogram.py:105-111,152-158
(CrossEntropyLoss mean, Adam betas (0.9,0.999) eps 1e-8).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Parity status: PINNED -- tests/golden/resnet18_tiny.npz holds parameters, input, logits and gradients produced by
the reference's own models/resnet18.py (imported by file path in the build container,
tests/golden/make_golden_resnet.py); tests/test_oracle_resnet.py checks this restatement against them.

torch semantics encoded: Conv2d bias=False, OIHW weights, padding = k//2 (7x7 pad 3 stride 2; 3x3 pad 1; 1x1 pad 0);
BatchNorm2d eps 1e-5, momentum 0.1 (running = 0.9*running + 0.1*batch, running_var from the UNBIASED batch variance,
normalisation with the biased one); MaxPool2d(3, 2, 1); AdaptiveAvgPool2d(1); Linear with bias.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LAYERS = [2, 2, 2, 2]          # resnet18, models/resnet18.py:274


def param_names(num_filters=64):
    """Ordered (name, kind) list in the reference's state_dict naming."""
    names = ["conv1.weight", "bn1"]
    inpl = num_filters
    for li, nblocks in enumerate(LAYERS):
        planes = num_filters * (2 ** li)
        for bi in range(nblocks):
            pre = "layer%d.%d." % (li + 1, bi)
            stride = 2 if (li > 0 and bi == 0) else 1
            names += [pre + "conv1.weight", pre + "bn1", pre + "conv2.weight", pre + "bn2"]
            if stride != 1 or inpl != planes:
                names += [pre + "downsample.0.weight", pre + "downsample.1"]
            inpl = planes
    names += ["fc.weight", "fc.bias"]
    return names


def init_params(num_classes=60, num_filters=64, seed=0, dtype=torch.float32):
    """models/resnet18.py:181-186: kaiming_normal_(fan_out, relu) conv weights, BN weight 1 / bias 0, running stats 0 / 1;
    nn.Linear default init for fc.  (Seeded with a torch CPU generator -- not bit-identical to any reference run.)"""
    g = torch.Generator().manual_seed(seed)
    p = {}

    def conv(name, cout, cin, k):
        p[name] = (torch.randn((cout, cin, k, k), generator=g, dtype=torch.float64) * (2.0 / (cout * k * k)) ** 0.5).to(dtype)

    def bn(name, c):
        p[name + ".weight"], p[name + ".bias"] = torch.ones(c, dtype=dtype), torch.zeros(c, dtype=dtype)
        p[name + ".running_mean"], p[name + ".running_var"] = torch.zeros(c, dtype=dtype), torch.ones(c, dtype=dtype)

    conv("conv1.weight", num_filters, 1, 7)
    bn("bn1", num_filters)
    inpl = num_filters
    for li, nblocks in enumerate(LAYERS):
        planes = num_filters * (2 ** li)
        for bi in range(nblocks):
            pre = "layer%d.%d." % (li + 1, bi)
            stride = 2 if (li > 0 and bi == 0) else 1
            conv(pre + "conv1.weight", planes, inpl, 3)
            bn(pre + "bn1", planes)
            conv(pre + "conv2.weight", planes, planes, 3)
            bn(pre + "bn2", planes)
            if stride != 1 or inpl != planes:
                conv(pre + "downsample.0.weight", planes, inpl, 1)
                bn(pre + "downsample.1", planes)
            inpl = planes
    bound = 1.0 / inpl ** 0.5
    p["fc.weight"] = ((torch.rand((num_classes, inpl), generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
    p["fc.bias"] = ((torch.rand((num_classes,), generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
    return p


def _bn(x, p, name, training, new_stats):
    w, b = p[name + ".weight"], p[name + ".bias"]
    rm, rv = p[name + ".running_mean"], p[name + ".running_var"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if new_stats is not None:
            n = x.numel() // x.shape[1]
            new_stats[name + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach()
            new_stats[name + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var.detach() * n / (n - 1)
    else:
        mean, var = rm, rv
    sh = (1, -1, 1, 1)
    return (x - mean.view(sh)) * (torch.rsqrt(var + BN_EPS) * w).view(sh) + b.view(sh)


def _relu(z, masks, site):
    """ReLU, or -- when `masks` is given -- multiplication by a prescribed activation pattern (bool tensor).  Used by
    the GPU parity tests: two float32 implementations may put a pre-activation that is zero to within rounding on
    different sides; conditioning the oracle on the product's pattern removes that (measure-zero) ambiguity."""
    if masks is None:
        return torch.relu(z)
    return z * masks[site].to(z.dtype)


def forward(p, x, training=True, new_stats=None, taps=None, num_filters=None, masks=None):
    """models/resnet18.py:235-251.  x (B,1,H,W) -> logits."""
    h = F.conv2d(x, p["conv1.weight"], None, stride=2, padding=3)
    if taps is not None:
        taps["conv1"] = h
    h = _relu(_bn(h, p, "bn1", training, new_stats), masks, "bn1")
    h = F.max_pool2d(h, 3, 2, 1)
    if taps is not None:
        taps["pool"] = h
    for li, nblocks in enumerate(LAYERS):
        for bi in range(nblocks):
            pre = "layer%d.%d." % (li + 1, bi)
            stride = 2 if (li > 0 and bi == 0) else 1
            identity = h
            o = F.conv2d(h, p[pre + "conv1.weight"], None, stride=stride, padding=1)
            o = _relu(_bn(o, p, pre + "bn1", training, new_stats), masks, pre + "bn1")
            o = F.conv2d(o, p[pre + "conv2.weight"], None, stride=1, padding=1)
            o = _bn(o, p, pre + "bn2", training, new_stats)
            if (pre + "downsample.0.weight") in p:
                identity = F.conv2d(h, p[pre + "downsample.0.weight"], None, stride=stride)
                identity = _bn(identity, p, pre + "downsample.1", training, new_stats)
            h = _relu(o + identity, masks, pre + "out")
            if taps is not None:
                taps[pre + "out"] = h
    h = h.mean(dim=(2, 3))
    return h @ p["fc.weight"].t() + p["fc.bias"]


def trainable(p):
    return [k for k in p if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


def loss_and_grads(p, x, labels, masks=None):
    """main_spectrogram.py:152-157: CrossEntropyLoss() (mean over the batch) + backward."""
    names = trainable(p)
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    new_stats, taps = {}, {}
    logits = forward(q, x, True, new_stats, taps, masks=masks)
    loss = F.cross_entropy(logits, labels)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    return logits.detach(), loss.detach(), dict(zip(names, grads)), new_stats, {k: v.detach() for k, v in taps.items()}


def adam_step(p, grads, state, lr, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults (main_spectrogram.py:106)."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    for k, g in grads.items():
        m = state.setdefault("m." + k, torch.zeros_like(p[k]))
        v = state.setdefault("v." + k, torch.zeros_like(p[k]))
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        denom = (v.sqrt() / (1 - betas[1] ** t) ** 0.5).add_(eps)
        p[k].addcdiv_(m, denom, value=-lr / (1 - betas[0] ** t))


def cyclic_lr(epoch, base_lr=1e-4, max_lr=0.1, step_size_up=10):
    """torch CyclicLR(base_lr=1e-4, max_lr, step_size_up, mode='triangular', cycle_momentum=False) stepped once per
    epoch (main_spectrogram.py:107-111,189)."""
    import math
    total = 2 * step_size_up
    cycle = math.floor(1 + epoch / total)
    xx = 1 + epoch / total - cycle
    scale = xx / 0.5 if xx <= 0.5 else (xx - 1) / (0.5 - 1)
    return base_lr + (max_lr - base_lr) * scale


def spectrogram_model_forward(p, x, radar, image_size=256, training=True):
    """models/resnet.py:23-28 given a callable `radar` (B,3,T,V,M)->(B,n_fft,F) torch tensor."""
    s = radar(x).unsqueeze(1)
    s = F.interpolate(s, image_size)
    return forward(p, s, training)
