"""Oracle: ST-GCN forward / backward / optimizer on the CPU (torch CPU ops,
float32 or float64), restating the reference's TensorFlow/Keras model.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows, line by line:
  models/stgcn.py:6-8      INITIALIZER  (VarianceScaling 2.0, fan_out, truncated normal)
  models/stgcn.py:11-64    SpatioTemporalGraphConv (sgcn -> BN/ReLU/Conv 9x1/BN -> +res -> ReLU)
  models/stgcn.py:101-160  Model (data_bn, 10 blocks, pool, mean over persons, 1x1 logits)
  models/gcn.py:187-209    GraphConvTD (1x1 conv to K*F, reshape (N,K,F,T,V), einsum nkctv,kvw->nctw)
  main_gnn.py:219-239      loss = sum(softmax CE)/global_batch ; no L2 term is ever added
  main_gnn.py:303-314      PiecewiseConstantDecay + SGD(momentum .9, nesterov)

Keras/TF semantics encoded here (TF 2.0-2.3; TensorFlow is not installable in
the build image, so these are stated from the published Keras behaviour):
  * Conv2D kernels are HWIO (kh, kw, Cin, Cout), use_bias=True, zero-init bias.
  * padding='same' puts the extra pad row at the END: for the 9x1 stride-2 conv
    on T=300 the pads are (3, 4).  The strided 1x1 residual conv has no pad.
  * BatchNormalization(axis=1): eps=1e-3, momentum=0.99, train mode normalises
    with the biased batch variance; gamma=1, beta=0, moving mean 0 / var 1.
    Moving variance: the fused 4-D NCHW path feeds the unbiased batch variance,
    the 3-D data_bn path the biased one.
  * SGD nesterov: v <- m*v - lr*g ; w <- w + m*v - lr*g.
  * PiecewiseConstantDecay: value[i] while step <= boundary[i].

Parity status: PARITY UNPINNED for the floating-point model -- the reference
holds no test vectors for it and cannot be imported (models/stgcn.py:2 imports a
non-existent package `model`; TensorFlow absent).  The oracle is cross-checked
by (i) an independent float64 numpy re-derivation on small shapes
(oracle/stgcn_np.py), (ii) finite-difference gradient checks, (iii) the
bit-exact adjacency (oracle/graph.py, pinned).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .graph import spatial_adjacency

BN_EPS = 1e-3          # Keras BatchNormalization default epsilon
BN_MOMENTUM = 0.99     # Keras BatchNormalization default momentum

# (filters, stride, residual) of the ten blocks, models/stgcn.py:113-123
BLOCKS = [(64, 1, False), (64, 1, True), (64, 1, True), (64, 1, True),
          (128, 2, True), (128, 1, True), (128, 1, True),
          (256, 2, True), (256, 1, True), (256, 1, True)]
KS = 3   # spatial kernel size (number of adjacency slices)
KT = 9   # temporal kernel size


def same_pad(T, k, s):
    """TF 'SAME' padding along one axis -> (out, pad_begin, pad_end)."""
    out = -(-T // s)
    total = max((out - 1) * s + k - T, 0)
    return out, total // 2, total - total // 2


def _trunc_normal(shape, fan_out, gen, dtype):
    # VarianceScaling(scale=2, mode=fan_out, truncated_normal): stddev corrected by
    # .87962566103423978 so that the truncated distribution has the requested variance.
    std = math.sqrt(2.0 / fan_out) / .87962566103423978
    w = torch.empty(shape, dtype=torch.float64)
    torch.nn.init.trunc_normal_(w, 0.0, std, -2 * std, 2 * std, generator=gen)
    return w.to(dtype)


def block_residual_kind(cin, f, s, residual):
    """models/stgcn.py:41-56."""
    if not residual:
        return "none"
    if cin == f and s == 1:
        return "identity"
    return "conv"


def init_params(num_classes=60, in_channels=3, num_node=25, seed=0, dtype=torch.float32, blocks=None):
    """models/stgcn.py:101-133 (+ lazy residual build :41-56).  Returns a flat
    ordered dict name -> tensor in Keras layouts, plus BN moving statistics."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    p["A"] = torch.tensor(spatial_adjacency().astype(np.float32)).to(dtype)  # stgcn.py:105-109
    nch = num_node * in_channels
    p["data_bn.gamma"] = torch.ones(nch, dtype=dtype)
    p["data_bn.beta"] = torch.zeros(nch, dtype=dtype)
    p["data_bn.moving_mean"] = torch.zeros(nch, dtype=dtype)
    p["data_bn.moving_var"] = torch.ones(nch, dtype=dtype)
    cin = in_channels
    for i, (f, s, res) in enumerate(blocks or BLOCKS):
        pre = "l%d." % i
        p[pre + "gcn.kernel"] = _trunc_normal((1, 1, cin, KS * f), 1 * 1 * KS * f, g, dtype)
        p[pre + "gcn.bias"] = torch.zeros(KS * f, dtype=dtype)
        for bn in ("bn1", "bn2"):
            p[pre + bn + ".gamma"] = torch.ones(f, dtype=dtype)
            p[pre + bn + ".beta"] = torch.zeros(f, dtype=dtype)
            p[pre + bn + ".moving_mean"] = torch.zeros(f, dtype=dtype)
            p[pre + bn + ".moving_var"] = torch.ones(f, dtype=dtype)
        p[pre + "tcn.kernel"] = _trunc_normal((KT, 1, f, f), KT * 1 * f, g, dtype)
        p[pre + "tcn.bias"] = torch.zeros(f, dtype=dtype)
        if block_residual_kind(cin, f, s, res) == "conv":
            p[pre + "res.kernel"] = _trunc_normal((1, 1, cin, f), f, g, dtype)
            p[pre + "res.bias"] = torch.zeros(f, dtype=dtype)
            p[pre + "res_bn.gamma"] = torch.ones(f, dtype=dtype)
            p[pre + "res_bn.beta"] = torch.zeros(f, dtype=dtype)
            p[pre + "res_bn.moving_mean"] = torch.zeros(f, dtype=dtype)
            p[pre + "res_bn.moving_var"] = torch.ones(f, dtype=dtype)
        cin = f
    p["logits.kernel"] = _trunc_normal((1, 1, cin, num_classes), num_classes, g, dtype)
    p["logits.bias"] = torch.zeros(num_classes, dtype=dtype)
    return p


def randomize_affine(p, seed=1, scale=0.2):
    """Perturb biases / BN affine parameters away from their (0, 1) init so that
    parity tests exercise every term (a zero bias hides a missing bias path)."""
    g = torch.Generator().manual_seed(seed)
    for k, v in p.items():
        if k.endswith(".bias") or k.endswith(".beta"):
            v.copy_((torch.randn(v.shape, generator=g, dtype=torch.float64) * scale).to(v.dtype))
        elif k.endswith(".gamma"):
            v.copy_((1.0 + torch.randn(v.shape, generator=g, dtype=torch.float64) * scale).to(v.dtype))
        elif k.endswith(".moving_mean"):
            v.copy_((torch.randn(v.shape, generator=g, dtype=torch.float64) * scale).to(v.dtype))
        elif k.endswith(".moving_var"):
            v.copy_((1.0 + torch.rand(v.shape, generator=g, dtype=torch.float64)).to(v.dtype))
    return p


def is_trainable(name):
    """main_gnn.py:228-232: everything except the adjacency and BN moving stats."""
    return not (name == "A" or ".moving_" in name)


def trainable_names(p):
    return [k for k in p if is_trainable(k)]


# ----------------------------------------------------------------------------- ops
def hwio_to_oihw(k):
    return k.permute(3, 2, 0, 1).contiguous()


def batch_norm(x, gamma, beta, mm, mv, training, axes, unbiased_moving, new_stats=None, prefix=None):
    """Keras BatchNormalization(axis=1) on a tensor whose channel axis is 1."""
    shape = [1, -1] + [1] * (x.dim() - 2)
    if training:
        mean = x.mean(dim=axes)
        var = x.var(dim=axes, unbiased=False)
        if new_stats is not None:
            n = x.numel() // x.shape[1]
            v_mov = var * (n / (n - 1)) if unbiased_moving else var
            new_stats[prefix + ".moving_mean"] = (mm * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM))
            new_stats[prefix + ".moving_var"] = (mv * BN_MOMENTUM + v_mov.detach() * (1 - BN_MOMENTUM))
    else:
        mean, var = mm, mv
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean.view(shape)) * (inv * gamma).view(shape) + beta.view(shape)


def graph_conv_td(x, kernel, bias, A):
    """models/gcn.py:199-209.  x (B,Cin,T,V); kernel (1,1,Cin,K*F); A (K,V,V)."""
    y = F.conv2d(x, hwio_to_oihw(kernel), bias)
    B, KF, T, V = y.shape
    K = A.shape[0]
    y = y.reshape(B, K, KF // K, T, V)
    return torch.einsum("nkctv,kvw->nctw", y, A)


def temporal_conv(x, kernel, bias, stride):
    """models/stgcn.py:29-36: Conv2D(F,[9,1],strides=[s,1],'same') NCHW."""
    kt = kernel.shape[0]
    _, pb, pe = same_pad(x.shape[2], kt, stride)
    x = F.pad(x, (0, 0, pb, pe))
    return F.conv2d(x, hwio_to_oihw(kernel), bias, stride=(stride, 1))


def data_bn(x, p, training, new_stats=None):
    """models/stgcn.py:136-147.  x (N,C,T,V,M) -> (N*M,C,T,V)."""
    N, C, T, V, M = x.shape
    h = x.permute(0, 4, 3, 1, 2).reshape(N * M, V * C, T)
    h = batch_norm(h, p["data_bn.gamma"], p["data_bn.beta"], p["data_bn.moving_mean"],
                   p["data_bn.moving_var"], training, (0, 2), False, new_stats, "data_bn")
    h = h.reshape(N, M, V, C, T).permute(0, 1, 3, 4, 2).reshape(N * M, C, T, V)
    return h


class _RoundBF16(torch.autograd.Function):
    """Where the bf16 configuration stores a tensor as bfloat16: the VALUE is rounded in the forward
    pass and the GRADIENT flowing back through the same tensor in the backward pass (the engine stores both the activation
    and its gradient as bfloat16); the rounding itself is treated as the identity (straight-through), which is exactly
    what a network that computes with the rounded values does."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.to(torch.bfloat16).to(x.dtype) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.bwd else g), None, None


def _q(x, site, quant):
    """quant: None (the reference arithmetic) or a set of storage sites to emulate in bfloat16: 'x0' data_bn output, 'g'
    graph-conv output, 'h' the BN + ReLU operand staged for the temporal conv, 'u' temporal-conv output, 'r' residual-conv
    output, 'y' block output, 'w' the conv weights as MFMA operands (forward value only); '<site>:fwd' rounds the value
    but not the gradient."""
    if not quant:
        return x
    if site in quant:
        return _RoundBF16.apply(x, True, site != "w")
    if site + ":fwd" in quant:
        return _RoundBF16.apply(x, True, False)
    return x


def _relu(z, masks, site):
    """ReLU, or multiplication by a prescribed activation pattern (see oracle/resnet.py:_relu)."""
    if masks is None:
        return torch.relu(z)
    return z * masks[site].to(z.dtype)


def st_block(x, p, i, A, training, new_stats=None, taps=None, blocks=None, masks=None, quant=None):
    """models/stgcn.py:58-64 for block i.  x (B,Cin,T,V).  quant: see _q (bf16-storage emulation, tests only)."""
    f, s, res = (blocks or BLOCKS)[i]
    pre = "l%d." % i
    kind = block_residual_kind(x.shape[1], f, s, res)
    if kind == "none":
        r = None
    elif kind == "identity":
        r = x
    else:
        r = _q(F.conv2d(x, hwio_to_oihw(_q(p[pre + "res.kernel"], "w", quant)), p[pre + "res.bias"], stride=(s, 1)), "r", quant)
        r = batch_norm(r, p[pre + "res_bn.gamma"], p[pre + "res_bn.beta"], p[pre + "res_bn.moving_mean"],
                       p[pre + "res_bn.moving_var"], training, (0, 2, 3), True, new_stats, pre + "res_bn")
    g = _q(graph_conv_td(x, _q(p[pre + "gcn.kernel"], "w", quant), p[pre + "gcn.bias"], A), "g", quant)
    h = batch_norm(g, p[pre + "bn1.gamma"], p[pre + "bn1.beta"], p[pre + "bn1.moving_mean"],
                   p[pre + "bn1.moving_var"], training, (0, 2, 3), True, new_stats, pre + "bn1")
    h_pre = h
    h = _q(_relu(h, masks, pre + "h"), "h", quant)
    u = _q(temporal_conv(h, _q(p[pre + "tcn.kernel"], "w", quant), p[pre + "tcn.bias"], s), "u", quant)
    z = batch_norm(u, p[pre + "bn2.gamma"], p[pre + "bn2.beta"], p[pre + "bn2.moving_mean"],
                   p[pre + "bn2.moving_var"], training, (0, 2, 3), True, new_stats, pre + "bn2")
    if r is not None:
        z = z + r
    y = _q(_relu(z, masks, pre + "y"), "y", quant)
    if taps is not None:
        taps[pre + "g"] = g
        taps[pre + "u"] = u
        taps[pre + "y"] = y
        taps[pre + "h_pre"] = h_pre      # the two ReLU inputs of the block (tests: activation-pattern ties)
        taps[pre + "y_pre"] = z
    return y


def forward(p, x, training, new_stats=None, taps=None, blocks=None, masks=None, quant=None):
    """models/stgcn.py:135-160.  x (N,C,T,V,M) -> logits (N, classes)."""
    N, C, T, V, M = x.shape
    h = _q(data_bn(x, p, training, new_stats), "x0", quant)
    if taps is not None:
        taps["x0"] = h
    A = p["A"]
    for i in range(len(blocks or BLOCKS)):
        h = st_block(h, p, i, A, training, new_stats, taps, blocks, masks, quant)
    pooled = h.mean(dim=(2, 3))                    # GlobalAveragePooling2D, stgcn.py:154
    feat = pooled.reshape(N, M, -1).mean(dim=1)    # stgcn.py:155-156
    if taps is not None:
        taps["feat"] = feat
    kernel = p["logits.kernel"]                    # (1,1,256,classes)
    return feat @ kernel[0, 0] + p["logits.bias"]  # 1x1 conv on a 1x1 map, stgcn.py:157-158


def loss_fn(logits, labels, global_batch_size):
    """main_gnn.py:224-226.  labels: int64 class ids (the reference one-hot encodes them)."""
    ce = F.cross_entropy(logits, labels, reduction="sum")
    return ce * (1.0 / global_batch_size)


def loss_and_grads(p, x, labels, global_batch_size=None, blocks=None, masks=None, quant=None):
    """One train_step's differentiable part (main_gnn.py:221-233)."""
    names = trainable_names(p)
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    new_stats, taps = {}, {}
    logits = forward(q, x, True, new_stats, taps, blocks, masks, quant)
    gbs = global_batch_size or x.shape[0]
    loss = loss_fn(logits, labels, gbs)
    used = names
    grads = torch.autograd.grad(loss, [leaves[k] for k in used])
    return logits.detach(), loss.detach(), dict(zip(used, grads)), new_stats, {k: v.detach() for k, v in taps.items()}


def lr_schedule(iteration, base_lr=0.1, steps=(10, 50), batch_size=64):
    """main_gnn.py:303-308 PiecewiseConstantDecay(boundaries=(step*40000)//batch_size)."""
    boundaries = [(s * 40000) // batch_size for s in steps]
    values = [base_lr * (0.1 ** i) for i in range(len(steps) + 1)]
    for b, v in zip(boundaries, values):
        if iteration <= b:
            return v
    return values[-1]


def sgd_nesterov_step(p, grads, velocity, lr, momentum=0.9):
    """tf.keras.optimizers.SGD(momentum, nesterov=True), main_gnn.py:312-314."""
    for k, g in grads.items():
        v = velocity.setdefault(k, torch.zeros_like(p[k]))
        v.mul_(momentum).sub_(lr * g)
        p[k].add_(momentum * v - lr * g)


def synthetic_batch(n, seed=0, T=300, V=25, M=2, C=3, num_classes=60, dtype=torch.float32,
                    single_body_frac=0.8):
    """SURVEY section 8(d) synthetic NTU-like clips: 0.12*randn clamped to [-1.1, .75],
    second body zeroed for ~80 % of the clips."""
    g = torch.Generator().manual_seed(seed)
    x = (0.12 * torch.randn(n, C, T, V, M, generator=g)).clamp_(-1.1, 0.75)
    if M > 1:
        drop = torch.rand(n, generator=g) < single_body_frac
        x[drop, :, :, :, 1:] = 0
    g2 = torch.Generator().manual_seed(seed + 1)
    y = torch.randint(0, num_classes, (n,), generator=g2)
    return x.to(dtype), y
