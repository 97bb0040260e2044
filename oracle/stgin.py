"""Oracle: the ST-GIN sibling model (graph isomorphism convolution) on the CPU -- torch CPU ops, float32 / float64.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates, line by line:
  models/gcn.py:112-163   GraphIsoConvTD: A_ = concat(A, diag(1 + epsilon)); x' = einsum('nctv,kvw->nkctw', x, A_);
                          per slice k an MLP  Conv2D(h,1x1) -> BN -> ReLU -> Conv2D(h,1x1) -> BN -> ReLU  (filters = [h, h],
                          return_logits = False); sum over k; `epsilon` is a trainable scalar (init 0)
  models/stgin.py:11-66   SpatioTemporalGraphConv: sgcn = GraphIsoConvTD([filters/2, filters/2], kernel_size = 3), then the
                          same tgcn / residual / ReLU as models/stgcn.py (BN -> ReLU -> Conv2D(filters,[9,1],stride) -> BN)
  models/stgin.py:82-140  Model: adjacency = Graph().A[:2] (non-trainable), data_bn, the ten blocks of ST-GCN, pool, logits

Parity status: PARITY UNPINNED like oracle/stgcn.py (models/stgin.py:2 imports the non-existent package `model`;
TensorFlow is absent); the shared pieces (BatchNorm, temporal convolution, data_bn, loss) are oracle/stgcn.py's, the
adjacency slices are the pinned ones of oracle/graph.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import stgcn as S
from .graph import spatial_adjacency

BLOCKS = S.BLOCKS
KS = 3


def init_params(num_classes=60, in_channels=3, num_node=25, seed=0, dtype=torch.float32, blocks=None):
    g = torch.Generator().manual_seed(seed)
    p = {}
    p["A"] = torch.tensor(spatial_adjacency().astype(np.float32)).to(dtype)[:2].clone()       # models/stgin.py:87-90
    nch = num_node * in_channels
    p["data_bn.gamma"], p["data_bn.beta"] = torch.ones(nch, dtype=dtype), torch.zeros(nch, dtype=dtype)
    p["data_bn.moving_mean"], p["data_bn.moving_var"] = torch.zeros(nch, dtype=dtype), torch.ones(nch, dtype=dtype)

    def bn(prefix, c):
        p[prefix + ".gamma"], p[prefix + ".beta"] = torch.ones(c, dtype=dtype), torch.zeros(c, dtype=dtype)
        p[prefix + ".moving_mean"], p[prefix + ".moving_var"] = torch.zeros(c, dtype=dtype), torch.ones(c, dtype=dtype)

    cin = in_channels
    for i, (f, s, res) in enumerate(blocks or BLOCKS):
        pre, h = "l%d." % i, f // 2
        for k in range(KS):                                                   # models/gcn.py:124-142
            q = pre + "mlp%d." % k
            p[q + "c1.kernel"] = S._trunc_normal((1, 1, cin, h), h, g, dtype)
            p[q + "c1.bias"] = torch.zeros(h, dtype=dtype)
            bn(q + "bn1", h)
            p[q + "c2.kernel"] = S._trunc_normal((1, 1, h, h), h, g, dtype)
            p[q + "c2.bias"] = torch.zeros(h, dtype=dtype)
            bn(q + "bn2", h)
        p[pre + "epsilon"] = torch.zeros((), dtype=dtype)                    # models/gcn.py:144-147
        bn(pre + "bn1", h)                                                    # tgcn's first BatchNorm sees h channels
        p[pre + "tcn.kernel"] = S._trunc_normal((S.KT, 1, h, f), S.KT * f, g, dtype)
        p[pre + "tcn.bias"] = torch.zeros(f, dtype=dtype)
        bn(pre + "bn2", f)
        if S.block_residual_kind(cin, f, s, res) == "conv":
            p[pre + "res.kernel"] = S._trunc_normal((1, 1, cin, f), f, g, dtype)
            p[pre + "res.bias"] = torch.zeros(f, dtype=dtype)
            bn(pre + "res_bn", f)
        cin = f
    p["logits.kernel"] = S._trunc_normal((1, 1, cin, num_classes), num_classes, g, dtype)
    p["logits.bias"] = torch.zeros(num_classes, dtype=dtype)
    return p


def graph_iso_conv(x, p, pre, A, training, new_stats, taps, masks):
    """models/gcn.py:149-163"""
    V = A.shape[-1]
    self_conn = torch.diag(torch.ones(V, dtype=x.dtype) + p[pre + "epsilon"]).unsqueeze(0)
    A_ = torch.cat([A, self_conn], dim=0)
    z = torch.einsum("nctv,kvw->nkctw", x, A_)
    out = 0
    for k in range(KS):
        q = pre + "mlp%d." % k
        a = F.conv2d(z[:, k], S.hwio_to_oihw(p[q + "c1.kernel"]), p[q + "c1.bias"])
        a = S.batch_norm(a, p[q + "bn1.gamma"], p[q + "bn1.beta"], p[q + "bn1.moving_mean"], p[q + "bn1.moving_var"], training,
                         (0, 2, 3), True, new_stats, q + "bn1")
        if taps is not None:
            taps[q + "h1_pre"] = a
        a = S._relu(a, masks, q + "h1")
        a = F.conv2d(a, S.hwio_to_oihw(p[q + "c2.kernel"]), p[q + "c2.bias"])
        a = S.batch_norm(a, p[q + "bn2.gamma"], p[q + "bn2.beta"], p[q + "bn2.moving_mean"], p[q + "bn2.moving_var"], training,
                         (0, 2, 3), True, new_stats, q + "bn2")
        if taps is not None:
            taps[q + "h2_pre"] = a
        a = S._relu(a, masks, q + "h2")
        out = out + a
    if taps is not None:
        taps[pre + "s"] = out
    return out


def st_block(x, p, i, A, training, new_stats=None, taps=None, blocks=None, masks=None):
    """models/stgin.py:58-66"""
    f, s, res = (blocks or BLOCKS)[i]
    pre = "l%d." % i
    kind = S.block_residual_kind(x.shape[1], f, s, res)
    if kind == "none":
        r = None
    elif kind == "identity":
        r = x
    else:
        r = F.conv2d(x, S.hwio_to_oihw(p[pre + "res.kernel"]), p[pre + "res.bias"], stride=(s, 1))
        r = S.batch_norm(r, p[pre + "res_bn.gamma"], p[pre + "res_bn.beta"], p[pre + "res_bn.moving_mean"],
                         p[pre + "res_bn.moving_var"], training, (0, 2, 3), True, new_stats, pre + "res_bn")
    g = graph_iso_conv(x, p, pre, A, training, new_stats, taps, masks)
    h = S.batch_norm(g, p[pre + "bn1.gamma"], p[pre + "bn1.beta"], p[pre + "bn1.moving_mean"], p[pre + "bn1.moving_var"],
                     training, (0, 2, 3), True, new_stats, pre + "bn1")
    if taps is not None:
        taps[pre + "h_pre"] = h
    h = S._relu(h, masks, pre + "h")
    u = S.temporal_conv(h, p[pre + "tcn.kernel"], p[pre + "tcn.bias"], s)
    z = S.batch_norm(u, p[pre + "bn2.gamma"], p[pre + "bn2.beta"], p[pre + "bn2.moving_mean"], p[pre + "bn2.moving_var"],
                     training, (0, 2, 3), True, new_stats, pre + "bn2")
    if r is not None:
        z = z + r
    y = S._relu(z, masks, pre + "y")
    if taps is not None:
        taps[pre + "u"], taps[pre + "y"], taps[pre + "y_pre"] = u, y, z
    return y


def forward(p, x, training, new_stats=None, taps=None, blocks=None, masks=None):
    """models/stgin.py:117-140"""
    N, C, T, V, M = x.shape
    h = S.data_bn(x, p, training, new_stats)
    if taps is not None:
        taps["x0"] = h
    for i in range(len(blocks or BLOCKS)):
        h = st_block(h, p, i, p["A"], training, new_stats, taps, blocks, masks)
    feat = h.mean(dim=(2, 3)).reshape(N, M, -1).mean(dim=1)
    return feat @ p["logits.kernel"][0, 0] + p["logits.bias"]


def loss_and_grads(p, x, labels, global_batch_size=None, blocks=None, masks=None):
    names = S.trainable_names(p)
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    new_stats, taps = {}, {}
    logits = forward(q, x, True, new_stats, taps, blocks, masks)
    loss = S.loss_fn(logits, labels, global_batch_size or x.shape[0])
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    return logits.detach(), loss.detach(), dict(zip(names, grads)), new_stats, {k: v.detach() for k, v in taps.items()}
