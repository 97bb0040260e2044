"""Independent float64 numpy re-derivation of the ST-GCN forward, used ONLY to
cross-check oracle/stgcn.py (different formulation on purpose):

  * graph conv in the reordered form  out = sum_k W_k (x A_k) + sum_k b_k colsum(A_k)
    instead of conv-then-einsum (models/gcn.py:199-209),
  * temporal conv as an explicit tap loop with hand-computed TF-SAME pads
    (models/stgcn.py:29-36),
  * batch-norm written out from its definition (Keras eps 1e-3, biased variance).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np

from .stgcn import BLOCKS, BN_EPS, block_residual_kind, same_pad


def _bn(x, gamma, beta, axes):
    mean = x.mean(axis=axes, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=axes, keepdims=True)
    shape = [1, -1] + [1] * (x.ndim - 2)
    return (x - mean) / np.sqrt(var + BN_EPS) * gamma.reshape(shape) + beta.reshape(shape)


def _gcn(x, kernel, bias, A):
    B, Cin, T, V = x.shape
    K = A.shape[0]
    W = kernel[0, 0]                       # (Cin, K*F)
    F_ = W.shape[1] // K
    out = np.zeros((B, F_, T, V))
    for k in range(K):
        xa = x @ A[k]                      # (B,Cin,T,V) . (V,W)
        Wk = W[:, k * F_:(k + 1) * F_]     # (Cin, F)
        out += np.einsum("bctw,cf->bftw", xa, Wk)
        out += bias[k * F_:(k + 1) * F_][None, :, None, None] * A[k].sum(axis=0)[None, None, None, :]
    return out


def _tconv(x, kernel, bias, stride):
    B, C, T, V = x.shape
    kt = kernel.shape[0]
    To, pb, _ = same_pad(T, kt, stride)
    Fo = kernel.shape[3]
    out = np.zeros((B, Fo, To, V)) + bias[None, :, None, None]
    for to in range(To):
        for dt in range(kt):
            t = to * stride + dt - pb
            if 0 <= t < T:
                out[:, :, to, :] += np.einsum("bcv,cf->bfv", x[:, :, t, :], kernel[dt, 0])
    return out


def forward_np(p, x, blocks=None):
    """train-mode forward; p: dict name -> numpy float64; x (N,C,T,V,M)."""
    N, C, T, V, M = x.shape
    h = np.zeros((N * M, C, T, V))
    for n in range(N):
        for m in range(M):
            h[n * M + m] = x[n, :, :, :, m]
    # data_bn channel = v*C + c, statistics over (n, m, t)  (models/stgcn.py:142-147)
    for v in range(V):
        for c in range(C):
            s = h[:, c, :, v]
            mean = s.mean()
            var = ((s - mean) ** 2).mean()
            ch = v * C + c
            h[:, c, :, v] = (s - mean) / np.sqrt(var + BN_EPS) * p["data_bn.gamma"][ch] + p["data_bn.beta"][ch]
    A = p["A"]
    for i in range(len(blocks or BLOCKS)):
        f, s, res = (blocks or BLOCKS)[i]
        pre = "l%d." % i
        kind = block_residual_kind(h.shape[1], f, s, res)
        if kind == "none":
            r = 0.0
        elif kind == "identity":
            r = h
        else:
            r = np.einsum("bctv,cf->bftv", h[:, :, ::s, :], p[pre + "res.kernel"][0, 0]) \
                + p[pre + "res.bias"][None, :, None, None]
            r = _bn(r, p[pre + "res_bn.gamma"], p[pre + "res_bn.beta"], (0, 2, 3))
        g = _gcn(h, p[pre + "gcn.kernel"], p[pre + "gcn.bias"], A)
        a = np.maximum(_bn(g, p[pre + "bn1.gamma"], p[pre + "bn1.beta"], (0, 2, 3)), 0)
        u = _tconv(a, p[pre + "tcn.kernel"], p[pre + "tcn.bias"], s)
        h = np.maximum(_bn(u, p[pre + "bn2.gamma"], p[pre + "bn2.beta"], (0, 2, 3)) + r, 0)
    feat = h.mean(axis=(2, 3)).reshape(N, M, -1).mean(axis=1)
    return feat @ p["logits.kernel"][0, 0] + p["logits.bias"]
