#!/usr/bin/env python3
"""Fixture generator: the Path A primitives computed by an INDEPENDENT third implementation with published semantics
 -- scipy.signal.correlate on explicitly padded arrays for the convolutions, plain numpy for
BatchNormalization and the adjacency contraction -- on two of the reference's bundled NTU clips
(data/NTU_preprocessed_skeleton_examples.npy, clips 0 and 2 = tests/golden/ntu_clips_0_2.npy), in float64.

It pins `oracle/stgcn.py`'s conv / BN / SAME-pad / GraphConvTD / data_bn primitives (which are torch CPU ops) against code that
shares nothing with them but the published definitions:
  * TF 'SAME' padding (tf.nn.convolution docs): out = ceil(T / s), pad_total = max((out - 1) s + k - T, 0),
    pad_before = pad_total // 2, pad_after = pad_total - pad_before  ->  9x1 / stride 2 on T = 300: (3, 4);
  * Conv2D kernels HWIO, cross-correlation (no kernel flip), bias added per output channel;
  * Keras BatchNormalization(axis=1), training: (x - mean) / sqrt(biased_var + 1e-3) * gamma + beta; moving statistics
    m <- 0.99 m + 0.01 batch (unbiased batch variance in the fused 4-D path, biased in the 3-D path);
  * GraphConvTD (models/gcn.py:199-209): 1x1 conv to K F channels, channel k F + c -> (k, c), out = sum_k y_k . A_k;
  * the data_bn prologue (models/stgcn.py:136-147): channel index v C + c, statistics over (N M, T).
What stays unverifiable without TensorFlow is listed in oracle/stgcn.py's header.

Run from the repository root:  python tests/golden/make_golden_stgcn_primitives.py
Writes tests/golden/stgcn_primitives.npz (inputs are regenerated from the seeds below; outputs float64)."""
import os
import sys

import numpy as np
from scipy.signal import correlate

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "skeleton-action-recognition_amd"))


def tf_same_pad(T, k, s):
    out = -(-T // s)
    total = max((out - 1) * s + k - T, 0)
    return out, total // 2, total - total // 2


def conv_kx1(x, kernel, bias, stride):
    """x (B, Cin, T, V) float64; kernel HWIO (k, 1, Cin, Cout); 'SAME' along T, stride (s, 1)."""
    B, Cin, T, V = x.shape
    k, _, _, Cout = kernel.shape
    To, pb, pe = tf_same_pad(T, k, stride)
    xp = np.pad(x, ((0, 0), (0, 0), (pb, pe), (0, 0)))
    out = np.zeros((B, Cout, To, V))
    for b in range(B):
        for m in range(Cout):
            acc = np.zeros((xp.shape[2] - k + 1, V))
            for c in range(Cin):
                acc += correlate(xp[b, c], kernel[:, 0, c, m][:, None], mode="valid")   # cross-correlation along T
            out[b, m] = acc[::stride][:To] + bias[m]
    return out


def batch_norm_train(x, gamma, beta, mm, mv, axes, unbiased_moving):
    mean = x.mean(axis=axes, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=axes, keepdims=True)
    n = x.size // x.shape[1]
    y = (x - mean) / np.sqrt(var + 1e-3) * gamma.reshape(mean.shape) + beta.reshape(mean.shape)
    v_mov = var * (n / (n - 1.0)) if unbiased_moving else var
    return y, 0.99 * mm + 0.01 * mean.reshape(-1), 0.99 * mv + 0.01 * v_mov.reshape(-1)


def graph_conv_td(x, kernel, bias, A):
    B, Cin, T, V = x.shape
    K = A.shape[0]
    F = kernel.shape[3] // K
    y = np.zeros((B, K * F, T, V))
    for o in range(K * F):
        for c in range(Cin):
            y[:, o] += kernel[0, 0, c, o] * x[:, c]
        y[:, o] += bias[o]
    out = np.zeros((B, F, T, V))
    for k in range(K):
        out += y[:, k * F:(k + 1) * F] @ A[k]          # (B, F, T, V) @ (V, V): sum_v y[.., v] A_k[v, w]
    return out


def data_bn(x, gamma, beta):
    N, C, T, V, M = x.shape
    h = np.zeros((N * M, V * C, T))
    for n in range(N):
        for m in range(M):
            for v in range(V):
                for c in range(C):
                    h[n * M + m, v * C + c] = x[n, c, :, v, m]
    y, mm, mv = batch_norm_train(h, gamma, beta, np.zeros(V * C), np.ones(V * C), (0, 2), False)
    out = np.zeros((N * M, C, T, V))
    for v in range(V):
        for c in range(C):
            out[:, c, :, v] = y[:, v * C + c]
    return out, mm, mv


def main():
    from graph.ntu_rgb_d import Graph
    clips = np.load(os.path.join(HERE, "ntu_clips_0_2.npy")).astype(np.float64)      # (2, 3, 300, 25, 2)
    A = np.asarray(Graph().A, dtype=np.float32).astype(np.float64)
    rng = np.random.default_rng(20260403)
    F_, C = 8, 3
    g0, b0 = 1 + 0.2 * rng.standard_normal(75), 0.1 * rng.standard_normal(75)
    x0, dmm, dmv = data_bn(clips, g0, b0)                                             # (4, 3, 300, 25)
    kg, bg = 0.3 * rng.standard_normal((1, 1, C, 3 * F_)), 0.1 * rng.standard_normal(3 * F_)
    g = graph_conv_td(x0, kg, bg, A)                                                  # (4, 8, 300, 25)
    g1, b1 = 1 + 0.2 * rng.standard_normal(F_), 0.1 * rng.standard_normal(F_)
    h, mm1, mv1 = batch_norm_train(g, g1, b1, np.zeros(F_), np.ones(F_), (0, 2, 3), True)
    h = np.maximum(h, 0)
    kt, bt = 0.2 * rng.standard_normal((9, 1, F_, F_)), 0.1 * rng.standard_normal(F_)
    u1 = conv_kx1(h, kt, bt, 1)                                                       # (4, 8, 300, 25), pads (4, 4)
    u2 = conv_kx1(h, kt, bt, 2)                                                       # (4, 8, 150, 25), pads (3, 4)
    kr, br = 0.3 * rng.standard_normal((1, 1, C, F_)), 0.1 * rng.standard_normal(F_)
    r2 = conv_kx1(x0, kr, br, 2)                                                      # strided 1x1 residual: samples t = 0, 2, ...
    assert tf_same_pad(300, 9, 2) == (150, 3, 4) and tf_same_pad(300, 9, 1) == (300, 4, 4) and tf_same_pad(300, 1, 2) == (150, 0, 0)
    np.savez_compressed(os.path.join(HERE, "stgcn_primitives.npz"), seed=20260403,
                        dbn_gamma=g0, dbn_beta=b0, x0=x0.astype(np.float32), dbn_mm=dmm, dbn_mv=dmv,
                        kg=kg, bg=bg, g_sum=g.sum(axis=(2, 3)), g_probe=g[:, :, ::37, ::6],
                        bn_gamma=g1, bn_beta=b1, h_probe=h[:, :, ::37, ::6], bn_mm=mm1, bn_mv=mv1,
                        kt=kt, bt=bt, u1_sum=u1.sum(axis=(2, 3)), u1_probe=u1[:, :, ::37, ::6], u1_edges=u1[:, :, [0, 1, 2, 3, 296, 297, 298, 299]],
                        u2_sum=u2.sum(axis=(2, 3)), u2_probe=u2[:, :, ::19, ::6], u2_edges=u2[:, :, [0, 1, 2, 147, 148, 149]],
                        kr=kr, br=br, r2_probe=r2[:, :, ::19, ::6], r2_sum=r2.sum(axis=(2, 3)))
    print("wrote stgcn_primitives.npz")


if __name__ == "__main__":
    main()
