"""CPU: the ST-GCN oracle is self-consistent -- independent float64 numpy re-derivation, finite
differences, Keras/TF bookkeeping (SAME padding, parameter count, LR schedule, Nesterov)."""
import numpy as np
import torch

from oracle import stgcn as O
from oracle.stgcn_np import forward_np

SMALL = [(64, 1, False), (64, 1, True), (128, 2, True), (128, 1, True)]


def test_same_padding():
    assert O.same_pad(300, 9, 1) == (300, 4, 4)
    assert O.same_pad(300, 9, 2) == (150, 3, 4)     # extra pad at the END (TF SAME)
    assert O.same_pad(150, 9, 2) == (75, 3, 4)
    assert O.same_pad(300, 1, 2) == (150, 0, 0)
    assert O.same_pad(13, 9, 2) == (7, 4, 4)


def test_param_count():
    p = O.init_params(60)
    n = sum(v.numel() for k, v in p.items() if O.is_trainable(k))
    assert n == 3080082
    p = O.init_params(120)
    assert sum(v.numel() for k, v in p.items() if O.is_trainable(k)) == 3095502


def test_forward_matches_numpy_rederivation():
    p = O.randomize_affine(O.init_params(7, seed=3, dtype=torch.float64, blocks=SMALL))
    x, _ = O.synthetic_batch(2, seed=5, T=14, dtype=torch.float64, num_classes=7)
    lt = O.forward(p, x, True, blocks=SMALL).numpy()
    ln = forward_np({k: v.numpy() for k, v in p.items()}, x.numpy(), blocks=SMALL)
    assert np.abs(lt - ln).max() < 1e-11


def test_gradients_match_finite_differences():
    p = O.randomize_affine(O.init_params(5, seed=1, dtype=torch.float64, blocks=SMALL))
    x, y = O.synthetic_batch(2, seed=2, T=10, dtype=torch.float64, num_classes=5)
    _, _, grads, _, _ = O.loss_and_grads(p, x, y, blocks=SMALL)
    gen = torch.Generator().manual_seed(0)
    for name in ["l0.gcn.kernel", "l2.res.kernel", "l3.tcn.kernel", "l1.bn1.gamma", "data_bn.beta", "l2.res_bn.gamma",
                 "logits.bias", "l1.tcn.bias"]:
        g = grads[name]
        flat = int(torch.randint(0, g.numel(), (1,), generator=gen))
        eps = 1e-6
        vals = []
        for sgn in (1, -1):
            q = {k: v.clone() for k, v in p.items()}
            q[name].reshape(-1)[flat] += sgn * eps
            vals.append(O.loss_fn(O.forward(q, x, True, blocks=SMALL), y, 2).item())
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - g.reshape(-1)[flat].item()) <= 1e-5 * max(1.0, abs(fd)) + 1e-7, name


def test_conv_bias_gradients_vanish_before_batchnorm():
    """A bias in front of a train-mode BatchNorm has zero gradient (checks the oracle's BN wiring)."""
    p = O.randomize_affine(O.init_params(5, seed=1, dtype=torch.float64, blocks=SMALL))
    x, y = O.synthetic_batch(2, seed=2, T=10, dtype=torch.float64, num_classes=5)
    _, _, grads, _, _ = O.loss_and_grads(p, x, y, blocks=SMALL)
    for name in ["l1.tcn.bias", "l2.res.bias"]:
        assert grads[name].abs().max().item() < 1e-12
    # ... but NOT the graph-conv bias: it enters as b_k * colsum(A_k)[w], which varies per joint
    assert grads["l0.gcn.bias"].abs().max().item() > 1e-3


def test_lr_schedule_and_nesterov():
    assert O.lr_schedule(0) == 0.1 and O.lr_schedule(6250) == 0.1          # value[i] while step <= boundary
    assert abs(O.lr_schedule(6251) - 0.01) < 1e-12 and abs(O.lr_schedule(31251) - 0.001) < 1e-12
    p = {"w": torch.tensor([1.0, -2.0])}
    vel = {}
    g = {"w": torch.tensor([0.5, 0.25])}
    O.sgd_nesterov_step(p, g, vel, lr=0.1)
    assert torch.allclose(vel["w"], torch.tensor([-0.05, -0.025]))
    assert torch.allclose(p["w"], torch.tensor([1.0 - 0.045 - 0.05, -2.0 - 0.0225 - 0.025]))


def test_moving_statistics_update():
    p = O.init_params(5, blocks=SMALL[:1])
    x, _ = O.synthetic_batch(2, T=10, num_classes=5)
    new = {}
    O.forward(p, x, True, new, blocks=SMALL[:1])
    assert set(new) == {"data_bn.moving_mean", "data_bn.moving_var", "l0.bn1.moving_mean", "l0.bn1.moving_var",
                        "l0.bn2.moving_mean", "l0.bn2.moving_var"}
    assert torch.all(new["l0.bn1.moving_var"] > 0.98)   # 0.99*1 + 0.01*var


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the oracle's primitives against an independent scipy / numpy implementation with published
# semantics, on the reference's bundled NTU clips (fixture: tests/golden/stgcn_primitives.npz, generator next to it)

def test_oracle_primitives_match_the_independent_scipy_numpy_fixture(golden_dir):
    """oracle/stgcn.py (torch CPU ops) vs tests/golden/make_golden_stgcn_primitives.py (scipy.signal.correlate on explicitly
    padded arrays, numpy BatchNormalization, per-slice matrix products): data_bn prologue, GraphConvTD, train-mode BN +
    moving statistics (unbiased in the 4-D path, biased in data_bn), the 9x1 convolution with TF-SAME pads (4,4) / (3,4) and
    the strided 1x1 residual convolution, in float64 on two bundled NTU clips.  What this CANNOT pin (no TensorFlow here)
    is listed in oracle/stgcn.py's header."""
    import os

    import numpy as np
    import torch
    from oracle import stgcn as O
    from oracle.graph import spatial_adjacency
    g = np.load(os.path.join(golden_dir, "stgcn_primitives.npz"))
    clips = torch.from_numpy(np.load(os.path.join(golden_dir, "ntu_clips_0_2.npy"))).double()
    t = lambda k: torch.from_numpy(np.asarray(g[k], dtype=np.float64))
    A = torch.from_numpy(spatial_adjacency().astype(np.float32)).double()
    close = lambda a, b, tol=1e-10: float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-30)
    # data_bn (models/stgcn.py:136-147): channel index v C + c, statistics over (N M, T), biased moving variance
    p = {"data_bn.gamma": t("dbn_gamma"), "data_bn.beta": t("dbn_beta"), "data_bn.moving_mean": torch.zeros(75, dtype=torch.float64),
         "data_bn.moving_var": torch.ones(75, dtype=torch.float64)}
    ns = {}
    x0 = O.data_bn(clips, p, True, ns)
    assert close(x0, torch.from_numpy(g["x0"]).double(), 1e-6)          # fixture stored as float32
    assert close(ns["data_bn.moving_mean"], t("dbn_mm")) and close(ns["data_bn.moving_var"], t("dbn_mv"))
    # GraphConvTD (models/gcn.py:199-209)
    gg = O.graph_conv_td(x0, t("kg"), t("bg"), A)
    assert close(gg.sum(dim=(2, 3)), t("g_sum"), 2e-6)
    assert close(gg[:, :, ::37, ::6], t("g_probe"), 2e-6)               # x0 re-derived here in float64, fixture's x0 identical to 1e-16
    # BN train mode (axis 1 of a 4-D tensor: unbiased moving variance) + ReLU
    ns = {}
    F_ = g["bn_gamma"].shape[0]
    h = O.batch_norm(gg, t("bn_gamma"), t("bn_beta"), torch.zeros(F_, dtype=torch.float64), torch.ones(F_, dtype=torch.float64),
                     True, (0, 2, 3), True, ns, "bn")
    h = torch.relu(h)
    assert close(h[:, :, ::37, ::6], t("h_probe"), 1e-9)
    assert close(ns["bn.moving_mean"], t("bn_mm"), 1e-9) and close(ns["bn.moving_var"], t("bn_mv"), 1e-9)
    # 9x1 convolution, TF 'SAME': stride 1 pads (4,4); stride 2 pads (3,4) -- the first / last frames are where a wrong pad shows
    u1 = O.temporal_conv(h, t("kt"), t("bt"), 1)
    u2 = O.temporal_conv(h, t("kt"), t("bt"), 2)
    assert u1.shape[2] == 300 and u2.shape[2] == 150
    assert close(u1[:, :, ::37, ::6], t("u1_probe"), 1e-9) and close(u1[:, :, [0, 1, 2, 3, 296, 297, 298, 299]], t("u1_edges"), 1e-9)
    assert close(u2[:, :, ::19, ::6], t("u2_probe"), 1e-9) and close(u2[:, :, [0, 1, 2, 147, 148, 149]], t("u2_edges"), 1e-9)
    assert close(u1.sum(dim=(2, 3)), t("u1_sum"), 1e-9) and close(u2.sum(dim=(2, 3)), t("u2_sum"), 1e-9)
    # a symmetric (4,4) pad at stride 2 -- the classic mistake -- is NOT what the fixture holds
    import torch.nn.functional as F
    wrong = F.conv2d(F.pad(h, (0, 0, 4, 4)), O.hwio_to_oihw(t("kt")), t("bt"), stride=(2, 1))[:, :, :150]
    assert not close(wrong[:, :, [0, 1, 2, 147, 148, 149]], t("u2_edges"), 1e-3)
    # strided 1x1 residual convolution (models/stgcn.py:47-54): samples t = 0, 2, 4, ...
    r2 = O.temporal_conv(x0, t("kr"), t("br"), 2)
    assert close(r2[:, :, ::19, ::6], t("r2_probe"), 1e-9) and close(r2.sum(dim=(2, 3)), t("r2_sum"), 1e-9)
