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
